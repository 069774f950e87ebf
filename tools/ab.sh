cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -- python tools/prof_gpt.py bf16 2 137 30 > gpurun_out/pmc_$C.log 2>&1
ls gpurun_out/pmc_$C/*/ | head
python - <<PY
import csv,glob,collections,statistics
fs=glob.glob('gpurun_out/pmc_$C/*/*counter_collection.csv')
print(fs)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    d[r['Kernel_Name'][:100]].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'gemv' in k or 'attn_decode' in k or 'sampler' in k:
        v=v[-400:]
        print(f"$C {k[12:100]:88s} n={len(v):5d} med={statistics.median(v):12.1f} mean={statistics.mean(v):12.1f}")
PY
rm -f gpurun_out/pmc_$C/*/*kernel_trace.csv
done
