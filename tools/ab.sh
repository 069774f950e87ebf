timeout -k 10 300 python -m pytest tests/test_gpu_gpt.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['rtf'], d['stage_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in 137 1200; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt$P -- python tools/prof_gpt.py bf16 2 $P 40 > /dev/null 2>&1
python - <<PY
import csv,glob,statistics,collections
f=glob.glob('gpurun_out/pt$P/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:90]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    if 'attn_decode' in k:
        v=v[-600:]
        print(f"P=$P {k[12:70]:60s} med={statistics.median(v)/1e3:7.2f}us min={min(v)/1e3:7.2f}")
PY
rm -rf gpurun_out/pt$P
done
