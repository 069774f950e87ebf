cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/prof_attn_full.py && timeout -k 10 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_attn -- python tools/prof_attn_full.py > gpurun_out/pmc_attn.log 2>&1
rm -f gpurun_out/pmc_attn/*/*kernel_trace.csv
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_attn/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_full_f32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
