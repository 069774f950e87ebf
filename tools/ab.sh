cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_bv gpurun_out/prof_bv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bv -- python tools/prof_bigvgan.py > gpurun_out/prof_bv.log 2>&1
rm -f gpurun_out/prof_bv/*/*kernel_trace.csv
grep bigvgan gpurun_out/prof_bv.log
cut -c1-150 gpurun_out/prof_bv/*/*kernel_stats.csv | head -12
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_bv -- python tools/prof_bigvgan.py > gpurun_out/pmc_bv.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_bv/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv1d_mfma" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    n = len(d["SQ_WAVE_CYCLES"])
    tot = {c: sum(v) for c, v in d.items()}
    print(k, "launches", n, " mfma_busy/wave_cycles*4:", round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * tot["SQ_WAVE_CYCLES"]), 3),
          " wait_any", round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3), " wait_inst", round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
          " active", round(tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3), " lds_conf/wave_cyc", round(tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_WAVE_CYCLES"], 3))
PY
rm -f gpurun_out/pmc_bv/*/*kernel_trace.csv
