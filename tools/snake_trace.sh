# per-call durations of the Snake passes of ONE BigVGAN forward (serial streams), by grid size = by stage
set -e
bash tools/gpu_entry.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IXTTS_BV_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/prof_snk -- python3 tools/prof_bigvgan.py ${1:-1892} > gpurun_out/r03/snk.log 2>&1
f=$(find gpurun_out/r03/prof_snk -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "aa_snake_planes" in n:
        key = (n.split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], r.get("Workgroup_Size_X", ""))
        by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v = sorted(v)
    print(k, "calls", len(v), "median us %.1f" % v[len(v) // 2], "min %.1f" % v[0])
PY
rm -rf gpurun_out/r03/prof_snk
