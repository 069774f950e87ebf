cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps2 -- python tools/prof_s2mel.py 5 > gpurun_out/ps2.log 2>&1
rm -f gpurun_out/ps2/*/*kernel_trace.csv
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/ps2/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f}us {float(r['TotalDurationNs'])/1e6:9.1f}ms {r['Percentage']:>6s}%")
PY
grep s2mel gpurun_out/ps2.log
