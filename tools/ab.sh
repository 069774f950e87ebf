cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pbv -- python tools/quick_perf.py bigvgan > gpurun_out/pbv.log 2>&1
rm -f gpurun_out/pbv/*/*kernel_trace.csv
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/pbv/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f}us {float(r['TotalDurationNs'])/1e6:9.1f}ms {r['Percentage']:>6s}%")
PY
tail -2 gpurun_out/pbv.log
