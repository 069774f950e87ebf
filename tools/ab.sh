cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_rows
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rows -- python tools/prof_rows.py > gpurun_out/prof_rows.log 2>&1
rm -f gpurun_out/prof_rows/*/*kernel_trace.csv
grep prefill gpurun_out/prof_rows.log | tail -1; cut -c1-140 gpurun_out/prof_rows/*/*kernel_stats.csv | head -5
timeout -k 10 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
