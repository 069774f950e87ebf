#!/bin/bash
# Collect the round's judged artefacts on the GPU box: bench JSON, rocprofv3 kernel stats of the same command,
# and the PMC traffic passes.  Usage (from the repo root on the box): bash tools/profile_bench.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python bench.py --steps 3 --warmup 1 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/prof.err
rm -f gpurun_out/$TAG/prof/*/*kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_$C -- python tools/prof_gpt.py bf16 2 137 30 > gpurun_out/$TAG/pmc_$C.log 2>&1
  rm -f gpurun_out/$TAG/pmc_$C/*/*kernel_trace.csv
done
cat gpurun_out/$TAG/bench.json
