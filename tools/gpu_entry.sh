#!/bin/bash
# Prefix of every gpurun call of this round: if the box is cold (first `import torch` pages the image in), run the cold-start
# breakdown right away -- the only chance to see what a fresh worker's first request waits for.
mkdir -p gpurun_out/r03
t0=$(date +%s.%N)
python -c "import torch" 2>/dev/null
t1=$(date +%s.%N)
dt=$(python -c "print(f'{$t1 - $t0:.1f}')")
echo "[gpu_entry] import torch took ${dt}s" | tee -a gpurun_out/r03/box_temperature.log
if python -c "import sys; sys.exit(0 if $t1 - $t0 > 12 else 1)"; then
  echo "[gpu_entry] cold box: running tools/cold_start.py first" | tee -a gpurun_out/r03/box_temperature.log
  timeout -k 10 900 python tools/cold_start.py > gpurun_out/r03/cold_start_coldbox_$(date +%H%M%S).log 2>&1
  tail -60 gpurun_out/r03/cold_start_coldbox_*.log
fi
