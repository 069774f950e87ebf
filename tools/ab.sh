python tools/prof_attn_full.py && timeout -k 10 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
