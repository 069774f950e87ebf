timeout -k 10 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log; timeout -k 10 300 python tools/prof_s2mel.py 25 2>&1 | grep s2mel
