timeout -k 10 900 python -m pytest tests/test_gpu_gpt.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
timeout -k 10 280 python tools/quick_perf.py gpt bf16only 2>&1 | grep "gpt bf16"
