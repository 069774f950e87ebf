python tools/prof_rows.py 2>&1 | grep prefill
timeout -k 10 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -5 gpurun_out/t.log
