timeout -k 10 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_infer_v2.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
python tools/beam_perf.py 2>&1 | grep beam
