timeout -k 10 300 python -m pytest tests/test_gpu_gpt.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['rtf'], d['stage_ms_per_step'], d['roofline'])"
