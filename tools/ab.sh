timeout -k 10 600 python bench.py --bigvgan-only 2>/dev/null | tail -1
timeout -k 10 900 python bench.py --decode beam --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_beam.json 2> gpurun_out/bench_beam.err; tail -2 gpurun_out/bench_beam.err; cut -c1-260 gpurun_out/bench_beam.json; grep -o "stage_ms_per_step[^}]*}" gpurun_out/bench_beam.json
