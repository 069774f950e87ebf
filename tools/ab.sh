# scratch A/B
timeout -k 10 280 python tools/quick_perf.py gpt bf16only > gpurun_out/ab_new.txt 2>&1 && grep "gpt bf16" gpurun_out/ab_new.txt && for c in 300 700 1200; do IXTTS_LIB=voice-tts_amd/libixtts_hip_trace.so timeout -k 10 280 python tools/trace_decode.py $c 2 > gpurun_out/trace_$c.txt 2>&1; tail -9 gpurun_out/trace_$c.txt; done
