#!/bin/bash
# Is the bench power-capped?  Samples power / sclk / mclk with rocm-smi while bench.py runs.
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showuse --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|Max Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/power_probe.log 2>&1 &
SMI=$!
sleep 2
python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-profile-pass 2>&1 | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/power_bench.log
wait $SMI
