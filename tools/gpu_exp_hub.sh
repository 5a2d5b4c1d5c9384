#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -8
timeout 900 python tools/bench_graphs.py --networks grid,power_law,small_world,random,community 2>&1 | grep -E "^\{|Error|error"
} > gpurun_out/exp_hub.log 2>&1
