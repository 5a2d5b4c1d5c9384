#!/bin/bash
# fused RHS: per-mode cycle accounting + bench (A/B inside one box)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/exp_tests.log
{
  NDCN_FUSED_TIMING=9 python bench.py --steps 2 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "fused2 timing|ms_per_step" | grep -v "block   0" | cut -c1-220
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"kernels": {"rhs_fused[^}]*}'
} > gpurun_out/exp_epi.log 2>&1
