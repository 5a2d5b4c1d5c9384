#!/bin/bash
mkdir -p gpurun_out
for d in 0 4096 4097; do
  echo "=== NDCN_FUSED_DBG=$d"
  NDCN_FUSED_DBG=$d NDCN_FUSED_TIMING=9 python bench.py --steps 2 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200 | head -5
done > gpurun_out/exp_epi2.log 2>&1
