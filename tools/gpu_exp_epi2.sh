#!/bin/bash
mkdir -p gpurun_out
for d in 8192 8256; do
  echo "=== NDCN_FUSED_DBG=$d  (8192: mfma-wave columns = MFMA loop | dump; +64: no weight refills)"
  NDCN_FUSED_DBG=$d NDCN_FUSED_TIMING=9 python bench.py --steps 2 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200 | sed -n '1p;3p;6p;8p'
done > gpurun_out/exp_epi2.log 2>&1
