#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python tools/bench_kernels.py 2>&1 | grep -v amdgpu > gpurun_out/kernels.log
# HBM traffic + L2 hit counters for three SpMM variants (separate passes: FETCH_SIZE costs 3 TCC slots)
for v in spmm_wide_bpc4 spmm_wide_tile256 spmm_blocked; do
  case $v in
    spmm_wide_bpc4) export NDCN_SPMM_BLOCKS_PER_CU=4; unset TILE NDCN_SPMM_WIDE;;
    spmm_wide_tile256) export NDCN_SPMM_BLOCKS_PER_CU=4 TILE=256; unset NDCN_SPMM_WIDE;;
    spmm_blocked) export NDCN_SPMM_WIDE=0; unset TILE;;
  esac
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${v}_$(echo $c | tr ' ' '_')" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" --one $v > /dev/null 2>&1)
  done
done
find gpurun_out -name "*counter_collection.csv" | head -20
cat gpurun_out/kernels.log
