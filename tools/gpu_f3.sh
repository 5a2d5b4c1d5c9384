#!/bin/bash
# fused3: parity tests, then A/B on the bench (NDCN_FUSED3_DBG: 1 no MFMA, 2 no fold, 4 no epilogue - timing only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -x -q > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log
tail -5 gpurun_out/pytest_quick.log | cut -c1-400
timeout 900 bash tools/gpu_ab_env.sh NDCN_RHS_FUSED3=0 NDCN_RHS_FUSED3=1 "NDCN_RHS_FUSED3=1 NDCN_FUSED3_DBG=1" "NDCN_RHS_FUSED3=1 NDCN_FUSED3_DBG=6"
