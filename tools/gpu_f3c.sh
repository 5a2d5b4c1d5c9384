#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
bash tools/gpu_ab_env.sh "NDCN_RHS_FUSED3=1" "NDCN_FUSED3_DBG=70"
for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp $f ndcn_amd/libndcn_hip.so; echo "##### $f"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1; bash tools/gpu_ab_env.sh "NDCN_FUSED3_DBG=0"; done
