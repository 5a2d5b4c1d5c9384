#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_ab_env.sh "NDCN_RHS_FUSED3=1"
for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp $f ndcn_amd/libndcn_hip.so; echo "##### $f"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fused or rhs" 2>&1 | grep -E "passed|failed" | tail -1; bash tools/gpu_ab_env.sh "NDCN_FUSED3_DBG=0" "NDCN_FUSED3_DBG=70"; done
