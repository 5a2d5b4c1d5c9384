#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
bash tools/gpu_ab_env.sh "NDCN_RHS_FUSED3=0" "NDCN_RHS_FUSED3=1"
for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp $f ndcn_amd/libndcn_hip.so; echo "##### $f"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -1; bash tools/gpu_ab_env.sh "NDCN_FUSED3_DBG=0"; done
