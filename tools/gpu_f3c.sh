#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_ab_env.sh "NDCN_RHS_FUSED3=1"
for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp $f ndcn_amd/libndcn_hip.so; echo "##### $f"; bash tools/gpu_ab_env.sh "NDCN_FUSED3_DBG=0"; done
