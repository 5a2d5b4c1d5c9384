#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for pipe in 0 1; do for d in 8192 8193; do
  echo "== PIPE=$pipe NDCN_FUSED_DBG=$d (+1 no MFMA)"
  NDCN_FUSED_PIPE=$pipe NDCN_FUSED_DBG=$d NDCN_FUSED_TIMING=4 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200
done; done
