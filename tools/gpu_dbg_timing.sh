#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for d in 8192 8198 8196; do
  echo "== NDCN_FUSED_DBG=$d (8192: split mfma|dump; +2 no gather; +4 no epilogue)"
  NDCN_FUSED_DBG=$d NDCN_FUSED_TIMING=3 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200 | head -3
done
