#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for d in ${DBG_LIST:-0 6 70 1}; do
  echo "== NDCN_FUSED3_DBG=$d"
  NDCN_FUSED3_DBG=$d NDCN_FUSED3_TIMING=${NT:-3} python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "fused3 timing" | cut -c1-220
done
