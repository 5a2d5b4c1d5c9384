#!/bin/bash
# cycle accounting of the fused RHS under NDCN_FUSED_DBG switches (results are WRONG under dbg != 0/8192: timing only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for d in ${DBG_LIST:-8192 8256 8258 8194}; do
  echo "== NDCN_FUSED_DBG=$d (8192: split mfma|dump; +2 no gather; +4 no epilogue; +64 no weight refills)"
  NDCN_FUSED_DBG=$d NDCN_FUSED_TIMING=${NT:-4} python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200
done
