cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do echo "== dbg $d"; NDCN_REC_DBG=$d python tools/dbg_rec.py 2>&1 | grep -A12 "grid (16" | grep "np 5\|np 3"; done
