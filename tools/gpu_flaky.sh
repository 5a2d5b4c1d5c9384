#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
echo "--- hip VJP, whole file x6"
for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_autograd.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -2 | tr '\n' ' '; echo; done
echo "--- NDCN_VJP=torch whole file x6"
for i in 1 2 3 4 5 6; do NDCN_VJP=torch python -m pytest tests/test_gpu_autograd.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -2 | tr '\n' ' '; echo; done
