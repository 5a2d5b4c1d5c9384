#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
echo "--- hip VJP, whole file x14"
for i in $(seq 1 14); do python -m pytest tests/test_gpu_autograd.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -1; done | sort | uniq -c
