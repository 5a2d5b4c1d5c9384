#!/bin/bash
# GPU: autograd tests + training-step timing
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_autograd.py -m gpu -x -q > gpurun_out/pytest_autograd.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_autograd.log
tail -15 gpurun_out/pytest_autograd.log | cut -c1-300
timeout 900 python tools/bench_train.py > gpurun_out/train.log 2>&1
cat gpurun_out/train.log | cut -c1-400
