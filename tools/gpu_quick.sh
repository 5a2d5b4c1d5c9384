#!/bin/bash
# quick GPU check: selected tests + kernel microbench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "${1:-}" ]; then K=(-k "$1"); else K=(); fi
timeout 1500 python -m pytest tests -m gpu -x -q "${K[@]}" > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log
tail -15 gpurun_out/pytest_quick.log | cut -c1-300
if [ -n "${2:-}" ]; then
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1
  grep variant gpurun_out/kernels.log | cut -c1-300
fi
