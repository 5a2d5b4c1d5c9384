#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python tools/bench_kernels.py 2>&1 | grep -v amdgpu > gpurun_out/kernels.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
cut -c1-150 gpurun_out/kernels.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-200
