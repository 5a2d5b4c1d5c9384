#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1
grep -v amdgpu gpurun_out/kernels.log | grep -E "variant|block 0" | cut -c1-330
grep -n "passed\|failed\|FAILED" gpurun_out/pytest_gpu.log | tail -4 | cut -c1-200
grep '^{"metric' gpurun_out/bench.log | cut -c1-330
