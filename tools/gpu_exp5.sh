#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_kernels.py 2>&1 | grep -v amdgpu > gpurun_out/kernels.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
timeout 600 python bench.py --sharded --no-cpu-baseline > gpurun_out/bench_sharded.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_sharded.log
tail -8 gpurun_out/pytest_gpu.log | cut -c1-220
cut -c1-200 gpurun_out/kernels.log
tail -2 gpurun_out/bench.log | cut -c1-300
tail -2 gpurun_out/bench_sharded.log | cut -c1-300
