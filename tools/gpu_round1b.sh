#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r01 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1; echo "rocprof exit $?" >> "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log")
find gpurun_out/prof -type f | head -20 >> gpurun_out/rocprof.log
grep -v "^\[" gpurun_out/kernels.log | tail -12; tail -8 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log | cut -c1-1500
