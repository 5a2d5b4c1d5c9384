#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1
grep -v amdgpu gpurun_out/kernels.log | cut -c1-1200
