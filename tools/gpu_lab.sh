#!/bin/bash
# SpMM laboratory on the GPU box: tools/micro/spmm_lab (built in the container, travels with the snapshot)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 tools/micro/spmm_lab ${1:-1000} ${2:-20} > gpurun_out/spmm_lab.log 2>&1
echo "exit $?" >> gpurun_out/spmm_lab.log
cat gpurun_out/spmm_lab.log | cut -c1-330
