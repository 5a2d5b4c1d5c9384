#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 1200 python tools/bench_kernels.py 2>&1 | grep -v amdgpu > gpurun_out/kernels.log
for v in spmm_union_r16; do
  export NDCN_UNION_ROWS=16 NDCN_UNION_CAP=56
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${v}_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" --one $v > /dev/null 2>&1)
  done
done
tail -15 gpurun_out/pytest_gpu.log
cat gpurun_out/kernels.log
