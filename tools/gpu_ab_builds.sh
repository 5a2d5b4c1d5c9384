#!/bin/bash
# A/B of builds of the library inside one box: default build vs gpurun_in_*.so (tests, cycle accounting, bench)
mkdir -p gpurun_out
run() {
  python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | head -3
  NDCN_FUSED_DBG=8192 NDCN_FUSED_TIMING=9 python bench.py --steps 2 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "fused2 timing" | grep "block 100" | cut -c1-200 | sed -n '1p;3p;6p;8p'
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"kernels": {"rhs_fused[^}]*}'
}
{
  echo "=== default"; run
  for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp $f ndcn_amd/libndcn_hip.so; echo "=== $f"; run; done
} > gpurun_out/exp_ab.log 2>&1
