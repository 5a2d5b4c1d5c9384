#!/bin/bash
# First GPU pass: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/env.log 2>&1
import torch, subprocess
print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
from ndcn_amd import device_info, _lib
_lib.load()
print(device_info())
import os
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l][:2])
PY
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r01 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1; echo "rocprof exit $?" >> "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log")
ls -R gpurun_out/prof | head -30 >> gpurun_out/rocprof.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
