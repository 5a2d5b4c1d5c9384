#!/bin/bash
# the pure-HBM right-hand side (no_control) on the metric's grid: bench line + rocprofv3 kernel stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --no-control --no-cpu-baseline 2> gpurun_out/bench_nc.err | grep '^{"metric' > gpurun_out/r02j_bench_no_control.json
python -c "
import json; d=json.load(open('gpurun_out/r02j_bench_no_control.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d['roofline'])[:600]); print(json.dumps(d['kernels'])[:700])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pnc -o x -- python $GRAFT_REPO_ROOT/bench.py --no-control --no-cpu-baseline --no-profile-pass > /tmp/pnc.log 2>&1)
f=$(find /tmp/pnc -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02j_bench_no_control_kernel_stats.csv; head -8 "$f" | cut -c1-200
