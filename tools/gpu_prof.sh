#!/bin/bash
# rocprofv3 kernel stats of bench.py (per fused-kernel variant)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=${1:-tmp}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$R" -o $R -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass ${2:-} > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_$R.log" 2>&1)
grep '^{"metric' gpurun_out/rocprof_$R.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
find gpurun_out/prof_$R -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_kernel_stats.csv
head -12 gpurun_out/${R}_kernel_stats.csv | cut -d, -f1-4 | sed 's/ndcn:://; s/(int const\*.*EpiArgs)//' | cut -c1-120
