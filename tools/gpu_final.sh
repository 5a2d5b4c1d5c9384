#!/bin/bash
# Round-end measurement pass: GPU tests, smoke, bench, rocprofv3 kernel stats and HBM-traffic PMC passes.
set -u
R=${1:-r01}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$R" -o $R -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${R}_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass --steps 6 --warmup 1 > /dev/null 2>&1)
done
python - <<'PY' > gpurun_out/traffic_$R.json
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_*_%s/**/*counter_collection.csv' % c, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out[c] = {k: {'launches': len(v), 'mean_KB': sum(v) / len(v)} for k, v in agg.items()}
print(json.dumps(out, indent=1))
PY
grep -n "passed\|failed\|FAILED" gpurun_out/pytest_gpu.log | tail -5 | cut -c1-200
tail -2 gpurun_out/smoke.log
grep '^{"metric' gpurun_out/bench.log | cut -c1-400
head -8 gpurun_out/prof_$R/*kernel_stats.csv | cut -c1-200
