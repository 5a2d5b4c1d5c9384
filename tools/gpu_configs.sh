#!/bin/bash
# every single-GPU configuration through bench.py (+ the new full-size property tests)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r02b}
timeout 1500 python -m pytest tests -m gpu -x -q -k "config_c or hipgraph" > gpurun_out/pytest_cfg.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_cfg.log
tail -5 gpurun_out/pytest_cfg.log | cut -c1-300
: > gpurun_out/${R}_bench_configs.jsonl
for c in M C2 C3 C5; do
  extra=""; [ "$c" != "M" ] && extra="--no-cpu-baseline"
  timeout 900 python bench.py --config $c $extra 2> gpurun_out/bench_$c.err | grep '^{"metric' >> gpurun_out/${R}_bench_configs.jsonl
  tail -2 gpurun_out/bench_$c.err | cut -c1-300
done
for l in degree; do
  for c in C2 C3; do
    timeout 900 python bench.py --config $c --layout $l --no-cpu-baseline 2> gpurun_out/bench_${c}_$l.err | grep '^{"metric' >> gpurun_out/${R}_bench_configs.jsonl
  done
done
python tools/bench_c5.py 2>/dev/null | tail -1 > gpurun_out/${R}_c5_graph.json
python - "$R" <<'PY'
import json, sys
for l in open('gpurun_out/%s_bench_configs.jsonl' % sys.argv[1]):
    d = json.loads(l)
    r = d.get('roofline') or {}
    print(d['config']['workload'][:60], '| ms/step', d['ms_per_step'], '| value', d['value'], '| roof', r.get('kernel'), r.get('avg_ms'), r.get('frac'), '| spmm', d.get('spmm_standalone'))
PY
cat gpurun_out/${R}_c5_graph.json | cut -c1-500
