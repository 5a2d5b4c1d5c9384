#!/bin/bash
# A/B of an environment knob inside one box: bench ms/step + per-variant kernel times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { (cd /tmp && env $1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$2 -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass > /tmp/p_$2.log 2>&1); grep -o '"ms_per_step": [0-9.]*' /tmp/p_$2.log; f=$(find /tmp/p_$2 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rhs_fused' in r['Name'] and '_kernel<' in r['Name']:
        print('  ', r['Name'].split('(')[0][-40:], 'avg %.3f ms' % (float(r['AverageNs']) / 1e6))
PY
}
i=0
for kv in "$@"; do i=$((i+1)); echo "=== $kv"; run "$kv" v$i; done
