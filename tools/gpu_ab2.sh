#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --no-profile-pass --steps 12 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$1 -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --steps 12 > /dev/null 2>&1); f=$(find /tmp/p_$1 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rhs_fused2_kernel' in r['Name']:
        print('  ', r['Name'].split('<')[1].split('>')[0], 'avg %.3f ms' % (float(r['AverageNs']) / 1e6))
PY
}
cp ndcn_amd/libndcn_hip.so /tmp/default.so
echo "=== default"; run d
for f in gpurun_in_*.so; do cp $f ndcn_amd/libndcn_hip.so; echo "=== $f"; run $(basename $f .so); done
cp /tmp/default.so ndcn_amd/libndcn_hip.so
