#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for c in M C2 C3; do
  timeout 900 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], {k:(v['launches'],v['avg_ms']) for k,v in d['kernels'].items()})"
done
