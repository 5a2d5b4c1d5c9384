#!/bin/bash
# same-box comparison of the single-GPU path and the sharded path on one rank (kernel mix of each): bash tools/sharded_ab.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD
run() {   # $1 label, rest: env assignments / bench flags
  local lab=$1; shift
  echo "=== $lab"
  (cd /tmp && rm -rf /tmp/p_$lab && env "$@" timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o x -- python $R/bench.py $FLAGS --no-cpu-baseline --no-profile-pass > /tmp/p_$lab.log 2>&1)
  grep -o '"ms_per_step": [0-9.]*' /tmp/p_$lab.log || tail -3 /tmp/p_$lab.log
  python - /tmp/p_$lab <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if float(r['Percentage']) >= 0.4:
        print('   %-60s n=%4s avg %.3f ms %5.1f %%' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
}
FLAGS="" run single X=0
FLAGS="--gpus 1 --sharded --sharded-impl device" run sharded0 NDCN_SELF_HALO=0
FLAGS="--gpus 1 --sharded --sharded-impl device" run sharded2000 NDCN_SELF_HALO=2000
