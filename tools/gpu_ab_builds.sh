#!/bin/bash
# A/B of builds of the library inside one box: default vs gpurun_in_*.so (bench + FETCH_SIZE of the fused launches)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  rm -rf /tmp/pmc_ab; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_ab -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass --steps 6 --warmup 1 > /dev/null 2>&1)
  python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_ab/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'rhs_fused2_kernel<([^>]*)>', r['Kernel_Name'])
        if m and r['Counter_Name'] == 'FETCH_SIZE': agg[m.group(1)].append(float(r['Counter_Value']))
print({k: round(2 * sum(v) / len(v) / 1e6, 2) for k, v in sorted(agg.items())}, 'GB read per launch (2 x FETCH_SIZE)')
PY
}
{
  echo "=== default"; run
  for f in gpurun_in_*.so; do cp $f ndcn_amd/libndcn_hip.so; echo "=== $f"; run; done
} > gpurun_out/exp_ab.log 2>&1
