#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp && NDCN_SELF_HALO=2000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psh -o x -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29717 $GRAFT_REPO_ROOT/bench.py --gpus 1 --sharded --steps 10 --warmup 2 --no-cpu-baseline --no-profile-pass > /tmp/psh.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/psh.log
f=$(find /tmp/psh -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%.1f us avg' % (float(r['AverageNs'])/1e3), '%.1f ms' % (float(r['TotalDurationNs'])/1e6))
PY
