#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cd /tmp && NDCN_SELF_HALO=2000 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/psg -o x -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29719 $GRAFT_REPO_ROOT/bench.py --gpus 1 --sharded --steps 10 --warmup 2 --no-cpu-baseline --no-profile-pass > /tmp/psg.log 2>&1
f=$(find /tmp/psg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-46:], r.get('Stream_Id','?')) for r in rows))
# take the last 40% of the trace (timed region), compute gaps of the union busy time
ev=ev[int(len(ev)*0.55):]
t0=ev[0][0]; tend=max(e[1] for e in ev)
busy=0; cur_s,cur_e=ev[0][0],ev[0][1]
gaps=collections.Counter(); gapn=collections.Counter()
prev=ev[0]
for e in ev[1:]:
    if e[0] > cur_e:
        g=e[0]-cur_e
        gaps[(prev_name:=prev[2], e[2])]+=g; gapn[(prev[2], e[2])]+=1
        busy+=cur_e-cur_s; cur_s,cur_e=e[0],e[1]
    else:
        cur_e=max(cur_e,e[1])
    if e[1]>=cur_e: prev=e
busy+=cur_e-cur_s
print('span ms %.1f busy ms %.1f idle ms %.1f' % ((tend-t0)/1e6, busy/1e6, (tend-t0-busy)/1e6))
for k,v in gaps.most_common(14): print('%7.1f us total %4d x  %s -> %s' % (v/1e3, gapn[k], k[0][-40:], k[1][-40:]))
PY
