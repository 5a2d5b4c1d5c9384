#!/bin/bash
# GPU idle time between kernels of the (C++ device-solver) bench: where the launches leave gaps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile-pass > /tmp/pg.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/pg.log
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-46:]) for r in rows))
# timed region: from the 4th-last... take events after the first 35% (setup + warm-up)
ev=ev[int(len(ev)*0.45):]
t0=ev[0][0]; tend=max(e[1] for e in ev)
busy=0; cur_s,cur_e=ev[0][0],ev[0][1]
gaps=collections.Counter(); gapn=collections.Counter(); prev=ev[0]
for e in ev[1:]:
    if e[0] > cur_e:
        g=e[0]-cur_e
        gaps[(prev[2], e[2])]+=g; gapn[(prev[2], e[2])]+=1
        busy+=cur_e-cur_s; cur_s,cur_e=e[0],e[1]
    else:
        cur_e=max(cur_e,e[1])
    if e[1]>=cur_e: prev=e
busy+=cur_e-cur_s
print('span ms %.1f busy ms %.1f idle ms %.1f (%.1f %%)' % ((tend-t0)/1e6, busy/1e6, (tend-t0-busy)/1e6, 100*(tend-t0-busy)/(tend-t0)))
for k,v in gaps.most_common(12): print('%7.1f us total %4d x (%.1f us each)  %s -> %s' % (v/1e3, gapn[k], v/1e3/gapn[k], k[0][-38:], k[1][-38:]))
PY
