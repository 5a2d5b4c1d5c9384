"""Where does the GPU idle?  Reads a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv), takes the LAST `--frac` of the run (the timed
steps), and lists the largest gaps between consecutive kernels with the kernels on either side, plus busy / idle totals.

    rocprofv3 --kernel-trace -d gpurun_out/tr -o t -- python tools/micro/one_train_step.py ; python tools/micro/trace_gaps.py gpurun_out/tr
"""
import csv, glob, sys, os
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    if s > end:
        gaps.append((s - end, rows[i - 1][2], n))
    end = max(end, e)
gaps.sort(reverse=True)
print('kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
from collections import Counter
by = Counter()
for g, a, b in gaps:
    by[(a, b)] += g
print('idle by (kernel before -> kernel after), top 15:')
for (a, b), g in by.most_common(15):
    print('  %8.3f ms   %s  ->  %s' % (g / 1e6, a, b))
