"""Per-step kernel mix from a rocprofv3 --kernel-trace --stats run: python prof_summary.py <dir> <steps> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step %.3f, launches per step %.1f" % (tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-86s n/step %6.1f  avg %8.1f us  %5.1f %%" % (r["Name"][:86], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
