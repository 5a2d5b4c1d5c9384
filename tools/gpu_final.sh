#!/bin/bash
# Round-end measurement pass: GPU tests, smoke, bench, rocprofv3 kernel stats and HBM-traffic PMC passes.
set -u
R=${1:-r01}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$R" -o $R -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${R}_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass --steps 6 --warmup 1 > /dev/null 2>&1)
done
python - "$R" <<'PY'
import csv, glob, json, collections, sys
R = sys.argv[1]
raw = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_%s_%s/**/*counter_collection.csv' % (R, c), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    raw[c] = {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
kern = {}
for k, (n, f) in raw['FETCH_SIZE'].items():
    w = raw['WRITE_SIZE'].get(k, (0, 0.0))[1]
    kern[k] = {'launches': n, 'FETCH_SIZE_KB': round(f, 1), 'WRITE_SIZE_KB': round(w, 1),
               'hbm_bytes_per_launch': int((2 * f + w) * 1024)}
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 6 --warmup 1` "
        "(tools/gpu_final.sh); KB per launch. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports "
        "half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM); calibrated on combine_kernel (reads 2 panels "
        "= 2.048 GB, FETCH_SIZE = 1.024 GB).")
json.dump({'note': note, 'kernels': kern}, open('gpurun_out/%s_traffic_pmc.json' % R, 'w'), indent=1)
PY
grep -n "passed\|failed\|FAILED" gpurun_out/pytest_gpu.log | tail -5 | cut -c1-200
tail -2 gpurun_out/smoke.log
grep '^{"metric' gpurun_out/bench.log | cut -c1-400
head -12 gpurun_out/prof_$R/*kernel_stats.csv | cut -c1-200
cp gpurun_out/prof_$R/*kernel_stats.csv gpurun_out/${R}_bench_kernel_stats.csv
grep '^{"metric' gpurun_out/bench.log > gpurun_out/${R}_bench_line.json
