#!/bin/bash
# The one GPU-box runner (gpurun -- 'bash tools/gpu.sh <cmd> ...').  Everything it writes goes to gpurun_out/; the
# summaries worth keeping are copied to profiles/ by hand afterwards.
#
#   tests [pytest -k expr]       GPU test suite (or a selection)
#   final  TAG                   tests + smoke + bench (M) + rocprofv3 kernel stats + HBM-traffic PMC passes for M
#   configs TAG [cfg ...]        per configuration (default: M NC C2 C3 C4 C5): PMC passes FIRST (FETCH_SIZE / WRITE_SIZE, separate
#                                runs, kernel-trace only) -> profiles/TAG_traffic_pmc_<cfg>.json on the box, then the
#                                bench line (which picks that summary up as roofline.traffic) -> TAG_bench_configs.jsonl
#   ab "K=V ..." ["K=V ..."]     A/B of environment knobs inside ONE box: ms/step + rocprofv3 per-variant kernel averages
#   abbuild                      A/B of library builds inside one box: the default build vs every gpurun_in_*.so
#   abalt ROUNDS [bench flags]   the same, ALTERNATING (A B A B ...): ms/step + per-variant kernel averages per run
#   c2lab                        BASELINE config 2 with the sweep's S store / the dense stage's S read, MFMA, epilogue switched off (needs
#                                gpurun_in_f3timing.so = rhs_fused3.hip built with -DNDCN_F3_TIMING): what a fused tail could save
#   mlab                         the metric's case with parts of the lattice kernel switched off (same timing build)
#   timing                       fused3 cycle accounting (needs a -DNDCN_F3_TIMING build)
#   sq                           SQ issue / stall counters of the fused RHS kernels
#   power                        power / clock samples while the bench runs
#   sharded TAG                  multi-GPU code path on one GPU: tests + bench --sharded with the self-halo hook
#   train                        autograd tests + training-step timing
#   lab NAME [args]              hipcc tools/micro/NAME.hip on the box and run it
#   locality TAG [cfg:layout ..] where the gather kernels' bytes come from (default C2:none C2:degree C3:none C3:degree C4:none M:none):
#                                per variant TCC_HIT_sum / TCC_MISS_sum (L2 hit rate), FETCH_SIZE / WRITE_SIZE (fabric bytes,
#                                Infinity-Cache hits included) and the kernels' average time -> gpurun_out/TAG_locality.json
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD=${1:-tests}; shift || true

bench_args() {          # configuration name -> bench.py flags
  case "$1" in
    M) echo "";; NC) echo "--no-control";; M_euler) echo "--method euler";; M_rk4) echo "--method rk4";; *) echo "--config $1";;
  esac
}

pmc_summary() {         # $1 = TAG, $2 = cfg, $3 = output json ; reads gpurun_out/pmc_${TAG}_${cfg}_{FETCH,WRITE}_SIZE
python - "$1" "$2" "$3" <<'PY'
import csv, glob, json, collections, sys
tag, cfg, out = sys.argv[1:4]
raw = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_%s_%s_%s/**/*counter_collection.csv' % (tag, cfg, c), recursive=True)
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
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (kernel-trace only) over `bench.py %s --steps 6 --warmup 1` "
        "(tools/gpu.sh); KB per launch. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the "
        "bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM); calibrated on combine_kernel (reads 2 panels = 2.048 GB, "
        "FETCH_SIZE = 1.024 GB).  Infinity-Cache hits are counted, not excluded (same section)." % cfg)
json.dump({'note': note, 'config': cfg, 'kernels': kern}, open(out, 'w'), indent=1)
print('pmc', cfg, {k[-44:]: v['hbm_bytes_per_launch'] for k, v in kern.items() if v['hbm_bytes_per_launch'] > 5e7})
PY
}

pmc_passes() {          # $1 = TAG, $2 = cfg
  local a ctr; a=$(bench_args "$2")
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${1}_${2}_$ctr" -o p -- \
       python "$GRAFT_REPO_ROOT/bench.py" $a --no-cpu-baseline --no-profile-pass --steps 6 --warmup 1 > /dev/null 2>&1)
  done
  pmc_summary "$1" "$2" "profiles/${1}_traffic_pmc_${2}.json"
  cp "profiles/${1}_traffic_pmc_${2}.json" gpurun_out/
}

kernel_stats() {        # $1 = label, rest = bench flags ; prints per-variant averages of the RHS kernels
  local lab=$1; shift
  (cd /tmp && rm -rf /tmp/p_$lab && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o x -- \
     python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass "$@" > /tmp/p_$lab.log 2>&1)
  grep -o '"ms_per_step": [0-9.]*' /tmp/p_$lab.log
  local f; f=$(find /tmp/p_$lab -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "gpurun_out/${lab}_kernel_stats.csv" && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) >= 0.5:
        print('   %-46s n=%4s avg %.3f ms  %5.1f %%' % (r['Name'].split('(')[0][-46:], r['Calls'], float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
}

case "$CMD" in
tests)
  if [ -n "${1:-}" ]; then K=(-k "$1"); else K=(); fi
  timeout 1500 python -m pytest tests -m gpu -x -q "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
  ;;
final)
  R=${1:-r03}
  timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  pmc_passes "$R" M
  timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
  kernel_stats "${R}_bench"
  grep -n "passed\|failed\|FAILED" gpurun_out/pytest_gpu.log | tail -5 | cut -c1-200
  tail -2 gpurun_out/smoke.log
  grep '^{"metric' gpurun_out/bench.log | cut -c1-400
  grep '^{"metric' gpurun_out/bench.log > gpurun_out/${R}_bench_line.json
  ;;
configs)
  R=${1:-r03}; shift || true
  CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=(M NC C2 C3 C4 C5)
  : > gpurun_out/${R}_bench_configs.jsonl
  for c in "${CFGS[@]}"; do
    [ -n "${NO_PMC:-}" ] || pmc_passes "$R" "$c"
    timeout 1200 python bench.py $(bench_args "$c") ${BENCH_EXTRA:-} 2> gpurun_out/bench_$c.err | grep '^{"metric' >> gpurun_out/${R}_bench_configs.jsonl
    tail -2 gpurun_out/bench_$c.err | cut -c1-300
  done
  python - "$R" <<'PY'
import json, sys
for l in open('gpurun_out/%s_bench_configs.jsonl' % sys.argv[1]):
    d = json.loads(l)
    r = d.get('roofline') or {}
    c = d.get('cpu_baseline') or {}
    print(d['config']['workload'][:50], '| ms/step', d['ms_per_step'], '| value %.3g' % d['value'], '| roof', r.get('kernel'), r.get('avg_ms'),
          r.get('frac'), 'traffic', r.get('traffic'), 'x%s' % r.get('traffic_over_algorithmic'), '| spmm', (d.get('spmm_standalone') or {}).get('frac_of_hbm_peak'),
          '| cpu %.3g' % c.get('value', 0))
PY
  ;;
ab)
  i=0
  for kv in "$@"; do i=$((i+1)); echo "=== $kv"; (export $kv; kernel_stats "ab_v$i"); done
  ;;
abbuild)
  run() {
    timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | head -3   # (a variant build may hang: bounded)
    kernel_stats "$1"
  }
  { echo "=== default"; run build_default
    cp ndcn_amd/libndcn_hip.so /tmp/default.so
    for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp "$f" ndcn_amd/libndcn_hip.so; echo "=== $f"; run "build_$(basename $f .so)"; done
    cp /tmp/default.so ndcn_amd/libndcn_hip.so
  } 2>&1 | tee gpurun_out/exp_ab.log
  ;;
abalt)
  # abalt ROUNDS [bench flags]: ALTERNATING runs of the default build and every gpurun_in_*.so inside one box (A B A B ...), per run
  # ms/step and the rocprofv3 per-variant kernel averages -> gpurun_out/exp_abalt.log.  Box class: the <1,4> launch's average
  # (1.52 ms fast class / 1.73 ms slow class).
  ROUNDS=${1:-3}; shift || true
  cp ndcn_amd/libndcn_hip.so /tmp/default.so
  { for r in $(seq 1 $ROUNDS); do
      cp /tmp/default.so ndcn_amd/libndcn_hip.so; echo "=== round $r default"; kernel_stats "abalt_default_$r" "$@"
      for f in gpurun_in_*.so; do [ -f "$f" ] || continue; cp "$f" ndcn_amd/libndcn_hip.so; echo "=== round $r $f"; kernel_stats "abalt_$(basename $f .so)_$r" "$@"; done
    done
    cp /tmp/default.so ndcn_amd/libndcn_hip.so
  } 2>&1 | tee gpurun_out/exp_abalt.log
  ;;
c2lab)
  # what fusing the dense stage into the sweep's tail could save on BASELINE config 2: the sweep without its S store, the dense stage
  # without the HBM side of its S read (needs gpurun_in_f3timing.so: rhs_fused3.hip built with -DNDCN_F3_TIMING), both
  cp ndcn_amd/libndcn_hip.so /tmp/default.so
  { echo "=== default";                          kernel_stats c2lab_default --config C2
    echo "=== sweep without the S store";        NDCN_SWEEP_DBG=1 kernel_stats c2lab_nostore --config C2
    cp gpurun_in_f3timing.so ndcn_amd/libndcn_hip.so
    echo "=== timing build, no switch";          kernel_stats c2lab_tb --config C2
    echo "=== dense stage: S rows from a 1 MiB window"; NDCN_FUSED3_DBG=8 kernel_stats c2lab_swin --config C2
    echo "=== dense stage: no MFMA";             NDCN_FUSED3_DBG=1 kernel_stats c2lab_nomfma --config C2
    echo "=== dense stage: no epilogue";         NDCN_FUSED3_DBG=4 kernel_stats c2lab_noepi --config C2
    echo "=== dense stage: window + no epilogue"; NDCN_FUSED3_DBG=12 kernel_stats c2lab_swin_noepi --config C2
    cp /tmp/default.so ndcn_amd/libndcn_hip.so
  } 2>&1 | tee gpurun_out/exp_c2lab.log
  ;;
mlab)
  # M dopri5 with parts of the lattice kernel switched off (needs gpurun_in_f3timing.so: rhs_fused3.hip built with -DNDCN_F3_TIMING;
  # results wrong, timing only): where the power-capped step's time goes - rocprofv3 per-variant averages per switch
  cp ndcn_amd/libndcn_hip.so /tmp/default.so
  { echo "=== default build";                    kernel_stats mlab_default
    cp gpurun_in_f3timing.so ndcn_amd/libndcn_hip.so
    echo "=== timing build, no switch";          kernel_stats mlab_tb
    echo "=== no MFMA (1)";                      NDCN_FUSED3_DBG=1 kernel_stats mlab_nomfma
    echo "=== no weight refills (64)";           NDCN_FUSED3_DBG=64 kernel_stats mlab_norefill
    echo "=== no MFMA, no refills (65)";         NDCN_FUSED3_DBG=65 kernel_stats mlab_nomfma_norefill
    echo "=== no fold (2)";                      NDCN_FUSED3_DBG=2 kernel_stats mlab_nofold
    echo "=== no epilogue (4)";                  NDCN_FUSED3_DBG=4 kernel_stats mlab_noepi
    cp /tmp/default.so ndcn_amd/libndcn_hip.so
  } 2>&1 | tee gpurun_out/exp_mlab.log
  ;;
gaps)
  # timeline of one bench run: idle time between consecutive kernels (host round trips of the adaptive controller)
  (cd /tmp && rm -rf /tmp/p_gaps && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gaps -o x -- \
     python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass "$@" > /tmp/p_gaps.log 2>&1)
  grep -o '"ms_per_step": [0-9.]*' /tmp/p_gaps.log
  f=$(find /tmp/p_gaps -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY' | tee gpurun_out/gaps.txt
import csv, sys, collections
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ndcn::', '')[:44]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the timed region: the last 20 * 6 fused launches and what lies between them
idx = [i for i, r in enumerate(rows) if 'rhs_fused3_kernel<false, 2, 1>' in r[2] or 'rhs_fused3_kernel<false, 2, 5>' in r[2]]
lo = idx[-21] + 1 if len(idx) > 21 else 0
sel = rows[lo:idx[-1] + 1]
busy = sum(e - s for s, e, _ in sel)
span = sel[-1][1] - sel[0][0]
print('timed-region kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)' % (len(sel), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
gap_after = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(sel, sel[1:]):
    gap_after[n0 + ' -> ' + n1].append(s1 - e0)
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1])):
    print('  %-96s n=%3d  avg gap %7.1f us  total %.3f ms' % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
print('one step:')
for s, e, n in sel[:14]:
    print('   %-46s start %+9.1f us  dur %7.1f us' % (n, (s - sel[0][0]) / 1e3, (e - s) / 1e3))
PY
  ;;
timing)
  for d in ${DBG_LIST:-0 6 70 1}; do
    echo "== NDCN_FUSED3_DBG=$d"
    NDCN_FUSED3_DBG=$d NDCN_FUSED3_TIMING=${NT:-3} python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "fused3 timing" | cut -c1-220
  done
  ;;
sq)
  mkdir -p gpurun_out/sq
  groups=(
   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
   "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
   "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY"
   "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
   "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH"
   "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT"
   "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
  )
  i=0
  for g in "${groups[@]}"; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/sq/g$i" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass --steps 2 --warmup 0 > "$GRAFT_REPO_ROOT/gpurun_out/sq/g$i.log" 2>&1)
    i=$((i+1))
  done
  python - <<'PY'
import csv, glob, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sq/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m_ = re.search(r'(rhs_fused\d_kernel<[^>]*>)', r['Kernel_Name'])
        if m_:
            out[m_.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/sq/summary.txt', 'w') as fh:
    for mode in sorted(out):
        fh.write('== %s\n' % mode)
        for c in sorted(out[mode]):
            v = out[mode][c]
            fh.write('  %-32s n=%3d mean=%.4g\n' % (c, len(v), sum(v) / len(v)))
print(open('gpurun_out/sq/summary.txt').read()[:6000])
PY
  ;;
sqk)
  # sqk 'KERNEL_REGEX' script.py [args]: the SQ issue / stall counter groups of `sq` for the kernels matching the regex, under any script
  RX=$1; shift
  rm -rf gpurun_out/sqk; mkdir -p gpurun_out/sqk
  groups=(
   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
   "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
   "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY"
   "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
   "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH"
   "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT"
   "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
   "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
   "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"
  )
  i=0
  for g in "${groups[@]}"; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/sqk/g$i" -o p -- python "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$GRAFT_REPO_ROOT/gpurun_out/sqk/g$i.log" 2>&1)
    i=$((i+1))
  done
  python - "$RX" <<'PY'
import csv, glob, collections, re, sys
rx = re.compile(sys.argv[1])
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sqk/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m_ = rx.search(r['Kernel_Name'])
        if m_:
            out[m_.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/sqk/summary.txt', 'w') as fh:
    for mode in sorted(out):
        fh.write('== %s\n' % mode)
        for c in sorted(out[mode]):
            v = out[mode][c]
            fh.write('  %-32s n=%3d mean=%.4g\n' % (c, len(v), sum(v) / len(v)))
print(open('gpurun_out/sqk/summary.txt').read()[:8000])
PY
  ;;
power)
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showuse --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|Max Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/power_probe.log 2>&1 &
  SMI=$!
  sleep 2
  python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-profile-pass 2>&1 | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/power_bench.log
  wait $SMI
  tail -5 gpurun_out/power_probe.log; cat gpurun_out/power_bench.log
  ;;
sharded)
  R=${1:-r03}
  timeout 1200 python -m pytest tests -m gpu -x -q -k "sharded or two_rank or bench_two or halo" > gpurun_out/pytest_sharded.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sharded.log
  tail -6 gpurun_out/pytest_sharded.log | cut -c1-300
  for spec in "M 0" "M 2000" "C4 0" "C4 scatter:200000"; do
    set -- $spec
    NDCN_SELF_HALO=$2 timeout 900 python bench.py --gpus 1 --sharded --sharded-impl device --config $1 --steps 10 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_sharded_$1_${2/:/_}.err | grep '^{"metric' > gpurun_out/${R}_bench_sharded_$1_${2/:/_}.json
    python -c "import json; d=json.load(open('gpurun_out/${R}_bench_sharded_$1_${2/:/_}.json')); print('$1 self_halo $2', d['ms_per_step'], d['halo_exchange'])"
    tail -2 gpurun_out/bench_sharded_$1_${2/:/_}.err | cut -c1-200
  done
  ;;
train)
  timeout 1200 python -m pytest tests/test_gpu_autograd.py -m gpu -x -q > gpurun_out/pytest_autograd.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_autograd.log
  tail -15 gpurun_out/pytest_autograd.log | cut -c1-300
  timeout 900 python tools/bench_train.py > gpurun_out/train.log 2>&1
  cut -c1-400 gpurun_out/train.log
  ;;
locality)
  R=${1:-r04}; shift || true
  VARS=("$@"); [ ${#VARS[@]} -eq 0 ] && VARS=(C2:none C2:degree C3:none C3:degree C4:none M:none)
  for v in "${VARS[@]}"; do
    cfg=${v%%:*}; lay=${v##*:}
    a=$(bench_args "$cfg"); [ "$lay" != none ] && a="$a --layout $lay"
    for ctr in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE WRITE_SIZE; do
      tag=${ctr%% *}
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/loc_${R}_${cfg}_${lay}_$tag" -o p -- \
         python "$GRAFT_REPO_ROOT/bench.py" $a --no-cpu-baseline --no-profile-pass --steps 4 --warmup 1 > /dev/null 2>&1)
    done
  done
  python - "$R" "${VARS[@]}" <<'PY'
import csv, glob, json, collections, sys, re
tag, variants = sys.argv[1], sys.argv[2:]
out = {'note': 'rocprofv3 --kernel-trace --pmc, one counter group per pass over `bench.py <cfg> [--layout L] --steps 4 --warmup 1` (tools/gpu.sh locality). '
               'l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); fabric_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950: FETCH_SIZE '
               'tallies 128-byte requests at 64 B, MI355X_MICROARCH.md; Infinity-Cache hits are included); avg_us from the kernel trace of the same pass.',
       'variants': {}}
for v in variants:
    cfg, lay = v.split(':')
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for grp in ('TCC_HIT_sum', 'FETCH_SIZE', 'WRITE_SIZE'):
        base = 'gpurun_out/loc_%s_%s_%s_%s' % (tag, cfg, lay, grp)
        for f in glob.glob(base + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ndcn::', '')
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if grp == 'TCC_HIT_sum':
            for f in glob.glob(base + '/**/*kernel_trace.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ndcn::', '')
                    dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    rows = {}
    for k, c in agg.items():
        if not re.search(r'rhs_fused|spmm_', k):
            continue
        m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else 0.0
        hit, miss = m('TCC_HIT_sum'), m('TCC_MISS_sum')
        fab = (2 * m('FETCH_SIZE') + m('WRITE_SIZE')) * 1024
        us = sum(dur[k]) / len(dur[k]) / 1e3 if dur.get(k) else 0.0
        rows[k] = {'launches': len(c.get('TCC_HIT_sum', [])), 'avg_us': round(us, 1), 'l2_hit_rate': round(hit / (hit + miss), 4) if hit + miss else None,
                   'l2_requests': int(hit + miss), 'fabric_bytes': int(fab), 'fabric_GBps': round(fab / us / 1e3, 1) if us else None}
    out['variants'][v] = rows
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches'])[:6]:
        print('%-12s %-44s n=%3d %8.1f us  L2 hit %s  fabric %.3f GB  %s GB/s' % (v, k[:44], r['launches'], r['avg_us'], r['l2_hit_rate'], r['fabric_bytes'] / 1e9, r['fabric_GBps']))
json.dump(out, open('gpurun_out/%s_locality.json' % tag, 'w'), indent=1)
PY
  ;;
lab)
  N=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/$N.hip -o /tmp/$N && timeout 600 /tmp/$N "$@" > gpurun_out/$N.log 2>&1
  echo "exit $?" >> gpurun_out/$N.log
  cut -c1-330 gpurun_out/$N.log
  ;;
*) echo "unknown command $CMD"; exit 2;;
esac
