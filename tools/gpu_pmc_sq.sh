#!/bin/bash
# SQ issue/stall counters of the fused RHS kernel (one --pmc group per pass; kernel-trace only)
export TMPDIR=/tmp
mkdir -p gpurun_out/sq
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > "$GRAFT_REPO_ROOT/gpurun_out/sq/avail.txt")
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
 "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT"
 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"
 "SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"
)
i=0
for g in "${groups[@]}"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/sq/g$i" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-profile-pass --steps 2 --warmup 0 > "$GRAFT_REPO_ROOT/gpurun_out/sq/g$i.log" 2>&1)
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sq/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rhs_fused2_kernel' not in k: continue
        import re
        m_ = re.search(r'rhs_fused2_kernel<([^>]*)>', k)
        mode = m_.group(1) if m_ else k[:60]
        out[mode][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/sq/summary.txt', 'w') as fh:
    for mode in sorted(out):
        fh.write('== %s\n' % mode)
        for c in sorted(out[mode]):
            v = out[mode][c]
            fh.write('  %-32s n=%3d mean=%.4g\n' % (c, len(v), sum(v) / len(v)))
PY
tail -5 gpurun_out/sq/g0.log
