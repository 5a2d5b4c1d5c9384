"""Generates the loop body of the SECOND-GENERATION column-sweep prototype (tools/micro/sweep_v2_body.inc, used by
tools/micro/sweep_lab.hip only; the shipped kernel is csrc/spmm_sweep.hip).  Measured: 16 fetches in flight through an LDS
ring are NOT faster than 8 through registers (0.197-0.204 vs 0.195-0.200 ms, profiles/r04i_sweep_lab.txt): the sweep is
not bound by the depth of its fetch queue.  Kept as the record of that experiment.

The wave's inner loop handles one CHUNK of 64 entries per iteration, fully unrolled: every step folds one entry and issues the
fetch of the entry 16 positions ahead, and everything that depends on the position inside the chunk is an immediate - the lane
the entry's words are read from (v_readlane), the slot of the 16-deep LDS ring the row lands in, the pair of registers the
row is read into.  Writing 64 such steps by hand would be 1 400 lines; this script writes them.

    python tools/gen_sweep_asm.py            # rewrites tools/micro/sweep_v2_body.inc
"""
import os

D = 16            # fetches in flight per wave = slots of the LDS ring (1 KiB each)
CHUNK = 64        # entries per chunk: one per lane of {v208, v209} (A), {v210, v211} (B), {v212, v213} (C)

T = ((200, 201, 202, 203), (204, 205, 206, 207))      # the two register quads rows are read into from LDS


def issue(pos, slot):
    """fetch of the entry at chunk position pos (>= 64: chunk B) into ring slot `slot` by LDS-DMA"""
    reg, lane = (208, pos) if pos < CHUNK else (210, pos - CHUNK)
    return ['v_readlane_b32 s18, v%d, %d' % (reg, lane),
            's_and_b32 s69, s18, 0xffffff',
            's_lshl_b32 s69, s69, 10',
            's_add_u32 s20, %[xlo], s69',
            's_addc_u32 s21, %[xhi], 0',
            's_add_u32 m0, s22, %d' % (slot * 1024),
            's_nop 0',
            'global_load_lds_dwordx4 %[voff], s[20:21]']


def step(i):
    out = []
    t_next, t_cur = T[(i + 1) & 1], T[i & 1]
    # the row of entry i + 1 has landed (all but the 14 youngest fetches are complete): start reading it out of the ring
    out += ['s_waitcnt vmcnt(%d)' % (D - 2),
            'ds_read_b128 v[%d:%d], v214 offset:%d' % (t_next[0], t_next[3], ((i + 1) % D) * 1024),
            's_waitcnt lgkmcnt(1)']                       # the read of entry i (issued one step ago) is complete
    out += ['v_readlane_b32 s16, v208, %d' % i,
            'v_readlane_b32 s17, v209, %d' % i,
            's_lshr_b32 s68, s16, 22',
            's_and_b32 s68, s68, 0x3fc',
            's_set_gpr_idx_on s68, gpr_idx(SRC2,DST)']
    out += ['v_fma_f32 v%d, s17, v%d, v%d' % (c, t_cur[c], c) for c in range(4)]
    out += ['s_set_gpr_idx_off']
    out += issue(i + D, i % D)
    return out


def crossing(i):
    """before the 16 fetches of positions i + 16 .. i + 31 are issued: has the sweep entered a later column block?"""
    pos = i + D
    reg, lane = (208, pos) if pos < CHUNK else (210, pos - CHUNK)
    L = 'X%d' % i
    out = ['v_readlane_b32 s18, v%d, %d' % (reg, lane),
           's_and_b32 s66, s18, 0xffffff',
           's_lshr_b32 s66, s66, %[logb]',
           's_cmp_le_u32 s66, s65',
           's_cbranch_scc1 L_nocross_%s_%%=' % L]
    if pos >= CHUNK:                                       # the last chunk's successor belongs to the next slab
        out += ['s_cmp_le_u32 s64, 1',
                's_cbranch_scc1 L_nocross_%s_%%=' % L]
    out += ['s_mov_b32 s65, s66',
            's_or_b32 s67, s66, %[etag]',
            'PUBLISH',
            's_cmp_lt_u32 s66, %[window]',
            's_cbranch_scc1 L_prefetch_%s_%%=' % L,
            's_cmp_lg_u32 s77, 0',
            's_cbranch_scc1 L_prefetch_%s_%%=' % L,
            's_sub_u32 s70, s67, %[wm1]',
            's_cmp_eq_u32 s76, 0',
            's_cbranch_scc1 L_slow_%s_%%=' % L,
            'BEHIND',
            's_cbranch_scc1 L_prefetch_%s_%%=' % L,
            'L_slow_%s_%%=:' % L,
            's_mov_b32 s71, 0',
            'L_spin_%s_%%=:' % L,
            'global_load_dwordx4 v[220:223], %[voff], %[prog] sc1',
            's_waitcnt vmcnt(0)',
            'BEHIND',
            's_cbranch_scc1 L_prefetch_%s_%%=' % L,
            's_sleep 4',
            's_add_u32 s71, s71, 1',
            's_cmp_lt_u32 s71, 200',
            's_cbranch_scc1 L_spin_%s_%%=' % L,
            's_mov_b32 s77, 1',
            'L_prefetch_%s_%%=:' % L,
            'global_load_dwordx4 v[220:223], %[voff], %[prog] sc1',
            's_mov_b32 s76, 1',
            'L_nocross_%s_%%=:' % L]
    return out


def body():
    out = []
    for i in range(CHUNK):
        if i % D == 0:
            out += crossing(i)
        out += step(i)
    return out


def prologue():
    out = []
    for i in range(D):
        out += issue(i, i)
    return out


def render(lines):
    res = []
    for l in lines:
        if l in ('PUBLISH', 'BEHIND'):
            res.append('    %s' % l)
        else:
            res.append('    "%s\\n"' % l)
    return '\n'.join(res)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'tools', 'micro', 'sweep_v2_body.inc')
    with open(path, 'w') as fh:
        fh.write('// GENERATED by tools/gen_sweep_asm.py - do not edit.  Register map: sweep_lab.hip (sweep_asm2_kernel).\n')
        fh.write('#define SWEEP_PROLOGUE \\\n' + render(prologue()).replace('\n', ' \\\n') + '\n\n')
        fh.write('#define SWEEP_CHUNK \\\n' + render(body()).replace('\n', ' \\\n') + '\n')
    print('wrote', path)


if __name__ == '__main__':
    main()
