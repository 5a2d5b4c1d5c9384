// Column-sweep SpMM lab: S = A X on a graph WITHOUT locality (BASELINE config 2: G(n,p), n = 1e5, mean degree 40, H = 256) with
// the partial sums of all rows register-resident and every XCD sweeping the columns of X in ascending order, so that the rows of
// X an XCD needs at one time form a window that stays in its 4 MiB L2.
//   wave (xcd, slot) owns RW consecutive rows ("slab"); its entries, merged over the slab's rows and sorted by column, are
//   streamed by scalar loads; each entry = {local row << 24 | column, value}: acc[row] = fma(value, X[column], acc[row]) - per row
//   the entries arrive in ascending column order, i.e. the fma chain of a sequential CSR loop (bit-identical results).
//   Soft synchronisation per XCD: a wave that has issued its last fetch of column block b counts it in done[xcd][b]; a wave
//   enters block b only when every wave of its XCD has left block b - S (bounded spin: a hint for locality, never needed for
//   correctness).
//   ./sweep_lab  -> table LOGB x S -> ms, against the one-row-per-wave gather on the same operator; results compared bit for bit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u16v __attribute__((ext_vector_type(16)));

constexpr int RW = 49, WAVES = 8, D = 8, SLOTS = 256;      // rows per wave, waves per workgroup, fetches in flight, waves per XCD

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void ref_kernel(const float *__restrict__ X, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  const float *__restrict__ val, float *__restrict__ Y, int n_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < n_rows; r += n_waves) {
        const int b = rowptr[r], e = rowptr[r + 1];
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = b; j < e; j += 8) {
            f4 v[8];
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u < e ? j + u : e - 1;
                a[u] = val[jj];
                v[u] = reinterpret_cast<const f4 *>(X + (size_t)col[jj] * 256)[lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j + u < e) {
                    acc.x = __builtin_fmaf(a[u], v[u].x, acc.x); acc.y = __builtin_fmaf(a[u], v[u].y, acc.y);
                    acc.z = __builtin_fmaf(a[u], v[u].z, acc.z); acc.w = __builtin_fmaf(a[u], v[u].w, acc.w);
                }
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<f4 *>(Y + (size_t)r * 256) + lane);
    }
}

// in-place updates (tied operands): the accumulators never move between registers, whatever branch of the switch ran
#define FMA1(ACC, XV) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ACC) : "s"(a), "v"(XV))
#define CASE(K) case K: FMA1(acc[K][0], x.x); FMA1(acc[K][1], x.y); FMA1(acc[K][2], x.z); FMA1(acc[K][3], x.w); break;

__device__ __forceinline__ void fold(float (&acc)[RW][4], int r, float a, const f4 &x) {
    switch (r) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19)
        CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
        CASE(30) CASE(31) CASE(32) CASE(33) CASE(34) CASE(35) CASE(36) CASE(37) CASE(38) CASE(39)
        CASE(40) CASE(41) CASE(42) CASE(43) CASE(44) CASE(45) CASE(46) CASE(47) CASE(48)
        default: break;
    }
}

// done[xcd * nblk + b] counts, over all launches, the waves of the XCD that have left column block b; target = launches * SLOTS.
template <int LOGB, int S, bool SINGLE>
__global__ __launch_bounds__(WAVES * 64) void sweep_kernel(const float *__restrict__ X, const u16v *__restrict__ ent, const int *__restrict__ slab_ptr,
                                                           float *__restrict__ Y, int n_rows, int rows_per_xcd, unsigned *done, unsigned target,
                                                           int nblk) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wv;
    const int slab = xcd * SLOTS + slot;
    const int e0 = slab_ptr[2 * slab], cnt = slab_ptr[2 * slab + 1];   // {first entry (a multiple of 8), entries}
    const u16v *p = ent + (e0 >> 3);
    unsigned *dn = done + xcd * nblk;
    const float *Xl = X + lane * 4;

    float acc[RW][4];
#pragma unroll
    for (int k = 0; k < RW; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
    f4 xb[D];
    int issue_blk = 0;

    auto cross = [&](int blk) {                                 // the wave leaves blocks issue_blk .. blk - 1 and enters blk
        if (lane == 0)
            for (int b = issue_blk; b < blk; ++b) __hip_atomic_fetch_add(dn + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (S < 1000 && blk >= S && blk < nblk) {
            int tries = 0;
            while (__hip_atomic_load(dn + blk - S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++tries < 256)
                __builtin_amdgcn_s_sleep(4);
        }
        issue_blk = blk;
    };
    auto issue = [&](unsigned key, f4 &dst) {
        const int c = key & 0xffffff;
        const int blk = c >> LOGB;
        if (blk != issue_blk) cross(blk);
        dst = *reinterpret_cast<const f4 *>(Xl + (size_t)c * 256);
    };

    u16v cur = p[0], nxt = p[1];
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < cnt) issue(cur[2 * s], xb[s]);
    for (int base = 0; base < cnt; base += D) {
        const u16v nn = p[(base >> 3) + 2];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int e = base + s;
            if (e < cnt) {
                const unsigned key = cur[2 * s];
                const float a = __builtin_bit_cast(float, cur[2 * s + 1]);
                fold(acc, SINGLE ? 0 : (int)(key >> 24), a, xb[s]);
                if (e + D < cnt) issue(nxt[2 * s], xb[s]);
            }
        }
        cur = nxt;
        nxt = nn;
    }
    cross(nblk);                                               // count the blocks behind the wave's last entry
    const int row0 = xcd * rows_per_xcd + slot * RW;
    const int row_end = min(n_rows, (xcd + 1) * rows_per_xcd);
#pragma unroll
    for (int k = 0; k < RW; ++k)
        if (row0 + k < row_end) __builtin_nontemporal_store(f4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]}, reinterpret_cast<f4 *>(Y + (size_t)(row0 + k) * 256) + lane);
}


// ---- the sweep with everything the compiler cannot express: the accumulator a fetched row is folded into is selected at RUN TIME
// (v[4 r .. 4 r + 3]) through the VGPR index mode (s_set_gpr_idx_on: M0-relative destination / src2), so the wave's whole loop is
// one asm statement on fixed physical registers:
//   v0..v199  accumulators (row r of the slab = v[4r..4r+3]; row 49 = dummy for the padding entries)   v200..v231  ring of 8 fetches
//   v232 prefetched poll value   v233 offset temp   v234 = 1   v236..v239 store staging
//   s[16:31] entries of the group being folded   s[32:47] next group (being fetched)   s[48:63] group after next (in flight)
//   s64 groups left  s65 column block the fetches are in  s66 / s67 / s70 / s78 temps  s68 register index  s69 row offset  s71 tries
//   s[72:73] entry pointer  s[74:75] saved exec  s76 block the prefetched poll is for  s77 "stop synchronising" (a wait timed out)
// Vector memory completes in order (loads, stores and atomics share vmcnt on gfx9): every slot waits for vmcnt(7); the
// atomics and polls issued in between only make those waits stricter.
#define V10(p) "v" #p "0", "v" #p "1", "v" #p "2", "v" #p "3", "v" #p "4", "v" #p "5", "v" #p "6", "v" #p "7", "v" #p "8", "v" #p "9"
#define S10(p) "s" #p "0", "s" #p "1", "s" #p "2", "s" #p "3", "s" #p "4", "s" #p "5", "s" #p "6", "s" #p "7", "s" #p "8", "s" #p "9"
#define FOLD(KEY, VAL, X0, X1, X2, X3)                       \
    "s_waitcnt vmcnt(7)\n"                                   \
    "s_lshr_b32 s68, s" #KEY ", 22\n"                        \
    "s_and_b32 s68, s68, 0x3fc\n"                            \
    "s_set_gpr_idx_on s68, gpr_idx(SRC2,DST)\n"              \
    "v_fma_f32 v0, s" #VAL ", v" #X0 ", v0\n"                \
    "v_fma_f32 v1, s" #VAL ", v" #X1 ", v1\n"                \
    "v_fma_f32 v2, s" #VAL ", v" #X2 ", v2\n"                \
    "v_fma_f32 v3, s" #VAL ", v" #X3 ", v3\n"                \
    "s_set_gpr_idx_off\n"
#define ISSUE(KEY, X0, X3)                                   \
    "s_and_b32 s69, s" #KEY ", 0xffffff\n"                   \
    "s_lshl_b32 s69, s69, 10\n"                              \
    "buffer_load_dwordx4 v[" #X0 ":" #X3 "], %[voff], %[rsx], s69 offen\n"
#define PUBLISH                 /* lane 0 stores the wave's tagged progress s67 into its slot of the XCD's progress line */ \
    "v_mov_b32 v233, s67\n"                                  \
    "s_mov_b64 s[74:75], exec\n"                             \
    "s_mov_b64 exec, 1\n"                                    \
    "global_store_dword v232, v233, %[prog]\n"               \
    "s_mov_b64 exec, s[74:75]\n"
#define BEHIND                  /* vcc = lanes holding a wave whose progress is below the threshold s70 */ \
    "v_min_u32 v234, v234, v235\n"                           \
    "v_min_u32 v236, v236, v237\n"                           \
    "v_min_u32 v234, v234, v236\n"                           \
    "v_cmp_gt_u32 vcc, s70, v234\n"                          \
    "s_cmp_eq_u64 vcc, 0\n"

template <int LOGB, int S>
__global__ __launch_bounds__(WAVES * 64) void sweep_asm_kernel(const float *__restrict__ X, const u16v *__restrict__ ent,
                                                               const int *__restrict__ slab_ptr, float *__restrict__ Y, int n_rows,
                                                               int rows_per_xcd, unsigned *prog_all, unsigned etag, int nblk, int *xcc_dbg) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wv;
    const int slab = xcd * SLOTS + slot;
    const int e0 = __builtin_amdgcn_readfirstlane(slab_ptr[2 * slab]), cnt = __builtin_amdgcn_readfirstlane(slab_ptr[2 * slab + 1]);
    const int ngrp = (cnt + 7) >> 3;
    const u16v *p = ent + (e0 >> 3);
    unsigned *prog = prog_all + xcd * SLOTS;          // one dword per wave of the XCD: (launch << 16) | column block it fetches from
    const int row0 = xcd * rows_per_xcd + slot * RW;
    const int row_end = min(n_rows, (xcd + 1) * rows_per_xcd);
    const int nvalid = __builtin_amdgcn_readfirstlane(max(0, min(RW, row_end - row0)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const unsigned long long xb_ = (unsigned long long)X, yb_ = (unsigned long long)(Y + (size_t)row0 * 256);
    const u4 rsx = {(unsigned)xb_, (unsigned)(xb_ >> 32) & 0xffffu, (unsigned)n_rows * 1024u, 0x00020000u};
    const u4 rsy = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)yb_), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(yb_ >> 32) & 0xffffu)),
                    (unsigned)nvalid * 1024u, 0x00020000u};
    const int voff = lane * 16;
    if (xcc_dbg && threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_dbg[blockIdx.x] = (int)id;
        reinterpret_cast<unsigned long long *>(xcc_dbg + 256)[blockIdx.x] = wall_clock64();
    }
    asm volatile(
        "s_mov_b64 s[72:73], %[ent]\n"
        "s_mov_b32 s64, %[ngrp]\n"
        "s_mov_b32 s65, 0\n"
        "s_mov_b32 s76, 0\n"
        "s_mov_b32 s77, 0\n"
        "v_mov_b32 v232, %[slot4]\n"
        "s_mov_b32 s67, %[etag]\n"
        PUBLISH
        "s_mov_b32 s68, 0\n"
        "L_zero_%=:\n"
        "s_set_gpr_idx_on s68, gpr_idx(DST)\n"
        "v_mov_b32 v0, 0\n"
        "s_set_gpr_idx_off\n"
        "s_add_u32 s68, s68, 1\n"
        "s_cmp_lt_u32 s68, 200\n"
        "s_cbranch_scc1 L_zero_%=\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cbranch_scc1 L_tail_%=\n"
        "s_load_dwordx16 s[16:31], s[72:73], 0x0\n"
        "s_load_dwordx16 s[32:47], s[72:73], 0x40\n"
        "s_waitcnt lgkmcnt(0)\n"
        ISSUE(16, 200, 203) ISSUE(18, 204, 207) ISSUE(20, 208, 211) ISSUE(22, 212, 215)
        ISSUE(24, 216, 219) ISSUE(26, 220, 223) ISSUE(28, 224, 227) ISSUE(30, 228, 231)
        "L_loop_%=:\n"
        "s_load_dwordx16 s[48:63], s[72:73], 0x80\n"
        "s_add_u32 s72, s72, 0x40\n"
        "s_addc_u32 s73, s73, 0\n"
        // does the group about to be fetched start in a later column block?
        "s_and_b32 s66, s32, 0xffffff\n"
        "s_lshr_b32 s66, s66, %[logb]\n"
        "s_cmp_le_u32 s66, s65\n"
        "s_cbranch_scc1 L_nocross_%=\n"
        "s_cmp_le_u32 s64, 1\n"                       // last group: s32.. belong to the next slab
        "s_cbranch_scc1 L_nocross_%=\n"
        "s_mov_b32 s65, s66\n"
        "s_or_b32 s67, s66, %[etag]\n"
        PUBLISH
        "s_cmp_lt_u32 s66, %[S]\n"
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_cmp_lg_u32 s77, 0\n"
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_sub_u32 s70, s67, %[S] - 1\n"             // every wave must fetch from block blk - S + 1 or later
        "s_cmp_eq_u32 s76, 0\n"
        "s_cbranch_scc1 L_slow_%=\n"
        BEHIND
        "s_cbranch_scc1 L_prefetch_%=\n"
        "L_slow_%=:\n"
        "s_mov_b32 s71, 0\n"
        "L_spin_%=:\n"
        "global_load_dwordx4 v[234:237], %[voff], %[prog] sc1\n"
        "s_waitcnt vmcnt(0)\n"
        BEHIND
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_sleep 4\n"
        "s_add_u32 s71, s71, 1\n"
        "s_cmp_lt_u32 s71, 200\n"
        "s_cbranch_scc1 L_spin_%=\n"
        "s_mov_b32 s77, 1\n"
        "L_prefetch_%=:\n"
        "global_load_dwordx4 v[234:237], %[voff], %[prog] sc1\n"
        "s_mov_b32 s76, 1\n"
        "L_nocross_%=:\n"
        FOLD(16, 17, 200, 201, 202, 203) ISSUE(32, 200, 203)
        FOLD(18, 19, 204, 205, 206, 207) ISSUE(34, 204, 207)
        FOLD(20, 21, 208, 209, 210, 211) ISSUE(36, 208, 211)
        FOLD(22, 23, 212, 213, 214, 215) ISSUE(38, 212, 215)
        FOLD(24, 25, 216, 217, 218, 219) ISSUE(40, 216, 219)
        FOLD(26, 27, 220, 221, 222, 223) ISSUE(42, 220, 223)
        FOLD(28, 29, 224, 225, 226, 227) ISSUE(44, 224, 227)
        FOLD(30, 31, 228, 229, 230, 231) ISSUE(46, 228, 231)
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 s[16:17], s[32:33]\n s_mov_b64 s[18:19], s[34:35]\n s_mov_b64 s[20:21], s[36:37]\n s_mov_b64 s[22:23], s[38:39]\n"
        "s_mov_b64 s[24:25], s[40:41]\n s_mov_b64 s[26:27], s[42:43]\n s_mov_b64 s[28:29], s[44:45]\n s_mov_b64 s[30:31], s[46:47]\n"
        "s_mov_b64 s[32:33], s[48:49]\n s_mov_b64 s[34:35], s[50:51]\n s_mov_b64 s[36:37], s[52:53]\n s_mov_b64 s[38:39], s[54:55]\n"
        "s_mov_b64 s[40:41], s[56:57]\n s_mov_b64 s[42:43], s[58:59]\n s_mov_b64 s[44:45], s[60:61]\n s_mov_b64 s[46:47], s[62:63]\n"
        "s_sub_u32 s64, s64, 1\n"
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 L_loop_%=\n"
        "L_tail_%=:\n"
        "s_waitcnt vmcnt(0)\n"
        "s_or_b32 s67, %[nblk], %[etag]\n"
        PUBLISH
        "L_store_%=:\n"
        "s_mov_b32 s68, 0\n"
        "s_mov_b32 s69, 0\n"
        "s_mov_b32 s64, %[nvalid]\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cbranch_scc1 L_done_%=\n"
        "L_st_%=:\n"
        "s_set_gpr_idx_on s68, gpr_idx(SRC0)\n"
        "v_mov_b32 v200, v0\n"
        "v_mov_b32 v201, v1\n"
        "v_mov_b32 v202, v2\n"
        "v_mov_b32 v203, v3\n"
        "s_set_gpr_idx_off\n"
        "buffer_store_dwordx4 v[200:203], %[voff], %[rsy], s69 offen nt\n"
        "s_add_u32 s68, s68, 4\n"
        "s_add_u32 s69, s69, 0x400\n"
        "s_sub_u32 s64, s64, 1\n"
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 L_st_%=\n"
        "L_done_%=:\n"
        "s_waitcnt vmcnt(0)\n"
        :
        : [voff] "v"(voff), [rsx] "s"(rsx), [rsy] "s"(rsy), [ent] "s"(p), [ngrp] "s"(ngrp), [prog] "s"(prog), [etag] "s"(etag), [nblk] "s"(nblk), [slot4] "s"(slot * 4),
          [nvalid] "s"(nvalid), [logb] "n"(LOGB), [S] "n"(S)
        : "memory", "vcc", "scc", "m0", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", V10(1), V10(2), V10(3), V10(4), V10(5), V10(6),
          V10(7), V10(8), V10(9), V10(10), V10(11), V10(12), V10(13), V10(14), V10(15), V10(16), V10(17), V10(18), V10(19), V10(20), V10(21),
          V10(22), "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "s16", "s17", "s18", "s19", S10(2), S10(3), S10(4), S10(5), S10(6), S10(7));
}


// ---- second generation: 16 fetches in flight per wave through an LDS ring (global_load_lds_dwordx4: no register data path),
// entries read 64 at a time by ONE vector load (lane i = entry i) and handed to the scalar unit by v_readlane with immediate
// lane numbers - the chunk loop is generated (tools/gen_sweep_asm.py -> tools/micro/sweep_v2_body.inc)
#undef PUBLISH
#undef BEHIND
#define PUBLISH                                              \
    "v_mov_b32 v217, s67\n"                                  \
    "s_mov_b64 s[74:75], exec\n"                             \
    "s_mov_b64 exec, 1\n"                                    \
    "global_store_dword v216, v217, %[prog]\n"               \
    "s_mov_b64 exec, s[74:75]\n"
#define BEHIND                                               \
    "v_min_u32 v220, v220, v221\n"                           \
    "v_min_u32 v222, v222, v223\n"                           \
    "v_min_u32 v220, v220, v222\n"                           \
    "v_cmp_gt_u32 vcc, s70, v220\n"                          \
    "s_cmp_eq_u64 vcc, 0\n"
#include "sweep_v2_body.inc"

template <int LOGB, int S>
__global__ __launch_bounds__(WAVES * 64) void sweep_asm2_kernel(const float *__restrict__ X, const unsigned *__restrict__ ent,
                                                                const int *__restrict__ slab_ptr, float *__restrict__ Y, int n_rows,
                                                                int rows_per_xcd, unsigned *prog_all, unsigned etag, int nblk) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wv;
    const int slab = xcd * SLOTS + slot;
    const int e0 = __builtin_amdgcn_readfirstlane(slab_ptr[2 * slab]), cnt = __builtin_amdgcn_readfirstlane(slab_ptr[2 * slab + 1]);
    const int nchunk = (cnt + 63) >> 6;
    const unsigned *p = ent + (size_t)e0 * 2;                 // e0 is a multiple of 64
    unsigned *prog = prog_all + xcd * SLOTS;
    const int row0 = xcd * rows_per_xcd + slot * RW;
    const int row_end = min(n_rows, (xcd + 1) * rows_per_xcd);
    const int nvalid = __builtin_amdgcn_readfirstlane(max(0, min(RW, row_end - row0)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const unsigned long long xb_ = (unsigned long long)X, yb_ = (unsigned long long)(Y + (size_t)(nvalid > 0 ? row0 : 0) * 256);
    const u4 rsy = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)yb_), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(yb_ >> 32) & 0xffffu)),
                    (unsigned)nvalid * 1024u, 0x00020000u};
    const int voff = lane * 16, v8 = lane * 8;
    const int wm1 = S - 1, window = S, logb = LOGB;
    asm volatile(
        "s_mov_b64 s[24:25], %[ent]\n"
        "s_mov_b32 s64, %[nchunk]\n"
        "s_mov_b32 s65, 0\n"
        "s_mov_b32 s76, 0\n"
        "s_mov_b32 s77, 0\n"
        "s_mov_b32 s22, %[ldsbase]\n"
        "v_add_u32 v214, s22, %[voff]\n"
        "v_mov_b32 v215, %[v8]\n"
        "v_mov_b32 v216, %[slot4]\n"
        "s_mov_b32 s67, %[etag]\n"
        PUBLISH
        "s_mov_b32 s68, 0\n"
        "L_zero_%=:\n"
        "s_set_gpr_idx_on s68, gpr_idx(DST)\n"
        "v_mov_b32 v0, 0\n"
        "s_set_gpr_idx_off\n"
        "s_add_u32 s68, s68, 1\n"
        "s_cmp_lt_u32 s68, 200\n"
        "s_cbranch_scc1 L_zero_%=\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cbranch_scc1 L_tail_%=\n"
        "global_load_dwordx2 v[208:209], v215, s[24:25]\n"
        "global_load_dwordx2 v[210:211], v215, s[24:25] offset:512\n"
        "global_load_dwordx2 v[212:213], v215, s[24:25] offset:1024\n"
        "s_waitcnt vmcnt(0)\n"
        SWEEP_PROLOGUE
        "s_waitcnt vmcnt(15)\n"
        "ds_read_b128 v[200:203], v214\n"
        "L_loop_%=:\n"
        SWEEP_CHUNK
        "v_mov_b32 v208, v210\n"
        "v_mov_b32 v209, v211\n"
        "v_mov_b32 v210, v212\n"
        "v_mov_b32 v211, v213\n"
        "s_add_u32 s24, s24, 512\n"
        "s_addc_u32 s25, s25, 0\n"
        "global_load_dwordx2 v[212:213], v215, s[24:25] offset:1024\n"
        "s_sub_u32 s64, s64, 1\n"
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 L_loop_%=\n"
        "L_tail_%=:\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_or_b32 s67, %[nblk], %[etag]\n"
        PUBLISH
        "s_mov_b32 s68, 0\n"
        "s_mov_b32 s69, 0\n"
        "s_mov_b32 s64, %[nvalid]\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cbranch_scc1 L_done_%=\n"
        "L_st_%=:\n"
        "s_set_gpr_idx_on s68, gpr_idx(SRC0)\n"
        "v_mov_b32 v200, v0\n"
        "v_mov_b32 v201, v1\n"
        "v_mov_b32 v202, v2\n"
        "v_mov_b32 v203, v3\n"
        "s_set_gpr_idx_off\n"
        "buffer_store_dwordx4 v[200:203], %[voff], %[rsy], s69 offen nt\n"
        "s_add_u32 s68, s68, 4\n"
        "s_add_u32 s69, s69, 0x400\n"
        "s_sub_u32 s64, s64, 1\n"
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 L_st_%=\n"
        "L_done_%=:\n"
        "s_waitcnt vmcnt(0)\n"
        :
        : [voff] "v"(voff), [v8] "v"(v8), [rsy] "s"(rsy), [ent] "s"(p), [nchunk] "s"(nchunk), [prog] "s"(prog), [etag] "s"(etag), [nblk] "s"(nblk),
          [slot4] "s"(slot * 4), [nvalid] "s"(nvalid), [logb] "s"(logb), [window] "s"(window), [wm1] "s"(wm1), [xlo] "s"((unsigned)xb_),
          [xhi] "s"((unsigned)(xb_ >> 32)), [ldsbase] "s"(wv * 16384)
        : "memory", "vcc", "scc", "m0", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", V10(1), V10(2), V10(3), V10(4), V10(5), V10(6),
          V10(7), V10(8), V10(9), V10(10), V10(11), V10(12), V10(13), V10(14), V10(15), V10(16), V10(17), V10(18), V10(19), V10(20), V10(21),
          "v220", "v221", "v222", "v223", "s16", "s17", "s18", "s19", S10(2), S10(3), S10(4), S10(5), S10(6), S10(7));
}


// ---- companion prefetcher (experiment; outcome NEGATIVE, profiles/r04i_sweep_lab.txt).  First form (every wave polls the progress
// line once per block): the poll sits behind the wave's own outstanding requests - in-order return makes every block cost a full
// drain, 4.4 us per block against the sweep's 2: the prefetcher TRAILS the sweep and its backlog competes with the next launch
// (0.35-0.73 ms).  Second form (below: a pacer wave that only polls and publishes through LDS, request waves that never wait):
// it keeps up - its workgroups end with the sweep - and the sweep takes 0.214-0.229 ms, i.e. nothing is gained over 0.197-0.204
// without it: the first-touch waits the all-hits run (0.142 ms) removes are not what a prefetcher removes - the lines still
// have to cross the fabric into the XCD's L2 at the same time as the hits are served.
// ---- companion prefetcher (experiment): the all-hits ceiling of the sweep is 0.142 ms (fold = 2048), the sweep itself takes
// 0.200: a fifth of its fetches are the FIRST touch of a row of X in the XCD and wait for the fabric, and vector memory returns in
// order.  A second kernel on its own stream - 4 workgroups of 4 waves per XCD, 16 registers, so that its waves fit beside the
// sweep's 2 x 248 on a SIMD - walks X in row order `ahead` blocks in front of the slowest wave of its XCD (it reads the same
// progress line) and touches every 128-byte line once: the sweep's own fetches then hit.
__global__ __launch_bounds__(256) void sweep_prefetch_kernel(const float *__restrict__ X, int n_rows, const unsigned *prog_all, unsigned etag,
                                                             int nblk, int logb, int ahead, unsigned *sink) {
    // second form: wave 0 of the workgroup is the PACER - it does nothing but poll the XCD's progress line and publish
    // {slowest wave's block, done} in LDS; waves 1..3 request lines and never wait on vector memory (the first form's poll sat
    // behind the wave's own outstanding requests: in-order return made every block cost a full drain)
    __shared__ volatile int s_lag, s_done;
    if (sink && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(sink)[2 * blockIdx.x] = wall_clock64();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3, groups = gridDim.x >> 3;
    const unsigned *prog = prog_all + xcd * SLOTS;
    if (threadIdx.x == 0) { s_lag = 0; s_done = 0; }
    __syncthreads();
    if (wave == 0) {
        int idle = 0, last = -1;
        for (;;) {
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned w = __hip_atomic_load(prog + lane * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tag = w >> 16, mine = etag >> 16;
                const unsigned blk = tag == mine ? (w & 0xffffu) : (((tag - mine) & 0xffffu) < 0x8000u ? (unsigned)nblk : 0u);   // older tag: not started
                m = min(m, blk);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o, 64));
            if (lane == 0) s_lag = (int)m;
            if ((int)m >= nblk) break;
            idle = ((int)m == last) ? idle + 1 : 0;
            last = (int)m;
            if (idle > 3000) break;                                   // nothing moves: the sweep is not running beside us
        }
        if (lane == 0) s_done = 1;
    } else {
        const int part = grp * 3 + (wave - 1), parts = groups * 3;
        int b = 0;
        while (b < nblk) {
            const int lag = s_lag;
            if (s_done) break;
            if (b < lag) b = lag;                                     // everybody is past these rows already
            if (b > lag + ahead) { __builtin_amdgcn_s_sleep(2); continue; }
            if (b >= nblk) break;
            const long row0 = (long)b << logb;
            const long rows = min(1l << logb, (long)n_rows - row0);
            const long lines = rows * 8;                              // 128-byte lines of the block
            const char *base = reinterpret_cast<const char *>(X) + row0 * 1024;
            for (long l0 = (long)part * 64; l0 < lines; l0 += (long)parts * 64) {
                const long l = l0 + lane;
                if (l < lines) {
                    unsigned tmp;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(tmp) : "v"(base + l * 128) : "memory");   // never waited for
                }
            }
            ++b;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (sink && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(sink)[2 * blockIdx.x + 1] = wall_clock64();
}

struct Dev {
    float *X, *Y, *Yref, *val;
    int *rowptr, *col, *slab_ptr;
    u16v *ent;
    unsigned *done;
    int n, nblk_max;
    int *xcc;
    unsigned long long *ts;
    unsigned epoch = 0;
};

template <int LOGB, int S, bool SINGLE = false>
static double run_sweep(Dev &d, int reps) {
    const int nblk = (d.n + (1 << LOGB) - 1) >> LOGB;
    const int rows_per_xcd = (d.n + 7) / 8;
    CK(hipMemset(d.done, 0, 8 * d.nblk_max * 4));
    d.epoch = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        ++d.epoch;
        hipLaunchKernelGGL((sweep_kernel<LOGB, S, SINGLE>), dim3(256), dim3(WAVES * 64), 0, 0, d.X, d.ent, d.slab_ptr, d.Y, d.n, rows_per_xcd, d.done,
                           d.epoch * SLOTS, nblk);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int LOGB, int S>
static double run_asm(Dev &d, int reps) {
    const int nblk = (d.n + (1 << LOGB) - 1) >> LOGB;
    const int rows_per_xcd = (d.n + 7) / 8;
    CK(hipMemset(d.done, 0, 8 * d.nblk_max * 4));
    d.epoch = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        ++d.epoch;
        hipLaunchKernelGGL((sweep_asm_kernel<LOGB, S>), dim3(256), dim3(WAVES * 64), 0, 0, d.X, d.ent, d.slab_ptr, d.Y, d.n, rows_per_xcd, d.done,
                           d.epoch << 16, nblk, d.xcc);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int LOGB, int S>
static double run_asm_pf(Dev &d, int reps, int ahead, int pf_groups) {
    const int nblk = (d.n + (1 << LOGB) - 1) >> LOGB;
    const int rows_per_xcd = (d.n + 7) / 8;
    CK(hipMemset(d.done, 0, 8 * d.nblk_max * 4));
    d.epoch = 0;
    static hipStream_t s1 = nullptr, s2 = nullptr;
    if (!s1) { CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
    hipEvent_t e0, e1, go;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreateWithFlags(&go, hipEventDisableTiming);
    auto launch = [&]() {
        ++d.epoch;
        hipEventRecord(go, s1);                                  // the prefetcher of launch i starts when launch i - 1 has finished
        hipStreamWaitEvent(s2, go, 0);
        hipLaunchKernelGGL(sweep_prefetch_kernel, dim3(8 * pf_groups), dim3(256), 0, s2, d.X, d.n, d.done, d.epoch << 16, nblk, LOGB, ahead,
                           (unsigned *)d.ts);
        hipLaunchKernelGGL((sweep_asm_kernel<LOGB, S>), dim3(256), dim3(WAVES * 64), 0, s1, d.X, d.ent, d.slab_ptr, d.Y, d.n, rows_per_xcd, d.done,
                           d.epoch << 16, nblk, d.xcc);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0, s1);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, s1);
    CK(hipEventSynchronize(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int LOGB, int S>
static double run_asm2(Dev &d, int reps) {
    const int nblk = (d.n + (1 << LOGB) - 1) >> LOGB;
    const int rows_per_xcd = (d.n + 7) / 8;
    CK(hipMemset(d.done, 0, 8 * d.nblk_max * 4));
    d.epoch = 0;
    static bool attr = false;
    CK(hipFuncSetAttribute((const void *)sweep_asm2_kernel<LOGB, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        ++d.epoch;
        hipLaunchKernelGGL((sweep_asm2_kernel<LOGB, S>), dim3(256), dim3(WAVES * 64), 128 * 1024, 0, d.X, (const unsigned *)d.ent, d.slab_ptr, d.Y, d.n,
                           rows_per_xcd, d.done, d.epoch << 16, nblk);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    (void)attr;
    return ms / reps;
}

static bool check(Dev &d, const char *what) {
    std::vector<float> a((size_t)d.n * 256), b((size_t)d.n * 256);
    CK(hipMemcpy(a.data(), d.Y, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d.Yref, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, bad_rows = 0;
    int shown = 0;
    for (int r = 0; r < d.n; ++r) {
        size_t rb = 0;
        for (int h = 0; h < 256; ++h) rb += memcmp(&a[(size_t)r * 256 + h], &b[(size_t)r * 256 + h], 4) != 0;
        bad += rb;
        bad_rows += rb != 0;
        if (rb && shown < 6) {
            ++shown;
            printf("    row %d: %zu of 256 differ; [0] got %.9g want %.9g  [1] %.9g / %.9g  [255] %.9g / %.9g\n", r, rb, a[(size_t)r * 256],
                   b[(size_t)r * 256], a[(size_t)r * 256 + 1], b[(size_t)r * 256 + 1], a[(size_t)r * 256 + 255], b[(size_t)r * 256 + 255]);
        }
    }
    printf("  %s: %zu of %zu elements (%zu rows) differ from the row gather%s\n", what, bad, a.size(), bad_rows,
           bad ? "  <-- MISMATCH" : " (bit-identical)");
    std::vector<unsigned> dn(8 * d.nblk_max);
    CK(hipMemcpy(dn.data(), d.done, dn.size() * 4, hipMemcpyDeviceToHost));
    printf("    progress words after %u launches, xcd 0 waves 0..7 (launch << 16 | blocks):", d.epoch);
    for (int i = 0; i < 8; ++i) printf(" %x", dn[i]);
    printf("\n");
    return bad == 0;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    const double deg = argc > 2 ? atof(argv[2]) : 40.0;
    // fold > 0: the sweep FETCHES row (column % fold) - every fetch an L2 hit once `fold` rows are resident: the ceiling of the
    // L2 -> CU path for this access pattern (results then differ from the row gather, which is the point of the exercise)
    const int fold = argc > 3 ? atoi(argv[3]) : 0;
    if ((n + 7) / 8 > SLOTS * RW) { printf("n too large for one pass\n"); return 1; }
    // G(n, p) by geometric skips, rows ascending, columns ascending
    std::vector<int> rowptr(n + 1, 0), col;
    std::vector<float> val;
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    const double p = deg / n, lq = log(1.0 - p);
    for (int r = 0; r < n; ++r) {
        long c = -1;
        for (;;) {
            c += 1 + (long)floor(log(1.0 - rnd()) / lq);
            if (c >= n) break;
            col.push_back((int)c);
            val.push_back((float)(rnd() - 0.5));
        }
        rowptr[r + 1] = (int)col.size();
    }
    const size_t nnz = col.size();
    printf("G(n,p): n = %d, nnz = %zu (mean degree %.1f), H = 256: X = %.0f MB\n", n, nnz, (double)nnz / n, n * 1024.0 / 1e6);
    // slabs
    const int rows_per_xcd = (n + 7) / 8;
    std::vector<int> slab_ptr(2 * 8 * SLOTS, 0);
    std::vector<unsigned> ent;
    struct E { unsigned c, r; float v; };
    std::vector<E> tmp;
    for (int x = 0; x < 8; ++x)
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int row0 = x * rows_per_xcd + sl * RW, row_end = std::min(n, (x + 1) * rows_per_xcd);
            tmp.clear();
            for (int k = 0; k < RW && row0 + k < row_end; ++k)
                for (int j = rowptr[row0 + k]; j < rowptr[row0 + k + 1]; ++j) tmp.push_back({(unsigned)col[j], (unsigned)k, val[j]});
            std::stable_sort(tmp.begin(), tmp.end(), [](const E &a, const E &b) { return a.c < b.c; });
            slab_ptr[2 * (x * SLOTS + sl)] = (int)(ent.size() / 2);
            slab_ptr[2 * (x * SLOTS + sl) + 1] = (int)tmp.size();
            for (const E &e : tmp) { ent.push_back(e.r << 24 | (fold > 0 ? e.c % (unsigned)fold : e.c)); unsigned u; memcpy(&u, &e.v, 4); ent.push_back(u); }
            while ((ent.size() / 2) % 64) { ent.push_back(49u << 24); ent.push_back(0); }   // padding: 0 * X[0] into the dummy row
        }
    for (int i = 0; i < 2 * 256; ++i) ent.push_back(0);             // the prefetch reads two groups past the end
    Dev d;
    d.n = n;
    d.nblk_max = (n >> 8) + 2;
    std::vector<float> hX((size_t)n * 256);
    for (auto &v : hX) v = (float)(rnd() - 0.5);
    CK(hipMalloc(&d.X, hX.size() * 4)); CK(hipMalloc(&d.Y, hX.size() * 4)); CK(hipMalloc(&d.Yref, hX.size() * 4));
    CK(hipMalloc(&d.val, nnz * 4)); CK(hipMalloc(&d.col, nnz * 4)); CK(hipMalloc(&d.rowptr, (n + 1) * 4));
    CK(hipMalloc(&d.slab_ptr, slab_ptr.size() * 4)); CK(hipMalloc(&d.ent, ent.size() * 4)); CK(hipMalloc(&d.done, 8 * d.nblk_max * 4)); CK(hipMalloc(&d.xcc, 256 * 4 + 256 * 8)); CK(hipMalloc(&d.ts, (256 + 128) * 8)); CK(hipMemset(d.ts, 0, (256 + 128) * 8));
    CK(hipMemcpy(d.X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.val, val.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d.col, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.rowptr, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.slab_ptr, slab_ptr.data(), slab_ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice));
    // the row gather: reference result + the time to beat
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(ref_kernel, dim3(1024), dim3(256), 0, 0, d.X, d.rowptr, d.col, d.val, d.Yref, n);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(ref_kernel, dim3(1024), dim3(256), 0, 0, d.X, d.rowptr, d.col, d.val, d.Yref, n);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("row gather (one row per wave, 8 fetches in flight, 16 waves per CU): %.3f ms\n", ms / 10);
    }
    printf("column sweep (asm, VGPR index mode), %d rows per wave, %d waves per CU, %d fetches in flight:\n", RW, WAVES, D);
    printf("  %-26s %10s %10s %10s %10s %10s %10s\n", "columns per block", "S = 2", "S = 3", "S = 4", "S = 6", "S = 8", "no sync");
    double t;
#define ROW(LOGB)                                                                                                        \
    printf("  %-26d", 1 << LOGB);                                                                                        \
    t = run_asm<LOGB, 2>(d, 10); printf(" %7.3f ms", t);                                                                 \
    t = run_asm<LOGB, 3>(d, 10); printf(" %7.3f ms", t);                                                                 \
    t = run_asm<LOGB, 4>(d, 10); printf(" %7.3f ms", t);                                                                 \
    t = run_asm<LOGB, 6>(d, 10); printf(" %7.3f ms", t);                                                                 \
    t = run_asm<LOGB, 8>(d, 10); printf(" %7.3f ms", t);                                                                 \
    t = run_asm<LOGB, 60000>(d, 10); printf(" %7.3f ms\n", t);
    ROW(8) ROW(9) ROW(10) ROW(11)
    // (argv[4] = 1: the companion-prefetcher experiment - slow to run: its workgroups spin with bounded waits)
    if (argc > 4 && atoi(argv[4]) == 1) {
    printf("with the companion prefetcher on a second stream (blocks ahead of the slowest wave x workgroups per XCD):\n");
    for (int ahead : {1, 2, 3})
        for (int grp : {2, 4}) {
            const double a = run_asm_pf<10, 3>(d, 10, ahead, grp), b2 = run_asm_pf<10, 2>(d, 10, ahead, grp), c2 = run_asm_pf<11, 2>(d, 10, ahead, grp);
            printf("  ahead %d, %d x 4 waves per XCD:  <1024, 3> %7.3f ms   <1024, 2> %7.3f ms   <2048, 2> %7.3f ms\n", ahead, grp, a, b2, c2);
        }
    run_asm_pf<10, 3>(d, 1, 3, 4);
    CK(hipDeviceSynchronize());
    check(d, "asm sweep<1024, 3> + prefetcher");
    {
        std::vector<unsigned long long> ts(64), sw(256);
        CK(hipMemcpy(ts.data(), d.ts, 64 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sw.data(), reinterpret_cast<char *>(d.xcc) + 256 * 4, 256 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 32; ++i) t0 = std::min(t0, ts[2 * i]);
        unsigned long long smin = ~0ull, smax = 0;
        for (int i = 0; i < 256; ++i) { smin = std::min(smin, sw[i]); smax = std::max(smax, sw[i]); }
        printf("  wall clock (100 MHz ticks) relative to the first prefetch workgroup: sweep workgroups start %+lld .. %+lld;", (long long)(smin - t0), (long long)(smax - t0));
        printf(" prefetch workgroups 0..3 end at");
        for (int i = 0; i < 4; ++i) printf(" %+lld%s", (long long)((ts[2 * i + 1] & ~(1ull << 63)) - t0), (ts[2 * i + 1] >> 63) ? "(gave up)" : "");
        int late = 0;
        for (int i = 0; i < 256; ++i) late += (sw[i] - smin) > 1000;
        printf("; %d sweep workgroups started > 10 us after the first\n", late);
    }
    }
    printf("second generation (LDS ring, 16 fetches in flight):\n");
    printf("  %-26s %10s %10s %10s %10s %10s %10s\n", "columns per block", "S = 2", "S = 3", "S = 4", "S = 6", "S = 8", "no sync");
#define ROW2(LOGB)                                                                                                       \
    printf("  %-26d", 1 << LOGB);                                                                                        \
    t = run_asm2<LOGB, 2>(d, 10); printf(" %7.3f ms", t);                                                                \
    t = run_asm2<LOGB, 3>(d, 10); printf(" %7.3f ms", t);                                                                \
    t = run_asm2<LOGB, 4>(d, 10); printf(" %7.3f ms", t);                                                                \
    t = run_asm2<LOGB, 6>(d, 10); printf(" %7.3f ms", t);                                                                \
    t = run_asm2<LOGB, 8>(d, 10); printf(" %7.3f ms", t);                                                                \
    t = run_asm2<LOGB, 60000>(d, 10); printf(" %7.3f ms\n", t);
    ROW2(9) ROW2(10) ROW2(11)
    run_asm2<10, 3>(d, 1);
    check(d, "asm2 sweep<1024, 3>");
    run_asm2<11, 2>(d, 1);
    check(d, "asm2 sweep<2048, 2>");
    printf("memory side only, compiler-generated, no sync (every entry folded into ONE accumulator; results meaningless):");
    t = run_sweep<11, 1000, true>(d, 10); printf(" %7.3f ms\n", t);
    run_asm<10, 2>(d, 1);
    check(d, "asm sweep<1024, 2>");
    run_asm<9, 3>(d, 1);
    check(d, "asm sweep<512, 3>");
    run_asm<10, 60000>(d, 1);
    check(d, "asm sweep<1024, no sync>");
    {
        std::vector<int> xc(256);
        CK(hipMemcpy(xc.data(), d.xcc, 256 * 4, hipMemcpyDeviceToHost));
        int ok = 0;
        for (int i = 0; i < 256; ++i) ok += xc[i] == (i & 7);
        printf("workgroup b ran on XCD b %% 8 for %d of 256 workgroups (first 16 XCC ids:", ok);
        for (int i = 0; i < 16; ++i) printf(" %d", xc[i]);
        printf(")\n");
    }
    return 0;
}
