// Column-sweep SpMM: S = A X (H = 256) for operators WITHOUT locality whose rows are long relative to their count - the
// reference's G(n,p) graph at BASELINE config 2 (heat_dynamics.py:89; 10^5 nodes, mean degree 40) - replacing
// torch.sparse.mm(A, x) (neural_dynamics.py:29) there.
//
// Why.  A row gather fetches nnz rows of X through the fabric: the 102 MB panel lives in the Infinity Cache, an XCD's 4 MiB
// L2 holds 4 % of it, the hit rate is 6 % (profiles/r04c_locality.json) and the launch runs at the fabric's rate for the
// pattern (0.56 ms; tools/micro/gather_lab.hip: 0.53 ms with everything but the fetches stripped).  The only lever is fewer
// fabric bytes: if the partial sums of ALL rows stay resident while every XCD walks the COLUMNS of X in ascending order,
// the rows of X an XCD needs at one time form a narrow window that stays in its L2, and 8 * n_cols rows cross the fabric
// instead of nnz (5 x fewer here).  The partial sums of 100 000 rows are 102 MB - exactly what the chip's register files
// hold (256 CUs x 512 KiB): each of the 8 x 256 waves of a pass keeps up to 49 rows in 196 VGPRs.
//
// How.  Wave (xcd, slot) owns a "slab" of consecutive rows.  ndcn_csr_create (csr_plan.hip: build_sweep_plan) merged the
// slab's entries over its rows and sorted them by column: a stream of {row within the slab << 24 | column, value} pairs the
// wave reads by scalar loads, 8 entries per s_load_dwordx16, two groups ahead.  Per entry: one 1 KiB buffer load of X[column]
// (8 in flight per wave, 64 KiB per CU), and acc[row] = fma(value, X[column], acc[row]) - per ROW the entries arrive in
// ascending column order, i.e. the fma chain of a sequential CSR loop: the result is bit-identical to it.
// The accumulator is selected at RUN TIME: v[4 r .. 4 r + 3] through the VGPR index mode (s_set_gpr_idx_on: M0-relative
// destination and src2) - no compiler expresses that (a 49-way switch costs 196 phi copies per entry: tools/micro/
// sweep_lab.hip history), so the wave's whole loop is ONE asm statement on fixed physical registers.
// Locality is kept by a soft per-XCD synchronisation: every wave publishes (launch tag << 16 | column block it fetches
// from) in its own word of the XCD's 1 KiB progress line (a plain store: no atomics - 256 waves adding to one counter
// serialise in the L2, 21 us per block); before it enters block b it looks at the line (ONE 1 KiB load, prefetched at the
// previous crossing, so the common case costs no round trip) and waits - bounded - until no wave of the XCD is more than
// `window` blocks behind.  A hint for speed only: a wait that times out switches the wave's synchronisation off.
// Vector memory completes in order on gfx9 (loads, stores and atomics share vmcnt): every slot waits for vmcnt(7); the
// progress stores and polls issued in between only make those waits stricter.
//
// Measured (MI355X, tools/micro/sweep_lab.hip, profiles/r04i_sweep_lab.txt): 0.195-0.200 ms against 0.559 ms for the row
// gather on the same operator (blocks of 1024-2048 columns, window 2-3), 0.34 ms without the synchronisation.
#include "kernels.h"

namespace ndcn {

// (not in an anonymous namespace: rocprofv3 summaries key on the kernel name up to its first parenthesis)
constexpr int kSweepRows = 49;       // rows per wave (196 accumulator registers); row 49 = the dummy the padding entries add to
constexpr int kSweepWaves = 8;       // waves per workgroup = per CU (256 VGPRs each)
constexpr int kSweepSlots = 256;     // waves per XCD

typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct SweepArgs {
    const float *X;
    const u32x16 *ent;          // groups of 8 entries
    const int32_t *slab;        // this pass: [2048][2]
    float *Y;
    uint32_t *prog;             // this pass: [8][256]
    unsigned x_bytes;
    int row_base, row_end;      // the pass's rows
    int rows_per_xcd, rpw;
    uint32_t *epoch;            // this pass's launch counter (device memory: a hipGraph replay must see a new tag too)
    int nblk, logb, window;
    int dbg;                    // NDCN_SWEEP_DBG (timing experiments, results wrong): 1 = the finished rows are not stored
};

// register map inside the asm (all clobbered):
//   v0..v199  accumulators (row r of the slab = v[4r..4r+3]; row 49 = dummy)   v200..v231  ring of 8 fetched rows
//   v232 = 4 * slot (offset of the wave's progress word)   v233 temp   v236..v239 store staging   v244..v247 prefetched progress line
//   s[16:31] entries being folded   s[32:47] next group (being fetched)   s[48:63] group after next (in flight)
//   s64 groups left / rows left   s65 column block the fetches are in   s66 s67 s70 temps   s68 register index   s69 row offset
//   s71 tries   s[72:73] entry pointer   s[74:75] saved exec   s76 a progress line was prefetched   s77 synchronisation off
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
#define PUBLISH                 /* lane 0 stores the wave's tagged progress s67 into its word of the XCD's progress line */ \
    "v_mov_b32 v233, s67\n"                                  \
    "s_mov_b64 s[74:75], exec\n"                             \
    "s_mov_b64 exec, 1\n"                                    \
    "global_store_dword v232, v233, %[prog]\n"               \
    "s_mov_b64 exec, s[74:75]\n"
#define BEHIND                  /* scc = no lane holds a wave whose progress is below the threshold s70 */ \
    "v_min_u32 v244, v244, v245\n"                           \
    "v_min_u32 v246, v246, v247\n"                           \
    "v_min_u32 v244, v244, v246\n"                           \
    "v_cmp_gt_u32 vcc, s70, v244\n"                          \
    "s_cmp_eq_u64 vcc, 0\n"

__global__ __launch_bounds__(kSweepWaves * 64) void spmm_sweep_kernel(SweepArgs a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * kSweepWaves + wv;
    const int slab = xcd * kSweepSlots + slot;
    const int e0 = __builtin_amdgcn_readfirstlane(a.slab[2 * slab]), cnt = __builtin_amdgcn_readfirstlane(a.slab[2 * slab + 1]);
    const int ngrp = (cnt + 7) >> 3;
    const u32x16 *p = a.ent + (e0 >> 3);
    uint32_t *prog = a.prog + xcd * kSweepSlots;
    const int row0 = a.row_base + xcd * a.rows_per_xcd + slot * a.rpw;
    const int xcd_end = min(a.row_end, a.row_base + (xcd + 1) * a.rows_per_xcd);
    const int nvalid = __builtin_amdgcn_readfirstlane(max(0, min(a.rpw, xcd_end - row0)));
    const unsigned long long xb = (unsigned long long)a.X, yb = (unsigned long long)(a.Y + (size_t)(nvalid > 0 ? row0 : 0) * 256);
    const u32x4s rsx = {(unsigned)xb, (unsigned)(xb >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
    const u32x4s rsy = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)yb),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(yb >> 32) & 0xffffu)), (unsigned)nvalid * 1024u, 0x00020000u};
    const int voff = lane * 16;
    const int wm1 = a.window - 1;
    // the tag distinguishes this launch's progress words from what earlier launches left in the line (16 bits: a wrap-around
    // after 65 536 launches can at worst make one launch run unsynchronised); wave 0 of workgroup 0 advances the counter when it
    // is done - a workgroup that starts later than that tags with the next launch's value and merely loses the hint
    const unsigned etag = ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) << 16;
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
        // does the group about to be fetched start in a later column block?  (the last group's successor belongs to the next slab)
        "s_and_b32 s66, s32, 0xffffff\n"
        "s_lshr_b32 s66, s66, %[logb]\n"
        "s_cmp_le_u32 s66, s65\n"
        "s_cbranch_scc1 L_nocross_%=\n"
        "s_cmp_le_u32 s64, 1\n"
        "s_cbranch_scc1 L_nocross_%=\n"
        "s_mov_b32 s65, s66\n"
        "s_or_b32 s67, s66, %[etag]\n"
        PUBLISH
        "s_cmp_lt_u32 s66, %[window]\n"
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_cmp_lg_u32 s77, 0\n"
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_sub_u32 s70, s67, %[wm1]\n"                // every wave of the XCD must fetch from block blk - window + 1 or later
        "s_cmp_eq_u32 s76, 0\n"
        "s_cbranch_scc1 L_slow_%=\n"
        BEHIND
        "s_cbranch_scc1 L_prefetch_%=\n"
        "L_slow_%=:\n"
        "s_mov_b32 s71, 0\n"
        "L_spin_%=:\n"
        "global_load_dwordx4 v[244:247], %[voff], %[prog] sc1\n"
        "s_waitcnt vmcnt(0)\n"
        BEHIND
        "s_cbranch_scc1 L_prefetch_%=\n"
        "s_sleep 4\n"
        "s_add_u32 s71, s71, 1\n"
        "s_cmp_lt_u32 s71, 200\n"
        "s_cbranch_scc1 L_spin_%=\n"
        "s_mov_b32 s77, 1\n"                          // timed out: the hint is not worth more waiting
        "L_prefetch_%=:\n"
        "global_load_dwordx4 v[244:247], %[voff], %[prog] sc1\n"
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
        "s_or_b32 s67, %[nblk], %[etag]\n"            // behind its last entry: nobody waits for this wave any more
        PUBLISH
        "s_mov_b32 s68, 0\n"
        "s_mov_b32 s69, 0\n"
        "s_mov_b32 s64, %[nvalid]\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cbranch_scc1 L_done_%=\n"
        "L_st_%=:\n"
        "s_set_gpr_idx_on s68, gpr_idx(SRC0)\n"
        "v_mov_b32 v236, v0\n"
        "v_mov_b32 v237, v1\n"
        "v_mov_b32 v238, v2\n"
        "v_mov_b32 v239, v3\n"
        "s_set_gpr_idx_off\n"
        "buffer_store_dwordx4 v[236:239], %[voff], %[rsy], s69 offen\n"     // (no nt: the dense stage reads S right back - 1.351 -> 1.318 ms per RK4 step on C2, alternating builds in one box)
        "s_add_u32 s68, s68, 4\n"
        "s_add_u32 s69, s69, 0x400\n"
        "s_sub_u32 s64, s64, 1\n"
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 L_st_%=\n"
        "L_done_%=:\n"
        "s_waitcnt vmcnt(0)\n"
        :
        : [voff] "v"(voff), [rsx] "s"(rsx), [rsy] "s"(rsy), [ent] "s"(p), [ngrp] "s"(ngrp), [prog] "s"(prog), [etag] "s"(etag),
          [nblk] "s"(a.nblk), [slot4] "s"(slot * 4), [nvalid] "s"((a.dbg & 1) ? 0 : nvalid), [logb] "s"(a.logb), [window] "s"(a.window), [wm1] "s"(wm1)
        : "memory", "vcc", "scc", "m0", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", V10(1), V10(2), V10(3), V10(4), V10(5), V10(6),
          V10(7), V10(8), V10(9), V10(10), V10(11), V10(12), V10(13), V10(14), V10(15), V10(16), V10(17), V10(18), V10(19), V10(20), V10(21),
          V10(22), V10(23), "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "s16", "s17", "s18", "s19", S10(2), S10(3),
          S10(4), S10(5), S10(6), S10(7));
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(a.epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


int spmm_sweep_supported(const ndcn_csr *A, int H) {
    static const int enabled = [] { const char *e = getenv("NDCN_SWEEP"); return e ? atoi(e) : 1; }();
    // (256 workgroups of a pass wait on each other's progress words: all of them must be resident, one per CU of the whole chip)
    return enabled && A && H == 256 && A->sweep_ent && A->sweep_slab && A->sweep_prog && A->sweep_passes > 0 &&
           A->sweep_rpw > 0 && A->sweep_rpw <= kSweepRows && A->n_cols * (int64_t)1024 < (1ll << 32) && device_is_whole_chip();
}

// Y = A X through the operator's column-sweep plan (H = 256, no halo panel, alpha = 1, no activation)
int spmm_sweep_f32(const ndcn_csr *A, const float *X, float *Y, hipStream_t st) {
    if (A->n_rows == 0) return NDCN_OK;
    static const int logb_env = [] { const char *e = getenv("NDCN_SWEEP_LOGB"); return e ? atoi(e) : 0; }();
    static const int win_env = [] { const char *e = getenv("NDCN_SWEEP_WINDOW"); return e ? atoi(e) : 0; }();
    const int logb = logb_env > 0 ? logb_env : A->sweep_logb, window = win_env > 0 ? win_env : A->sweep_window;
    ProfScope prof(PROF_SPMM, st, 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * 256 * (double)(A->n_rows + A->n_cols), 2.0 * A->nnz * 256);
    for (int p = 0; p < A->sweep_passes; ++p) {
        SweepArgs a;
        a.X = X;
        a.ent = reinterpret_cast<const u32x16 *>(A->sweep_ent);
        a.slab = A->sweep_slab + (size_t)p * 2 * kXcds * kSweepSlots;
        a.Y = Y;
        a.prog = A->sweep_prog + (size_t)p * kXcds * kSweepSlots;
        a.x_bytes = (unsigned)(A->n_cols * 1024);
        a.row_base = (int)(p * A->sweep_rows_per_pass);
        a.row_end = (int)std::min<int64_t>(A->n_rows, (p + 1) * A->sweep_rows_per_pass);
        const int np = a.row_end - a.row_base;
        a.rows_per_xcd = (np + kXcds - 1) / kXcds;
        a.rpw = (a.rows_per_xcd + kSweepSlots - 1) / kSweepSlots;
        a.epoch = A->sweep_prog + (size_t)A->sweep_passes * kXcds * kSweepSlots + p;
        a.logb = logb;
        a.nblk = (int)((A->n_cols + (1ll << logb) - 1) >> logb);
        a.window = window;
        static const int dbg_env = [] { const char *e = getenv("NDCN_SWEEP_DBG"); return e ? atoi(e) : 0; }();
        a.dbg = dbg_env;
        if (a.nblk >= 65536 || a.rpw > kSweepRows) { set_error("spmm_sweep: plan does not fit the kernel"); return NDCN_EINVAL; }
        hipLaunchKernelGGL(spmm_sweep_kernel, dim3(kCus), dim3(kSweepWaves * 64), 0, st, a);
        NDCN_LAUNCH_CHECK();
    }
    return NDCN_OK;
}

}  // namespace ndcn
