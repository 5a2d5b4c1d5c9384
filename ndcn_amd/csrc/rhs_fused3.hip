// The whole ODEFunc  K = relu(W (A X) + b)  (neural_dynamics.py:27-36) plus the Runge-Kutta algebra that consumes K
// (rk_common.py:45-60,72-78; misc.py:22-25,146-157), H = 256, for operators that carry the 16-row group-record plan
// (ndcn_csr::rec, built by ndcn_csr_create: csr_plan.hip): the gather side of spmm_rec.hip feeding the split-fp16 MFMA product of split16.h,
// one persistent workgroup per CU.
//
// Why (profiles/r02g_fused_timing.txt): rhs_fused2's gather waves fetch every neighbour row through the vector-memory
// path into registers - 9 fetches of 1 KiB per output row on the 8-neighbour lattice, 9 GB per launch L2 -> CU - and
// that path, not HBM and not the matrix pipe, bounds its low-stage launches.  The group-record plan stages each DISTINCT
// neighbour row of a 4 x 4 lattice patch once (36 rows for 16 outputs: 2.25 fetches per row) by LDS-DMA, no register round
// trip; the LDS that costs is found by shrinking the S tile from 64 to 32 rows.
//
//   16 waves: 8 producer | 8 MFMA, in lock-step over "steps" (one 16-row group each), ONE workgroup barrier per step
//     producer step s: own LDS-DMA requests of group s landed (counted vmcnt); barrier; request record s + 2 and its
//              share (5 of 40) of the union rows of group s + 1 (column ids out of the record that landed a step
//              earlier); then two rows: fold the row's (slot, value) entries over the staged union rows (stored order:
//              bit-identical to a sequential loop), form the row's power-of-two scale (DPP max over the wave) and write the
//              row as its two fp16 pieces [256 high | 256 low] -> row 16 (s & 1) + i of S tile (s / 2) & 1: the 1 KiB the
//              fp32 row would occupy.  The slot it overwrites holds K (fp32) of the tile staged 4 steps earlier: that row's
//              RK epilogue (K out, y_next / error sums) comes first; its row-local panels were requested a step ahead.
//     MFMA     step s: half (s & 1) of the 16 k-steps of tile s / 2 - 1 (the other S buffer): wave w owns output columns
//              [32 w, 32 w + 32): per k-step two ds_read_b128 (ready A operands - no VALU in the loop) and three
//              v_mfma_f32_32x32x16_f16 (low x high, high x low, high x high; small terms first).  The weights of 10 k-steps
//              stay in registers, the other 6 (the even ones) stream from L2 through a two-slot ring refilled right after
//              use, three k-steps of products ahead of the next use.  After the second half the eight waves meet on an LDS
//              counter (everyone has read S) and drop K = relu(unscale * acc + b) into the tile they consumed.
//   History: rounds 1-2 split every S value into three bf16 pieces INSIDE each MFMA wave (six products,
//   38 VALU per split, eight waves each splitting the whole tile: the loop was VALU-bound, 13 k cycles per 32-row tile); two
//   fp16 pieces make the split planes exactly as large as the fp32 tile, so the producer that holds the row splits it once.
//   A 12-wave form (4 producers | 8 MFMA waves, ALL weights resident) is a build option (f3_producers): correct, not faster.
//   LDS: 2 union buffers (80 KiB) + 3 records (6 KiB) + 2 S tiles (32 x 260 floats each, 65 KiB) + row ids / bias / scales
//   = 153 KiB.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "gather.h"
#include "rec_common.h"
#include "split16.h"

#pragma clang fp contract(off)   // the RK algebra must round like the reference's separate mul / add ops

namespace ndcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kF3R = 16, kF3Cap = 40, kF3RecW = 2;       // the plan shape this kernel is written for
#ifndef NDCN_F3_RING
#define NDCN_F3_RING 2
#endif
// The L2 -> CU weight stream competes with the producers' requests for the vector-memory path (round 2, bf16: 12 KiB per row;
// with the refills switched off the dopri5 step of the metric case took 6.2 ms instead of 11.9).  Whatever registers the MFMA
// waves have left hold k-steps of their weights for good: 10 of 16 (3 KiB per row still streamed; 8: 9.82, 9: 9.66 ms/step
// with the split still in the MFMA waves; 10 + ring 2 = 123 registers with ready operands: 9.21 on the same box class).
#ifndef NDCN_F3_RESIDENT
#define NDCN_F3_RESIDENT 10    // two fp16 planes: 8 registers per resident k-step
#endif
#ifndef NDCN_F3_PRODUCERS
#define NDCN_F3_PRODUCERS 8
#endif
#ifndef NDCN_F3_HEAVY_4P
#define NDCN_F3_HEAVY_4P 0
#endif
enum { F3_PLAIN = 0, F3_COMBINE = 1, F3_ERROR = 2, F3_RK4 = 3 };
// The wave split is a function of the variant (f3_producers).  Shipped: 16 waves everywhere (8 producers | 8 MFMA waves, 128
// registers: 10 resident weight k-steps).  NDCN_F3_HEAVY_4P = 1 gives the launches with >= 3 earlier stages the 12-wave form
// (4 producers | 8 MFMA waves, 168 registers: ALL 16 k-steps resident, no weight stream at all) - correct since the epilogue
// arguments stopped spilling scalar registers, but not faster: alternating runs on one box 8.92-8.94 (shipped) vs 8.98-9.00
// ms/step; the 12-wave form for EVERY launch 9.36 vs 9.03 (its light launches become producer-bound).  Single A/B pairs had
// suggested a gain on the heavy launches: the boxes differ by 4 % among themselves, repeats on one box by 0.3 %.
constexpr int kF3WM = 8;                                    // MFMA waves: one 32-column n-tile each
constexpr int f3_producers(int mode, int np) {
    return (NDCN_F3_HEAVY_4P && ((mode == F3_COMBINE && np >= 3) || (mode == F3_ERROR && np == 5))) ? 4 : NDCN_F3_PRODUCERS;
}
constexpr int kF3WavesMax = 16;
constexpr int kF3NBuf = 2, kF3NRec = 3;                  // one group in flight ahead of the one being folded
constexpr int kF3Tile = 32, kF3Ld = 260;                 // S tile: 2 groups; +4 floats per row: conflict-free b128
constexpr int kF3MaxPrev = 5;
constexpr int kF3E0 = kF3Cap + 2 * kF3R;
constexpr unsigned kF3OffRec = kF3NBuf * kF3Cap * 1024;
constexpr unsigned kF3OffS = kF3OffRec + kF3NRec * kF3RecW * 1024;
constexpr unsigned kF3OffRow = kF3OffS + 2 * kF3Tile * kF3Ld * 4;
constexpr unsigned kF3OffSync = kF3OffRow + 2 * kF3Tile * 4;
constexpr unsigned kF3OffBias = kF3OffSync + 16;
constexpr unsigned kF3OffUnscale = kF3OffBias + 256 * 4;      // per S-tile row: 1 / (row scale) of the fp16 split (split16.h)
constexpr unsigned kF3OffWun = kF3OffUnscale + 2 * kF3Tile * 4;   // per output column: 1 / (scale of the column's weight row)
constexpr unsigned kF3Lds = kF3OffWun + 256 * 4;

// workgroup barrier: this wave's LDS traffic has been performed first, and hipcc moves no memory access across it
__device__ __forceinline__ void f3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// cycle accounting is compiled in only on request (its counters cost the MFMA waves registers they do not have)
#ifdef NDCN_F3_TIMING
constexpr bool kF3Timing = true;
#else
constexpr bool kF3Timing = false;
#endif

// Which of a tile's 16 k-steps keep their weights in registers.  The accumulation order stays k = 0 .. 15; with at most
// 8 streamed k-steps they are the EVEN ones (0, 2, ..), so that a ring slot refilled right after its k-step 2 j is read
// again at k-step 2 (j + ring): 2 ring - 1 k-steps of products cover the L2 round trip (in a block behind the resident ones
// it was ring - 1).
template <int NS>
struct F3Order {
    static constexpr bool kInterleaved = NS <= 8;
    static constexpr bool streamed(int ks) { return kInterleaved ? ((ks & 1) == 0 && ks / 2 < NS) : ks >= 16 - NS; }
    static constexpr int stream_index(int ks) { return kInterleaved ? ks / 2 : ks - (16 - NS); }
    static constexpr int stream_ks(int j) { return kInterleaved ? 2 * j : 16 - NS + j; }
    static constexpr int res_index(int ks) {                  // position among the resident k-steps, in k order
        int n = 0;
        for (int q = 0; q < ks; ++q) n += streamed(q) ? 0 : 1;
        return n;
    }
};

struct F3Args {
    const int *rec;
    int n_groups;
    const float *X, *Xh;
    const float *Xadd;               // XOP 1: the evaluation's input is X + xadd_c * Xadd; XOP 2: X (.) [Xadd > 0] - formed on the staged rows
    float xadd_c;
    float *S_out;                    // SOUT variants: S = A X (the folded rows, fp32) is written too
    int n_own;
    const void *Wq;                  // split weights (pack_weight_256: two fp16 planes in MFMA B-operand order + scales)
    const float *bias;
    float *K;
    int relu;
    const int *rowptr, *colidx;      // direct gather of groups the plan could not stage
    const float *val;
    unsigned long long *dbg_cycles;  // NDCN_FUSED3_TIMING: per (block, wave) {cycles between barriers, cycles inside barriers}
    int dbg;                         // NDCN_FUSED3_DBG (timing experiments, results wrong): 1 no MFMA, 2 no fold, 4 no epilogue, 8 staged rows from a 1 MiB window, 64 no weight refills
};
// the experiment switches exist in timing builds only (in the product they would cost scalar registers and branches)
__device__ __forceinline__ int f3_dbg(const F3Args &a) { return kF3Timing ? a.dbg : 0; }
struct F3Epi {
    const float *y0;
    const float *kprev[kF3MaxPrev];
    float *y_next;
    double *partials;                // ERROR: [gridDim.x * producer waves][2]
    float c[kF3MaxPrev + 1];         // c[0 .. n_prev-1] for kprev, c[n_prev] for the new K ; RK4: c[0] = dt
    float rtol, atol;
    const float *y1;                 // ERROR: the state of the error record, by row of this launch
    float *y_aux;                    // COMBINE, nullable: second linear combination (no y0), coefficients c2[]
    float c2[kF3MaxPrev + 1];
};

// The epilogue arguments (8 pointers, 14 scalars) are NOT read through the kernel-parameter object: the compiler would keep all
// of them live in SGPRs across the producer loop - the variants with >= 2 earlier stages then spill 8-43 scalar registers
// into VGPR lanes, which costs every wave of the kernel registers (the MFMA waves' resident weights) - but from the kernarg
// segment where they are used, through a laundered pointer the compiler cannot hoist (a handful of s_loads per K row; as
// rhs_fused2.hip).
typedef const __attribute__((address_space(4))) F3Epi *F3EpiPtr;
constexpr int kF3EpiKernargOffset = (int)((sizeof(F3Args) + 7) / 8 * 8);       // kernel parameters: F3Args, F3Epi (both 8-aligned)

// XADD: the input of the evaluation is X + c Xadd (dopri5's first stage input y0 + dt beta_21 k1),
// formed where it is needed instead of by a kernel of its own (3 panels): every wave fetches the Xadd rows of the union rows
// it stages, a step ahead like them, and adds them into its LDS rows - one product, one sum per element, the roundings of
// combine_kernel - before the barrier that hands the group to the fold.  One more gather (1.07 panels) instead of 3 panels.
// XOP 2 (round 5): the input is X (.) [M > 0] with the mask panel M in Xadd - the transposed half of odeint_adjoint's right-hand side
// gathers gZ = a (.) [K > 0] this way (adjoint.py:34-59; _impl/adjoint_fused.py) instead of reading it from a panel that a launch
// of its own (3 panels) would have to write first.  SOUT: the folded rows S = A X leave the kernel too (one more store per row):
// the weight gradient of the same right-hand side, gZ^T S, needs them (instead of a second SpMM).
// NT: the epilogue's stores carry the non-temporal hint - right for panels far beyond the 256 MiB Infinity Cache (the metric's 1 GB
// panels: 8.35 against 8.58 ms per step without it), wrong for panels that live in it and are read right back by the next launch
// (BASELINE config 2, 102 MB: 1.323 against 1.290 ms per RK4 step); the launcher decides by the panel's size
// NOK: the store of K compiled out (RkOpt::no_k) - a template argument, instantiated only for the two launches that use it (COMBINE
// with no earlier stage: an Euler step / midpoint's second stage; RK4's fourth stage), so that the dopri5 variants carry no branch
// for it (as a run-time test of a.K it cost every variant 241 instructions and 33 waits: round-5 review)
template <bool HALO, int MODE, int NP, int XOP = 0, bool NT = true, bool SOUT = false, bool NOK = false>
__global__ __launch_bounds__(64 * (f3_producers(MODE, NP) + kF3WM)) void rhs_fused3_kernel(F3Args a, F3Epi epi_by_kernarg_only) {
    constexpr bool XADD = XOP != 0;                          // a second panel is gathered alongside X
    static_assert(!(XADD && HALO), "the halo rows of Xadd are not exchanged");
    (void)epi_by_kernarg_only;
    constexpr int kF3WP = f3_producers(MODE, NP), kF3Waves = kF3WP + kF3WM;   // producer waves (LDS-DMA + fold + epilogue) | MFMA waves
    constexpr int kF3CapD = kF3Cap / kF3WP;
    static_assert(kF3Cap % kF3WP == 0 && kF3R % kF3WP == 0 && kF3WP >= 4, "wave split");
    auto epi_args = [&]() -> F3EpiPtr {
        auto base = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        F3EpiPtr q = (F3EpiPtr)(base + kF3EpiKernargOffset);
        asm volatile("" : "+s"(q));                           // a fresh pointer per call: the loads stay where they are used
        return q;
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_off = lane * 16;
    const unsigned lds_x = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    const unsigned lds_r = lds_x + kF3OffRec;
    const int *rbuf = reinterpret_cast<const int *>(lds) + kF3OffRec / 4;
    float *s_tiles = lds + kF3OffS / 4;
    int *s_rowid = reinterpret_cast<int *>(lds) + kF3OffRow / 4;
    unsigned *s_sync = reinterpret_cast<unsigned *>(lds) + kF3OffSync / 4;
    float *s_bias = lds + kF3OffBias / 4;          // a fetch issued from inside the MFMA loop queues behind the producers' requests
    float *s_unscale = lds + kF3OffUnscale / 4;     // per S-tile row: 1 / (row scale) of the fp16 split
    float *s_wun = lds + kF3OffWun / 4;             // per output column: 1 / (weight-row scale)

    // groups of this workgroup: XCD x owns a contiguous chunk, its workgroups take the chunk's groups round-robin
    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wpx = gridDim.x / kXcds;
    const int chunk = (a.n_groups + kXcds - 1) / kXcds;
    const int g_lo = xcd * chunk, g_hi = min(a.n_groups, g_lo + chunk);
    const int g0 = g_lo + wg;
    const int my = g0 < g_hi ? (g_hi - g0 + wpx - 1) / wpx : 0;
    const int n_tiles = (my + 1) >> 1;
    const int n_steps = 2 * n_tiles + 4;                       // + 2 for the MFMA lag, + 2 for the epilogue lag
    if (my == 0) {
        if (MODE == F3_ERROR && wave < kF3WP && lane == 0) {  // the finish kernel sums EVERY slot
            double *partials = epi_args()->partials;
            partials[2 * (blockIdx.x * kF3WP + wave)] = 0.0;
            partials[2 * (blockIdx.x * kF3WP + wave) + 1] = 0.0;
        }
        return;
    }

    if (wave >= kF3WP) {
        // ------------------------------------------------------------------------------------------ MFMA waves
        // wave w owns kNT n-tiles of 32 output columns: [32 kNT w, 32 kNT (w + 1)).  Split weights: block (n-tile j, k-step
        // ks, plane pl) = 64 lanes x 16 bytes; a ring of kRing k-steps x kNT n-tiles x 3 planes in registers, every slot
        // refilled right after use.  The ring runs on across halves and tiles: the weights are the same for every tile.
        const int mw = wave - kF3WP;
        constexpr int kNT = 8 / kF3WM;
        constexpr int kRing = kF3WP == 4 ? 1 : NDCN_F3_RING;
        constexpr int kRes = kF3WP == 4 ? 16 : NDCN_F3_RESIDENT;    // resident k-steps: their weights never leave the registers
        constexpr int kNS = 16 - kRes;                       // streamed k-steps per tile
        static_assert(kNS == 0 || kRing <= kNS, "ring");
        constexpr int kPl = kS16Planes;                     // two fp16 pieces per weight (split16.h)
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.Wq), 0, kS16Bytes, 0x00020000);
        const int q_slab = (kNT * mw) * 16 * kPl * 1024;
        auto ldq = [&](int jj, int ks, int pl) {
            int ws = q_slab;
            asm volatile("" : "+s"(ws));
            return __builtin_amdgcn_raw_buffer_load_b128(rsQ, lane_off, ws + ((jj * 16 + ks) * kPl + pl) * 1024, 0);
        };
        u32x4 Bq[kRing][kNT][kPl];
        u32x4 Br[kRes > 0 ? kRes : 1][kNT][kPl];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int jj = 0; jj < kNT; ++jj)
#pragma unroll
                for (int pl = 0; pl < kPl; ++pl)
                    if (!F3Order<kNS>::streamed(ks)) Br[F3Order<kNS>::res_index(ks)][jj][pl] = ldq(jj, ks, pl);
#pragma unroll
        for (int u = 0; u < (kNS > 0 ? kRing : 0); ++u)
#pragma unroll
            for (int jj = 0; jj < kNT; ++jj)
#pragma unroll
                for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, F3Order<kNS>::stream_ks(u), pl);
        f32x16 acc[kNT];
        // k-steps [8 HALF, 8 HALF + 8) of the tile at `src` (tile index tb = 0 | 1 selects its rows' scales)
        auto mfma_half = [&](const float *src, int tb, auto half_tag) {
            constexpr int HALF = decltype(half_tag)::value;
            // per-lane addresses are re-derived from a laundered lane id at every use: hoisted out of the step loop they
            // occupy registers the ring needs, and hipcc then parks them in scratch - whose reloads queue behind the
            // producers' requests like every other vector-memory operation of this CU
            int ln = lane;
            asm volatile("" : "+v"(ln));
            // S row m of the tile = [256 fp16 high pieces | 256 fp16 low pieces] (written by the wave that folded the row):
            // this lane's A operands of k-step ks are the 16 bytes at 32 ks + 16 (lane >> 5) of either half of row lane & 31
            const char *ap = reinterpret_cast<const char *>(src) + (ln & 31) * (kF3Ld * 4) + 16 * (ln >> 5) + 256 * HALF;
            u32x4 A0 = *reinterpret_cast<const u32x4 *>(ap), A1 = *reinterpret_cast<const u32x4 *>(ap + 512);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = 8 * HALF + i;
                const bool res = !F3Order<kNS>::streamed(ks);
                const int j = res ? 0 : F3Order<kNS>::stream_index(ks);          // streamed k-step number j of the tile
                const int u = j % kRing;
                auto bq = [&](int jj, int pl) -> const u32x4 & { return res ? Br[res ? F3Order<kNS>::res_index(ks) : 0][jj][pl] : Bq[u][jj][pl]; };
                // small products first (the order of rhs_fused2.hip: identical rounding).  The next k-step's pieces are
                // requested from LDS into the SAME registers as soon as their last product has been issued (a second pair of
                // operand registers would cost a resident k-step)
#pragma unroll
                for (int jj = 0; jj < kNT; ++jj) s16_mfma(acc[jj], A1, bq(jj, 0));
                __builtin_amdgcn_sched_barrier(0);      // (pinned: a hoisted LDS read takes a second register set)
                if (i + 1 < 8) A1 = *reinterpret_cast<const u32x4 *>(ap + 32 * (i + 1) + 512);
#pragma unroll
                for (int jj = 0; jj < kNT; ++jj) s16_mfma(acc[jj], A0, bq(jj, 1));
#pragma unroll
                for (int jj = 0; jj < kNT; ++jj) s16_mfma(acc[jj], A0, bq(jj, 0));
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < 8) A0 = *reinterpret_cast<const u32x4 *>(ap + 32 * (i + 1));
                __builtin_amdgcn_sched_barrier(0);      // the refills stay BEHIND the products that read the slot (hoisted, they need more registers)
                if (!res && !(f3_dbg(a) & 64)) {
                    // streamed k-step j: its slot is refilled with streamed k-step j + kRing or, for the last kRing of a tile,
                    // with streamed k-step (slot index) of the NEXT tile
                    const int kn = F3Order<kNS>::stream_ks(j + kRing < kNS ? j + kRing : u);
#pragma unroll
                    for (int jj = 0; jj < kNT; ++jj)
#pragma unroll
                        for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, kn, pl);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto dump_tile = [&](float *dst, int tb) {
            // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]; multiplied back by 1 / (weight-row scale of column n), then by
            // 1 / (scale of S row m) - both exact; in this order no intermediate leaves the fp32 range for any row the scale clamps
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 4; ++q) {                           // (one scale at a time: the ring's registers are live here)
                const float *unp = s_unscale + tb * kF3Tile + 8 * q + 4 * (ln >> 5);
#pragma unroll
                for (int jj = 0; jj < kNT; ++jj) {
                    const int col = 32 * (kNT * mw + jj) + (ln & 31);
                    const float bv = s_bias[col], wu = s_wun[col];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int m = rr + 8 * q + 4 * (ln >> 5);
                        float o = (acc[jj][4 * q + rr] * wu) * unp[rr] + bv;
                        if (a.relu) o = relu_nan(o);
                        dst[m * kF3Ld + col] = o;
                    }
                }
            }
        };
        // One loop iteration = one tile = two steps, straight-line: the ring's registers then flow from half to half and
        // around ONE back edge.  (With a per-step loop and the halves on two branches hipcc reconciled the two register
        // assignments of the ring with v_mov copies behind s_waitcnt vmcnt at every join - i.e. each half ended by
        // waiting for the refills it had just requested.)
        unsigned long long cyc_work = 0, cyc_wait = 0, c_prev = 0;
        auto barrier_t = [&]() {
            const unsigned long long cb = (kF3Timing && a.dbg_cycles) ? __builtin_readcyclecounter() : 0;
            f3_barrier();
            if (kF3Timing && a.dbg_cycles) { const unsigned long long ce = __builtin_readcyclecounter(); cyc_work += cb - c_prev; cyc_wait += ce - cb; c_prev = ce; }
        };
        f3_barrier();                                               // [P]
        if (kF3Timing && a.dbg_cycles) c_prev = __builtin_readcyclecounter();
        barrier_t();                                                // [A_0]
        barrier_t();                                                // [A_1]
        for (int t = 0; t < n_tiles; ++t) {                         // the tile whose S rows were folded in steps 2 t, 2 t + 1
            float *tile = s_tiles + (t & 1) * kF3Tile * kF3Ld;
            barrier_t();                                            // [A_(2t+2)]
#pragma unroll
            for (int jj = 0; jj < kNT; ++jj)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[jj][i] = 0.f;
            if (!(f3_dbg(a) & 1)) mfma_half(tile, t & 1, std::integral_constant<int, 0>{});
            barrier_t();                                            // [A_(2t+3)]
            if (!(f3_dbg(a) & 1)) mfma_half(tile, t & 1, std::integral_constant<int, 1>{});
            // every MFMA wave has read its last S value before anyone overwrites the tile with K
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(s_sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(s_sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(kF3WM * (t + 1)))
                __builtin_amdgcn_s_sleep(1);
            dump_tile(tile, t & 1);
        }
        barrier_t();                                                // the two steps in which the producers finish the last tile's K rows
        barrier_t();
        // the last refills of the weight ring wrap into a tile that does not exist
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        if (kF3Timing && a.dbg_cycles && lane == 0) {
            a.dbg_cycles[2 * (blockIdx.x * kF3Waves + wave)] = cyc_work;
            a.dbg_cycles[2 * (blockIdx.x * kF3Waves + wave) + 1] = cyc_wait;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- producer waves
    // (LDS-DMA requests, fold, RK epilogue).  One vector-memory counter serves three kinds of requests that complete in
    // issue order: the DMA of the NEXT group (must have landed at the top of the next step), the row-local panels of slot
    // of each slot (consumed a step after their request) and the epilogue's stores.  since_dma / since_p[q]
    // count what was issued AFTER the respective request: exactly that many operations may still be outstanding when
    // its data is used.
    const int pw = wave;
    constexpr int kLoads = MODE == F3_PLAIN ? 0 : NP + 1 + (MODE == F3_ERROR ? 1 : 0);
    struct Panels { f32x4 km[NP > 0 ? NP : 1]; f32x4 y0v, y1v; };
    double err_sum = 0.0, err_bad = 0.0;
    constexpr int kRPW = kF3R / kF3WP;                            // rows (slots) per producer wave and step
    int since_dma = 0, since_p[kRPW] = {};
    auto issued = [&](int n) {
        since_dma += n;
#pragma unroll
        for (int q = 0; q < kRPW; ++q) since_p[q] += n;
    };
    if (lane < 2 * kF3Tile / kF3WP) s_rowid[pw * (2 * kF3Tile / kF3WP) + lane] = -1;     // 64 slots
    if (pw == 0 && lane == 0) *s_sync = 0u;
    if (pw < 4) s_bias[64 * pw + lane] = a.bias ? a.bias[64 * pw + lane] : 0.f;
    // 1 / (scale of weight row n), written behind the packed planes by pack_weight_256 (split16.h)
    if (pw < 4) s_wun[64 * pw + lane] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.Wq) + kS16Bytes)[64 * pw + lane];

    auto dma_rec = [&](int it) {
        if (pw < kF3RecW && it < my) {
            dma_row(reinterpret_cast<const float *>(a.rec + ((size_t)(g0 + it * wpx) * kF3RecW + pw) * 256),
                    lds_r + (unsigned)(((it % kF3NRec) * kF3RecW + pw) * 1024), lane_off);
            issued(1);
        }
    };
    f32x4 xk[XADD ? kF3CapD : 1];
    auto dma_x = [&](int it) {
        const int *r = rbuf + (it % kF3NRec) * kF3RecW * 256 + pw * kF3CapD;
        int cc[kF3CapD];
#pragma unroll
        for (int k = 0; k < kF3CapD; ++k) cc[k] = __builtin_amdgcn_readfirstlane(r[k]);
#pragma unroll
        for (int k = 0; k < kF3CapD; ++k) {
            const float *base = a.X;
            int c = cc[k];
            if (f3_dbg(a) & 8) c &= 1023;                        // (timing builds: every staged row out of a 1 MiB window - the launch without its gather's HBM side)
            if (HALO && c >= a.n_own) { base = a.Xh; c -= a.n_own; }
            dma_row(base + (size_t)c * 256, lds_x + (unsigned)(((it % kF3NBuf) * kF3Cap + pw * kF3CapD + k) * 1024), lane_off);
        }
        issued(kF3CapD);
        if (XADD) {
#pragma unroll
            for (int k = 0; k < kF3CapD; ++k)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xk[k]) : "v"((cc[k] << 10) + lane_off), "s"(a.Xadd) : "memory");
            issued(kF3CapD);
        }
    };
    // XADD: the staged rows of group `it` (this wave's share) become X + c Xadd; called once its requests have landed
    auto xadd_rows = [&](int it) {
        f32x4 *xb = reinterpret_cast<f32x4 *>(lds) + ((it % kF3NBuf) * kF3Cap + pw * kF3CapD) * 64 + lane;
#pragma unroll
        for (int k = 0; k < kF3CapD; ++k) {
            asm volatile("" : "+v"(xk[k]));
            if constexpr (XOP == 2) {
                f32x4 v = xb[k * 64];
                v.x = xk[k].x > 0.f ? v.x : 0.f; v.y = xk[k].y > 0.f ? v.y : 0.f;
                v.z = xk[k].z > 0.f ? v.z : 0.f; v.w = xk[k].w > 0.f ? v.w : 0.f;
                xb[k * 64] = v;
            } else {
                xb[k * 64] = xb[k * 64] + xk[k] * a.xadd_c;
            }
        }
    };
    auto ldp = [&](const float *base /*uniform*/, int voff) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(voff), "s"(base) : "memory");
        return v;
    };
    // row-local panels of one K row; branch-free (a slot without a row requests row 0 and ignores it): a register that
    // is the target of an in-flight fetch must be assigned on ONE path only, or hipcc copies it around
    auto request = [&](int row, Panels &p) {
        const int voff = (max(row, 0) << 10) + lane_off;           // panels < 4 GiB (launcher check)
        const F3EpiPtr e = epi_args();
#pragma unroll
        for (int m = 0; m < NP; ++m) p.km[m] = ldp(e->kprev[m], voff);
        p.y0v = ldp(e->y0, voff);
        if (MODE == F3_ERROR) p.y1v = ldp(e->y1, voff);             // the input of this evaluation is y1 (own rows)
        issued(kLoads);
    };
    auto arrived = [&](Panels &p) {
#pragma unroll
        for (int m = 0; m < NP; ++m) asm volatile("" : "+v"(p.km[m]));
        asm volatile("" : "+v"(p.y0v));
        if (MODE == F3_ERROR) asm volatile("" : "+v"(p.y1v));
    };
    auto stp = [&](float *base /*uniform*/, int voff, f32x4 v) {
        // (s_nop: the data registers of a 16-byte store must not be written in the next wait state)
        if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
    };
    // RK epilogue of one K row (as rhs_fused2.hip: epi_finish)
    auto epilogue = [&](int row, f32x4 kn, const Panels &p) {
        const int voff = (row << 10) + lane_off;
        if constexpr (!NOK) {                                       // (RkOpt::no_k: only y_next is wanted)
            stp(a.K, voff, kn);
            issued(1);
        }
        if (MODE == F3_PLAIN) return;
        const F3EpiPtr e = epi_args();
        if (MODE == F3_RK4) {
            // rk4_alt_step_func (rk_common.py:72-78), same operator order as fixed_stage_kernel ops 2-5
            const float dt = e->c[0];
            f32x4 s;
            if (NP == 0) s = (kn * dt) / 3.f;
            else if (NP == 1) s = (p.km[0] / -3.f + kn) * dt;
            else if (NP == 2) s = ((p.km[0] - p.km[NP > 1 ? 1 : 0]) + kn) * dt;
            else s = (((p.km[0] + p.km[NP > 1 ? 1 : 0] * 3.f) + p.km[NP > 2 ? 2 : 0] * 3.f) + kn) * (dt / 8.f);
            stp(e->y_next, voff, p.y0v + s);
            issued(1);
            return;
        }
        // sum of the stages left to right, the new one last (misc.py:22-25), each product rounded on its own
        f32x4 s = kn * e->c[NP];
        if (NP > 0) {
            f32x4 u = p.km[0] * e->c[0];
#pragma unroll
            for (int m = 1; m < NP; ++m) u = u + p.km[m] * e->c[m];
            s = u + s;
        }
        if (MODE == F3_COMBINE) {
            stp(e->y_next, voff, p.y0v + s);
            issued(1);
            if (e->y_aux) {                                          // wave-uniform
                f32x4 w2 = kn * e->c2[NP];
                if (NP > 0) {
                    f32x4 u2 = p.km[0] * e->c2[0];
#pragma unroll
                    for (int m = 1; m < NP; ++m) u2 = u2 + p.km[m] * e->c2[m];
                    w2 = u2 + w2;
                }
                stp(e->y_aux, voff, w2);
                issued(1);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float tol = e->atol + e->rtol * max_nan(fabsf(p.y0v[q]), fabsf(p.y1v[q]));
            const float z = s[q] / tol;
            err_sum += (double)(z * z);
            err_bad += (double)(int)(!(fabsf(p.y1v[q]) <= 3.402823466e38f));
        }
    };
    // fold of row i of the group staged for step s -> acc; returns the row id (-1: padding)
    auto fold = [&](int s, int i, f32x4 &acc) -> int {
        const int *r = rbuf + (s % kF3NRec) * kF3RecW * 256;
        const f32x4 *xb = reinterpret_cast<const f32x4 *>(lds) + (s % kF3NBuf) * kF3Cap * 64;
        const int row = __builtin_amdgcn_readfirstlane(r[kF3Cap + 2 * i]);
        const int meta = __builtin_amdgcn_readfirstlane(r[kF3Cap + 2 * i + 1]);
        acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (row < 0 || (f3_dbg(a) & 2)) return row;
        const int cnt = meta & 0xffff, ofs = meta >> 16;
        if (cnt != 0xffff) {
            int es = 0;
            float ev = 0.f;
            if (lane < cnt) { es = r[kF3E0 + 2 * (ofs + lane)]; ev = __builtin_bit_cast(float, r[kF3E0 + 2 * (ofs + lane) + 1]); }
            rec_row<(kF3WP == 8 || kLoads < 5)>(es, ev, cnt, [&](int slot) { return xb[slot * 64 + lane]; }, acc);
        } else {
            // group not staged: gather this row from the CSR arrays, 64 entries at a time (rare; full waits: every
            // request of this wave, the panels in flight included, has landed afterwards)
            const int j0 = a.rowptr[row], j1 = a.rowptr[row + 1];
            for (int jb = j0; jb < j1; jb += 64) {
                const int n = min(64, j1 - jb);
                int es = 0;
                float ev = 0.f;
                if (lane < n) {
                    const int eo = (jb + lane) * 4;
                    asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(es), "=&v"(ev) : "v"(eo), "s"(a.colidx), "s"(a.val) : "memory");
                }
                rec_row<false>(es, ev, n, [&](int c) {
                    const float *p = a.X;
                    if (HALO && c >= a.n_own) { p = a.Xh; c -= a.n_own; }
                    f32x4 v;
                    asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(lane_off), "s"(p + (size_t)c * 256) : "memory");
                    if (XADD) {
                        f32x4 w;
                        asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(lane_off), "s"(a.Xadd + (size_t)c * 256) : "memory");
                        if constexpr (XOP == 2) {
                            v.x = w.x > 0.f ? v.x : 0.f; v.y = w.y > 0.f ? v.y : 0.f; v.z = w.z > 0.f ? v.z : 0.f; v.w = w.w > 0.f ? v.w : 0.f;
                        } else {
                            v = v + w * a.xadd_c;
                        }
                    }
                    return v;
                }, acc);
            }
        }
        return row;
    };
    // slot q (0, 1) of step s: S-tile row and its row-id cell
    auto slot_of = [&](int s, int q) { return ((s >> 1) & 1) * kF3Tile + 16 * (s & 1) + pw + kF3WP * q; };
    auto rowid_at = [&](int sl) { return __builtin_amdgcn_readfirstlane(s_rowid[sl]); };

    dma_rec(0);
    dma_rec(1);
    rec_wait_vmcnt<0>();
    f3_barrier();                                                   // [P] the first records are readable
    dma_x(0);
    since_dma = 0;
    // One panel set per slot; the panels of slot q of step s + 1 are requested right after slot q of step s is done with
    // them - a whole step ahead.
    Panels pan[kRPW];
    if (MODE != F3_PLAIN) {
#pragma unroll
        for (int q = 0; q < kRPW; ++q) { request(-1, pan[q]); since_p[q] = 0; }     // (uniform bookkeeping: no K rows yet)
    }
    unsigned long long cyc_work = 0, cyc_wait = 0, cyc_dma = 0, c_prev = (kF3Timing && a.dbg_cycles) ? __builtin_readcyclecounter() : 0;
    for (int s = 0; s < n_steps; ++s) {
        const unsigned long long ca = (kF3Timing && a.dbg_cycles) ? __builtin_readcyclecounter() : 0;
        rec_wait_vmcnt_rt(since_dma);                               // this wave's share of group s (and of record s + 1) has landed
        if (XADD && s < my) xadd_rows(s);
        const unsigned long long cb = (kF3Timing && a.dbg_cycles) ? __builtin_readcyclecounter() : 0;
        f3_barrier();                                               // [A_s]
        if (kF3Timing && a.dbg_cycles) { const unsigned long long ce = __builtin_readcyclecounter(); cyc_work += ca - c_prev; cyc_dma += cb - ca; cyc_wait += ce - cb; c_prev = ce; }
        dma_rec(s + 2);
        if (s + 1 < my) dma_x(s + 1);
        since_dma = 0;
#pragma unroll
        for (int q = 0; q < kRPW; ++q) {
            const int sl = slot_of(s, q);
            const int er = rowid_at(sl);                            // the K row waiting in the slot (-1: none)
            f32x4 acc;
            int row = -1;
            if (s < my) row = fold(s, pw + kF3WP * q, acc);
            float *srow = s_tiles + sl * kF3Ld + 4 * lane;
            if (MODE != F3_PLAIN) { rec_wait_vmcnt_rt(since_p[q]); arrived(pan[q]); }
            if (er >= 0 && !(f3_dbg(a) & 4)) epilogue(er, *reinterpret_cast<const f32x4 *>(srow), pan[q]);
            if (SOUT && row >= 0) {
                stp(a.S_out, (row << 10) + lane_off, acc);
                issued(1);
            }
            if (row >= 0) {
                // the row leaves this wave as its two fp16 pieces (split16.h), scaled by a power of two from its largest
                // magnitude: [256 high | 256 low] in the 1 KiB the fp32 row (and later its K row) occupies
                unsigned sb, ub;
                s16_scale_bits(s16_wave_umax(s16_row_max_bits(acc)), sb, ub);
                u32x2_s16 h0, h1;
                s16_split4(acc, __builtin_bit_cast(float, sb), h0, h1);
                char *hrow = reinterpret_cast<char *>(s_tiles + sl * kF3Ld) + 8 * lane;
                *reinterpret_cast<u32x2_s16 *>(hrow) = h0;
                *reinterpret_cast<u32x2_s16 *>(hrow + 512) = h1;
                if (lane == 0) s_unscale[sl] = __builtin_bit_cast(float, ub);
            }
            s_rowid[sl] = row;
            // the panels of this slot in the NEXT step (its K rows were staged 3 steps ago)
            if (MODE != F3_PLAIN) { request(s + 1 < n_steps ? rowid_at(slot_of(s + 1, q)) : -1, pan[q]); since_p[q] = 0; }
        }
    }
    rec_wait_vmcnt<0>();
    if (MODE != F3_PLAIN) {
#pragma unroll
        for (int q = 0; q < kRPW; ++q) arrived(pan[q]);
    }
    if (XADD) {                                                     // (their registers stay reserved up to the wait above)
#pragma unroll
        for (int k = 0; k < kF3CapD; ++k) asm volatile("" : "+v"(xk[k]));
    }
    if (kF3Timing && a.dbg_cycles && lane == 0) {
        a.dbg_cycles[2 * (blockIdx.x * kF3Waves + wave)] = cyc_work;
        a.dbg_cycles[2 * (blockIdx.x * kF3Waves + wave) + 1] = cyc_wait + (cyc_dma << 32);
    }
    if (MODE == F3_ERROR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            err_sum += __shfl_down(err_sum, off, 64);
            err_bad += __shfl_down(err_bad, off, 64);
        }
        if (lane == 0) {
            double *partials = epi_args()->partials;
            partials[2 * (blockIdx.x * kF3WP + pw)] = err_sum;
            partials[2 * (blockIdx.x * kF3WP + pw) + 1] = err_bad;
        }
    }
}

static int env_int_f3(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

// the operator carries the 16-row plan; no halo panel in the hub sense (Xh is served), H = 256 is the caller's check
int rhs_fused3_supported(const ndcn_csr *A) {
    static const int enabled = env_int_f3("NDCN_RHS_FUSED3", 1);
    if (!enabled || !A || !A->rec || A->rec_groups <= 0) return 0;
    if (!(A->rec_rows == kF3R && A->rec_cap == kF3Cap && A->rec_kib == kF3RecW)) return 0;
    return (A->n_cols * (int64_t)1024 < (1ll << 32) && A->n_rows * (int64_t)1024 < (1ll << 32)) ? 1 : 0;
}

int rhs_fused3_variant(int mode, int n_prev) {
    if (mode == F3_PLAIN) return 1;
    if (mode == F3_COMBINE) return n_prev >= 0 && n_prev <= kF3MaxPrev;
    if (mode == F3_RK4) return n_prev >= 0 && n_prev <= 3;
    return mode == F3_ERROR && (n_prev == kF3MaxPrev || n_prev == 1);
}

// RkOpt::xmask / s_out (the two halves of odeint_adjoint's right-hand side): the dopri5 launches - COMBINE with 1..4 earlier stages,
// ERROR with 5
int rhs_adj_variant(int mode, int n_prev) {
    return (mode == F3_COMBINE && n_prev >= 1 && n_prev <= 4) || (mode == F3_ERROR && n_prev == 5);
}

int rhs_adj_supported(const ndcn_csr *A, int H, uint32_t flags, int mode, int n_prev) {
    static const int enabled = env_int_f3("NDCN_F3_ADJ", 1);
    if (!enabled || H != 256 || (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) || !rhs_fused3_supported(A)) return 0;
    if (A->hub_n > 0 || A->sweep_S) return 0;
    return rhs_adj_variant(mode, n_prev);
}

// RkOpt::xadd: the lattice plan, no halo panel, the launch that opens a dopri5 step (COMBINE with one earlier stage: k2) and the
// second evaluation of the initial step (ERROR with one earlier stage: f(y0 + h0 f0) carrying the d2 norm)
int rhs_xadd_supported(const ndcn_csr *A, int H, uint32_t flags, int mode, int n_prev) {
    static const int enabled = env_int_f3("NDCN_F3_XADD", 1);
    if (!enabled || H != 256 || (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) || !rhs_fused3_supported(A)) return 0;
    if (A->hub_n > 0) return 0;                                          // no second panel (hub sums); the caller vouches that
                                                                         // every column lies in the panels it passes (no X_halo)
    return ((mode == F3_COMBINE || mode == F3_ERROR) && n_prev == 1) ? 1 : 0;
}

template <bool HALO, int MODE, int NP, int XOP = 0, bool NT = true, bool SOUT = false, bool NOK = false>
static int launch_f3(const F3Args &a, const F3Epi &e, dim3 grid, hipStream_t st) {
    auto kern = rhs_fused3_kernel<HALO, MODE, NP, XOP, NT, SOUT, NOK>;
    static std::atomic<unsigned long long> attr_seen{0};
    if (once_per_device(attr_seen)) {
        NDCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kF3Lds));
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * (f3_producers(MODE, NP) + kF3WM)), kF3Lds, st, a, e);
    return NDCN_OK;
}

// Wq: the split weights of pack_weight_256 (Wp + 256 * 256 floats); arguments as rhs_fused2_f32
int rhs_fused3_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const void *Wq, const float *b, float *K,
                   uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev,
                   float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const RkOpt *opt) {
    if (A->n_rows == 0) return NDCN_OK;
    if (!rhs_fused3_variant(mode, n_prev)) { set_error("rhs_fused3: no kernel for mode %d with %d previous stages", mode, n_prev); return NDCN_EINVAL; }
    F3Args a;
    a.rec = A->rec; a.n_groups = A->rec_groups; a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.Wq = Wq; a.bias = b; a.K = K;
    // (the hint is honoured where a kernel without the store exists: the launches the fixed grids issue with it)
    const bool skip_k = opt && opt->no_k && y_next && !opt->xmask && !opt->xadd && !opt->s_out && !(opt->y_aux && opt->c_aux) &&
                        ((mode == F3_COMBINE && n_prev == 0) || (mode == F3_RK4 && n_prev == 3));
    const bool masked = opt && opt->xmask;
    a.Xadd = masked ? opt->xmask : ((opt && opt->xadd) ? opt->xadd : nullptr);
    a.xadd_c = (a.Xadd && !masked) ? opt->xadd_c : 0.f;
    a.S_out = (opt && opt->s_out) ? opt->s_out : nullptr;
    if (a.Xadd && !masked && (Xh || !((mode == F3_COMBINE || mode == F3_ERROR) && n_prev == 1))) {
        set_error("rhs_fused3: X + c Xadd is formed in the one-stage COMBINE / ERROR launches of an operator without a halo panel");
        return NDCN_EINVAL;
    }
    if ((masked || a.S_out) && (Xh || (masked && a.S_out) || (masked && opt->xadd) || !rhs_adj_variant(mode, n_prev))) {
        set_error("rhs_fused3: the masked input / the S output exist for the dopri5 launches of an operator without a halo panel (rhs_adj_supported)");
        return NDCN_EINVAL;
    }
    a.relu = (flags & NDCN_F_RELU) ? 1 : 0;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
    static const int dbg = env_int_f3("NDCN_FUSED3_DBG", 0);
    a.dbg = dbg;
    static const int timing = kF3Timing ? env_int_f3("NDCN_FUSED3_TIMING", 0) : 0;
    static unsigned long long *d_cyc = nullptr;
    static int timing_prints = 0;
    if (timing && !d_cyc) (void)hipMalloc(&d_cyc, (size_t)kCus * kF3WavesMax * 2 * sizeof(unsigned long long));
    a.dbg_cycles = timing ? d_cyc : nullptr;
    F3Epi e = {};
    e.y0 = y0; e.y_next = y_next; e.rtol = rtol; e.atol = atol; e.partials = static_cast<double *>(d_ws);
    e.y1 = (opt && opt->y1) ? opt->y1 : X;
    e.y_aux = (mode == F3_COMBINE && opt && opt->y_aux && opt->c_aux) ? opt->y_aux : nullptr;
    for (int m = 0; m <= kF3MaxPrev; ++m) e.c2[m] = (e.y_aux && m <= n_prev) ? opt->c_aux[m] : 0.f;
    for (int m = 0; m < kF3MaxPrev; ++m) e.kprev[m] = (m < n_prev && h_kprev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kF3MaxPrev; ++m) e.c[m] = (mode != F3_PLAIN && mode != F3_RK4 && m <= n_prev) ? h_c[m] : 0.f;
    if (mode == F3_RK4) e.c[0] = h_c[0];
    int per_xcd = kCus / kXcds;
    const int need = (a.n_groups + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds);
    const double P = 4.0 * 256 * (double)A->n_rows;
    double bytes = 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * 256 * (double)(A->n_rows + A->n_cols) + 4.0 * 256 * 256;
    if (mode != F3_PLAIN) bytes += P * (n_prev + 2);
    if (e.y_aux) bytes += P;
    if (a.Xadd) bytes += P;                                          // the second gather
    if (a.S_out) bytes += P;
    if (skip_k) bytes -= P;
    ProfScope prof(masked ? PROF_RHS_ADJ_T : (a.S_out ? PROF_RHS_ADJ_FWD : PROF_RHS_FUSED), st, bytes, 2.0 * A->nnz * 256 + 2.0 * (double)A->n_rows * 256 * 256);
    int rc = NDCN_OK;
    // panels that fit the Infinity Cache with room for the next launch's (<= 128 MiB): plain stores (operators without a halo panel)
    const bool cached = !Xh && (int64_t)A->n_rows * 1024 <= (128ll << 20);
#define NDCN_F3(HALO_, MODE_, NP_)                                                        \
    do {                                                                                  \
        if (!HALO_ && cached) rc = launch_f3<false, MODE_, NP_, 0, false>(a, e, grid, st); \
        else rc = launch_f3<HALO_, MODE_, NP_>(a, e, grid, st);                           \
    } while (0)
#define NDCN_F3_DISPATCH(HALO_)                                       \
    do {                                                              \
        if (mode == F3_PLAIN) NDCN_F3(HALO_, F3_PLAIN, 0);            \
        else if (mode == F3_ERROR) { if (n_prev == 1) NDCN_F3(HALO_, F3_ERROR, 1); else NDCN_F3(HALO_, F3_ERROR, 5); } \
        else if (mode == F3_RK4) switch (n_prev) {                    \
            case 0: NDCN_F3(HALO_, F3_RK4, 0); break;                 \
            case 1: NDCN_F3(HALO_, F3_RK4, 1); break;                 \
            case 2: NDCN_F3(HALO_, F3_RK4, 2); break;                 \
            default: NDCN_F3(HALO_, F3_RK4, 3); break;                \
        }                                                             \
        else switch (n_prev) {                                        \
            case 0: NDCN_F3(HALO_, F3_COMBINE, 0); break;             \
            case 1: NDCN_F3(HALO_, F3_COMBINE, 1); break;             \
            case 2: NDCN_F3(HALO_, F3_COMBINE, 2); break;             \
            case 3: NDCN_F3(HALO_, F3_COMBINE, 3); break;             \
            case 4: NDCN_F3(HALO_, F3_COMBINE, 4); break;             \
            default: NDCN_F3(HALO_, F3_COMBINE, 5); break;            \
        }                                                             \
    } while (0)
#define NDCN_F3_ADJ(XOP_, SOUT_)                                                                   \
    do {                                                                                            \
        if (mode == F3_ERROR) { if (cached) rc = launch_f3<false, F3_ERROR, 5, XOP_, false, SOUT_>(a, e, grid, st); else rc = launch_f3<false, F3_ERROR, 5, XOP_, true, SOUT_>(a, e, grid, st); } \
        else switch (n_prev) {                                                                      \
            case 1: if (cached) rc = launch_f3<false, F3_COMBINE, 1, XOP_, false, SOUT_>(a, e, grid, st); else rc = launch_f3<false, F3_COMBINE, 1, XOP_, true, SOUT_>(a, e, grid, st); break; \
            case 2: if (cached) rc = launch_f3<false, F3_COMBINE, 2, XOP_, false, SOUT_>(a, e, grid, st); else rc = launch_f3<false, F3_COMBINE, 2, XOP_, true, SOUT_>(a, e, grid, st); break; \
            case 3: if (cached) rc = launch_f3<false, F3_COMBINE, 3, XOP_, false, SOUT_>(a, e, grid, st); else rc = launch_f3<false, F3_COMBINE, 3, XOP_, true, SOUT_>(a, e, grid, st); break; \
            default: if (cached) rc = launch_f3<false, F3_COMBINE, 4, XOP_, false, SOUT_>(a, e, grid, st); else rc = launch_f3<false, F3_COMBINE, 4, XOP_, true, SOUT_>(a, e, grid, st); break; \
        }                                                                                           \
    } while (0)
    if (skip_k) {
        if (mode == F3_COMBINE) {
            if (Xh) rc = launch_f3<true, F3_COMBINE, 0, 0, true, false, true>(a, e, grid, st);
            else if (cached) rc = launch_f3<false, F3_COMBINE, 0, 0, false, false, true>(a, e, grid, st);
            else rc = launch_f3<false, F3_COMBINE, 0, 0, true, false, true>(a, e, grid, st);
        } else {
            if (Xh) rc = launch_f3<true, F3_RK4, 3, 0, true, false, true>(a, e, grid, st);
            else if (cached) rc = launch_f3<false, F3_RK4, 3, 0, false, false, true>(a, e, grid, st);
            else rc = launch_f3<false, F3_RK4, 3, 0, true, false, true>(a, e, grid, st);
        }
    } else if (masked) NDCN_F3_ADJ(2, false);
    else if (a.S_out) NDCN_F3_ADJ(0, true);
    else if (a.Xadd) {
        if (mode == F3_COMBINE) rc = launch_f3<false, F3_COMBINE, 1, 1>(a, e, grid, st);
        else rc = launch_f3<false, F3_ERROR, 1, 1>(a, e, grid, st);
    } else if (Xh) NDCN_F3_DISPATCH(true);
    else NDCN_F3_DISPATCH(false);
#undef NDCN_F3_DISPATCH
#undef NDCN_F3_ADJ
#undef NDCN_F3
    if (rc) return rc;
    NDCN_LAUNCH_CHECK();
    if (timing && timing_prints < timing) {                          // debugging aid: cycle accounting of workgroup 100
        (void)hipStreamSynchronize(st);
        const int wp = f3_producers(mode, n_prev), nw = wp + kF3WM;
        unsigned long long h[2 * kF3WavesMax];
        (void)hipMemcpy(h, d_cyc + (size_t)100 * 2 * nw, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double pw = 0, pq = 0, pd = 0, mw = 0, mq = 0;
        for (int w = 0; w < wp; ++w) { pw += h[2 * w] / (double)wp; pq += (h[2 * w + 1] & 0xffffffffull) / (double)wp; pd += (h[2 * w + 1] >> 32) / (double)wp; }
        for (int w = wp; w < nw; ++w) { mw += h[2 * w] / (double)kF3WM; mq += h[2 * w + 1] / (double)kF3WM; }
        fprintf(stderr, "[fused3 timing] mode %d n_prev %d block 100: producer waves work %.0f dma-wait %.0f barrier %.0f | mfma waves work %.0f barrier %.0f\n",
                mode, n_prev, pw, pd, pq, mw, mq);
        ++timing_prints;
    }
    if (mode == F3_ERROR) return partials_finish(e.partials, (int)grid.x * f3_producers(mode, n_prev), d_out, st, (opt && opt->accum) ? 1 : 0);
    return NDCN_OK;
}

}  // namespace ndcn
