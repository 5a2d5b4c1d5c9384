// Fused ODEFunc right-hand side with Runge-Kutta epilogue (H = 256):
//
//     K = relu((A X) W^T + b)          [neural_dynamics.py:27-36]
//   + optionally, in the same pass, the Runge-Kutta algebra that consumes K:
//       COMBINE :  y_next = y0 + sum_m c_m k_m   (k_new last)      [rk_common.py:51 / misc.py:22-25]
//       ERROR   :  sum ((sum_m c_m k_m) / (atol + rtol max(|y0|, |y1|)))^2 and the non-finite count of y1
//                                                                  [rk_common.py:60, misc.py:146-157, dopri5.py:101]
//       RK4     :  the input of the next stage of the 3/8-rule step, or the step itself, in the reference's operator
//                  order (stage index = number of earlier stages)  [rk_common.py:72-78, solvers.py:92]
//
// One persistent 1024-thread workgroup per CU: 4 MFMA waves (one per SIMD, fp32 MFMA) + 12 gather waves; 64-row
// tiles; LDS = S[2][64][260] fp32 (133 120 B).
//
//   phase A(t): MFMA waves    MFMA on S[t&1]                            -> K_t in registers
//               gather waves  one row at a time: issue the neighbour-row fetches of a row of S_{t+1} = (A X)[tile t+1],
//                             request the row-local RK panels of the same row of K_{t-1} (K_{t-1} sits in S[(t-1)&1]),
//                             fold the gathered rows, finish the epilogue (K row + RK algebra streamed to HBM as whole
//                             1 KiB rows) and drop the gathered row into S[(t-1)&1]
//   barrier
//   phase B(t): MFMA waves    K_t (+bias, relu) -> S[t&1], row-major
//   barrier
// A gather wave owns the same rows of every tile (p, p+12, ... < 64): it first reads K out of them, then overwrites
// them with the gathered S, so phase A needs no synchronisation among the gather waves.
//
// Measured lessons built in (1M-node grid, MI355X; profiles/r01*):
//   * fp32 MFMA waves slow every other wave on their SIMD (2x on the same gather code): per-neighbour work is kept
//     off the VALU (scalar-loaded CSR entries, buffer SGPR offsets, descriptors built on the scalar unit);
//   * a gather wave's chain is latency-bound, throughput comes from the number of chains: 12 gather waves, one row
//     (one fetch round for rows < 16 entries) each;
//   * vector memory returns in order and hipcc's waitcnt insertion cannot count fetches issued under run-time
//     conditions: fetches / stores of the gather waves are issued from inline asm and awaited by hand (gather.h);
//   * the weight operands sit in a 4-slot register ring refilled right after use (no operand copies), through a
//     buffer descriptor; accumulators leave through LDS as full rows instead of scattered 4-byte stores;
//   * launches with >= 3 earlier stages run at the HBM rate of a streaming kernel (texture-path FIFO full half of
//     the time): the row-local panels are fetched non-temporal so that they do not evict the gathered panel.
#include <stdlib.h>

#include "kernels.h"
#include "gather.h"
#include "split16.h"

#pragma clang fp contract(off)   // the RK algebra below must round like the reference's separate mul / add ops

namespace ndcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH2 = 256;
constexpr int kTile2 = 64;
constexpr int kLd2 = kH2 + 4;              // +4 floats: row m starts on 16-byte slot (4 m) mod 64 -> conflict-free b128
constexpr int kTileFloats2 = kTile2 * kLd2;
// NDCN_SPLIT = 1 (default): the dense 256 x 256 product runs on the fp16 matrix cores with fp32-grade results - every operand
// split error-free into two fp16 pieces behind a power-of-two scale, three partial products accumulated in fp32 (split16.h:
// measured against fp64, 1.9e-7 of sum |s w| vs 2.3e-7 for the fp32-MFMA chain on the same data) at 3/16 of the fp32 MFMA's
// matrix-pipe time.  NDCN_SPLIT = 0 builds the fp32-MFMA consumer (A/B reference).
// NDCN_F2_EXACT = 1 (csrc/rhs_fused2_exact.hip includes this file with it): the SAME kernel with the fp32-MFMA consumer under other names -
// rhs_fused2_exact_kernel / rhs_fused2_exact_f32 - the route of the range guard (rhs.hip: NDCN_PATH_EXACT32): gather, Linear on
// v_mfma_f32_32x32x2_f32 over the fp32 image of W, RK epilogue, one launch.  The definitions both builds would share stay in the default one.
#ifndef NDCN_F2_EXACT
#define NDCN_F2_EXACT 0
#endif
#if NDCN_F2_EXACT
#undef NDCN_SPLIT
#define NDCN_SPLIT 0
#define rhs_fused2_kernel rhs_fused2_exact_kernel
#define rhs_fused2_f32 rhs_fused2_exact_f32
#endif
#ifndef NDCN_SPLIT
#define NDCN_SPLIT 1
#endif
// gather waves: 12 beside the fp32 MFMA waves (16 waves per CU, 128 registers each); 8 beside the split consumer, whose
// four-k-step weight ring needs ~160 registers (12 waves per CU, 168 each)
constexpr int kProd = NDCN_SPLIT ? 8 : 12;
constexpr int kWaves = 4 + kProd;
constexpr int kRowsPerProd = (kTile2 + kProd - 1) / kProd;  // rows p, p+12, ... < 64 of every tile: 6 (p < 4) or 5
constexpr int kMaxPrev = 5;              // dopri5 needs at most 5 earlier stages with a non-zero coefficient

// Arguments of the RK epilogue.  They are NOT read through the kernel-parameter object: the compiler would keep
// all of them (8 pointers + 8 scalars) live in SGPRs across the gather loops, whose scalar CSR loads already
// fill the SGPR file - the COMBINE / ERROR variants then spilled 79 SGPRs into VGPR lanes (v_writelane /
// v_readlane = VALU work inside the loops that share a SIMD with the fp32 MFMA wave: +1.0 M cycles per launch,
// measured with every epilogue operation disabled).  The epilogue re-reads them from the kernarg segment once
// per tile instead (a handful of s_loads), through a laundered pointer the compiler cannot hoist.
struct EpiArgs {
    const float *y0;
    const float *kprev[kMaxPrev];
    float *y_next;
    double *partials;                       // ERROR: [gridDim.x * kProd][2]
    float c[kMaxPrev + 1];                  // c[0 .. n_prev-1] for kprev, c[n_prev] for the new K
    int n_prev;
    float rtol, atol;
    const float *y1;                        // ERROR: the state of the error record, by row of this launch
    float *y_aux;                           // COMBINE, nullable: second linear combination (no y0), coefficients c2[]
    float c2[kMaxPrev + 1];
};

struct Fused2Args {
    const float *X, *Xh;
    unsigned x_bytes, xh_bytes;             // sizes of the gathered panels (buffer descriptors: < 2^32)
    int n_own;
    const float *Wp, *bias;
    const void *Wq;                         // split weights (pack_weight_256: two fp16 planes in MFMA B-operand order + scales)
    float *K;                               // relu(...) output panel
    int n_rows, n_tiles, relu;
    const int *tile_order;                  // nullable: walk position -> 64-row tile (ndcn_csr::tile_order)
    unsigned long long *dbg_cycles;         // NDCN_FUSED_TIMING: per (block, wave) {work cycles, barrier-wait cycles}
    int dbg;                                // cycle-accounting experiments only (NDCN_FUSED_DBG): 1 skip MFMA, 2 skip gather,
                                            // 4 skip epilogue, 64 no weight refills, 8192 report MFMA loop | dump separately
};
typedef const __attribute__((address_space(4))) EpiArgs *EpiPtr;
// kernel parameters: rowptr, colidx, val (3 x 8 bytes), Fused2Args, EpiArgs - both structs are 8-aligned
constexpr int kEpiKernargOffset = 24 + (int)((sizeof(Fused2Args) + 7) / 8 * 8);

enum { MODE_PLAIN = 0, MODE_COMBINE = 1, MODE_ERROR = 2, MODE_RK4 = 3 };



// NP = number of earlier stages in the RK sum (compile-time: the number of fetches in flight must be known for the
// waits on the gathered rows not to include the younger, slower row-local fetches)
template <bool HALO, int MODE, int NP>
__global__ __launch_bounds__(64 * kWaves) void rhs_fused2_kernel(const int *__restrict__ rowptr,
                                                                      const int *__restrict__ colidx,
                                                                      const float *__restrict__ val, Fused2Args a,
                                                                      EpiArgs epi_by_kernarg_only) {
    (void)epi_by_kernarg_only;
    auto epi_args = [&]() -> EpiPtr {
        auto base = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        EpiPtr e = (EpiPtr)(base + kEpiKernargOffset);
        asm volatile("" : "+s"(e));                           // a fresh pointer per call: loads stay where they are used
        return e;
    };
    __shared__ __attribute__((aligned(16))) float s_tile[2 * kTileFloats2];
    // per S-tile row: the power-of-two scale of the fp16 split (split16.h) and 1 / (row scale * weight scale)
    __shared__ __attribute__((aligned(16))) float s_sc[2 * kTile2], s_un[2 * kTile2];
    const float *w_unscale = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.Wq) + kS16Bytes);   // [256]: per output column

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int p = wave - 4;                                   // producer index 0..11
    const f32x4 *X = reinterpret_cast<const f32x4 *>(a.X);
    const f32x4 *Wp = reinterpret_cast<const f32x4 *>(a.Wp);
    // buffer descriptors of the gathered panels (wave-uniform; byte sizes < 2^31 checked by the launcher)
    const u32x4 rsX = make_rsrc(a.X, a.x_bytes);
    const u32x4 rsH = make_rsrc(HALO ? a.Xh : a.X, HALO ? a.xh_bytes : a.x_bytes);
    const int lane_off = lane * 16;

    // tiles of this workgroup: XCD x owns a contiguous chunk; its workgroups take tiles round-robin
    const int xcd = blockIdx.x % kXcds;
    const int wg = blockIdx.x / kXcds;
    const int wgs_per_xcd = gridDim.x / kXcds;
    const int chunk = (a.n_tiles + kXcds - 1) / kXcds;
    const int t_lo = xcd * chunk, t_hi = min(a.n_tiles, t_lo + chunk);
    const int t_first = t_lo + wg;
    const int my_tiles = t_first < t_hi ? (t_hi - t_first + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) {                                      // uniform per workgroup
        if (MODE == MODE_ERROR && producer && lane == 0) {    // the finish kernel sums EVERY slot
            double *partials = epi_args()->partials;
            partials[2 * (blockIdx.x * kProd + p)] = 0.0;
            partials[2 * (blockIdx.x * kProd + p) + 1] = 0.0;
        }
        return;
    }

    double err_sum = 0.0, err_bad = 0.0;                      // MODE_ERROR, producers

    // ---- producer: row extents by scalar loads, one row ahead (row r of the CSR: entries [rowptr[r], rowptr[r+1])).
    // They used to come from one vector load per tile read out with v_readlane: hipcc then put s_waitcnt vmcnt(0) in
    // front of every row - i.e. each row first waited for the previous row's STORES to be acknowledged.
    struct RowExt { int j, j1; };
    auto row_ext = [&](int r) -> RowExt {
        RowExt e{0, 0};
        if (r < a.n_rows) { e.j = rowptr[r]; e.j1 = rowptr[r + 1]; }
        return e;
    };

    // ---- producer: RK epilogue of one K row ----------------------------------------------------------
    // Row-local panels go through buffer descriptors built on the scalar unit: address = descriptor base + SGPR row
    // offset (r << 10) + the fixed per-lane offset - no VALU address arithmetic per panel (the producers' VALU time
    // is what the MFMA waves squeeze).  Panels are < 2 GiB (launcher check).
    struct EpiRow { f32x4 km[NP > 0 ? NP : 1]; f32x4 y0v, y1v; };
    const unsigned panel_bytes = (unsigned)a.n_rows << 10;         // panels < 4 GiB (launcher check)
    auto ldp = [&](const float *base, unsigned row_off) {
        asm volatile("" : "+s"(base));      // keep the 4-SGPR descriptor transient: hoisted descriptors for 8 panels spill
        // read once: non-temporal (measured: 7.46 -> 6.77 GB read per 4-stage COMBINE launch, 12.36 -> 12.18 ms/step)
        return fetch128_stream(make_rsrc(base, panel_bytes), lane_off, row_off);
    };
    auto stp = [&](float *base, unsigned row_off, f32x4 v) {
        asm volatile("" : "+s"(base));
        store128(v, make_rsrc(base, panel_bytes), lane_off, row_off);
    };
    auto epi_load = [&](EpiPtr ea, int r, EpiRow &e) {
        const unsigned off = (unsigned)r << 10;
#pragma unroll
        for (int m = 0; m < NP; ++m) e.km[m] = ldp(ea->kprev[m], off);
        e.y0v = ldp(ea->y0, off);
        if (MODE == MODE_ERROR) e.y1v = ldp(ea->y1, off);      // the input of this evaluation is y1 (own rows); measured:
                                                               // non-temporal here too reads less (8.69 vs 9.01 GB)
    };
    auto epi_finish = [&](EpiPtr ea, int r, const float *src_row, const EpiRow &e) {
        const f32x4 kn = *reinterpret_cast<const f32x4 *>(src_row + 4 * lane);
        const unsigned off = (unsigned)r << 10;
        stp(a.K, off, kn);
        if (MODE == MODE_PLAIN) return;
        if (MODE == MODE_RK4) {
            // rk4_alt_step_func (rk_common.py:72-78), same operator order as fixed_stage_kernel ops 2-5
            const float dt = ea->c[0];
            f32x4 s;
            if (NP == 0) s = (kn * dt) / 3.f;                                            // y + dt*k1/3
            else if (NP == 1) s = (e.km[0] / -3.f + kn) * dt;                            // y + dt*(k2 - k1/3)
            else if (NP == 2) s = ((e.km[0] - e.km[1]) + kn) * dt;                       // y + dt*(k1 - k2 + k3)
            else s = (((e.km[0] + e.km[1] * 3.f) + e.km[2] * 3.f) + kn) * (dt / 8.f);    // y + (k1+3k2+3k3+k4)*dt/8
            stp(ea->y_next, off, e.y0v + s);
            return;
        }
        // sum of the stages left to right, the new one last (misc.py:22-25), each product rounded on its own
        f32x4 s = kn * ea->c[NP];
        if (NP > 0) {
            f32x4 u = e.km[0] * ea->c[0];
#pragma unroll
            for (int m = 1; m < NP; ++m) u = u + e.km[m] * ea->c[m];
            s = u + s;
        }
        if (MODE == MODE_COMBINE) {
            stp(ea->y_next, off, e.y0v + s);
            if (ea->y_aux) {                                       // wave-uniform; the store is OLDER than every later fetch
                f32x4 w2 = kn * ea->c2[NP];
                if (NP > 0) {
                    f32x4 u2 = e.km[0] * ea->c2[0];
#pragma unroll
                    for (int m = 1; m < NP; ++m) u2 = u2 + e.km[m] * ea->c2[m];
                    w2 = u2 + w2;
                }
                stp(ea->y_aux, off, w2);
            }
        } else {
            const float rtol = ea->rtol, atol = ea->atol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tol = atol + rtol * max_nan(fabsf(e.y0v[q]), fabsf(e.y1v[q]));
                const float z = s[q] / tol;
                err_sum += (double)(z * z);
                err_bad += (double)(int)(!(fabsf(e.y1v[q]) <= 3.402823466e38f));
            }
        }
    };
    // ---- producer phase A, one row at a time -------------------------------------------------------------
    //   1. issue the neighbour-row fetches of row k of the NEXT tile (extents prefetched): the whole row if it
    //      has < 16 entries, after whole batches of 8 otherwise
    //   2. request the row-local panels (y0, earlier stages) of row k of tile t_epi, whose K sits in `buf`; they
    //      are YOUNGER than (1), so the wait for the gathered rows (vector memory returns in order) does not
    //      include their HBM latency - they land while (3) runs
    //   3. fold the gathered rows (2 packed FMAs per neighbour)
    //   4. epilogue of the row: K out of `buf`, RK algebra, stores
    //   5. drop the gathered row into the same row of `buf`
    // A gather wave owns the same rows of every tile, so no producer-to-producer synchronisation is needed.
    // Twelve waves work like this per CU: a wave's chain (scalar index load -> fetches -> FMAs under MFMA
    // contention -> stores) is latency-bound - with half of the 8 waves of the previous version switched off the
    // others were not a cycle faster - so throughput comes from the number of chains in flight.
    auto producer_phase = [&](float *buf, int t_epi, int t_next, bool do_epi, bool do_gather) {
        EpiPtr ea = nullptr;
        if (MODE != MODE_PLAIN) ea = epi_args();               // one (laundered) read of the epilogue arguments per tile
        // walk positions -> tiles (a locality hint: an XCD's concurrent tiles then cover a compact lattice block)
        if (a.tile_order) {
            if (do_epi) t_epi = a.tile_order[t_epi];
            if (do_gather) t_next = a.tile_order[t_next];
        }
        const int r0 = t_epi * kTile2 + p, g0 = t_next * kTile2 + p;
        RowExt ext = do_gather ? row_ext(g0) : RowExt{0, 0};
        // The row-local panels of row k + 1 are requested while row k is being finished (two register sets, ping-pong):
        // they are YOUNGER than row k's neighbour fetches, so the wait for those leaves them in flight, and OLDER than
        // row k + 1's - complete, by in-order return, when that row's neighbours are.  A row then costs
        // max(gather latency, panel latency) instead of their sum.
        constexpr int kEpiFetches = MODE == MODE_PLAIN ? 0 : NP + 1 + (MODE == MODE_ERROR ? 1 : 0);
        EpiRow e[2];
        if (MODE != MODE_PLAIN && do_epi && r0 < a.n_rows) epi_load(ea, r0, e[0]);
#pragma unroll
        for (int k = 0; k < kRowsPerProd; ++k) {
            const int lr = p + kProd * k;
            if (lr >= kTile2) break;                           // wave-uniform: waves 4.. own one row less
            const int r = r0 + kProd * k;
            const RowExt cur = ext;
            if (do_gather && lr + kProd < kTile2) ext = row_ext(g0 + kProd * (k + 1));
            const bool ep = do_epi && r < a.n_rows;
            const bool ep_next = do_epi && lr + kProd < kTile2 && r + kProd < a.n_rows;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 x[16];
            float w[16];
            int m = 0;
            if (do_gather) {
                int j = cur.j;
                const int j1 = cur.j1;
                for (; j1 - j >= 16; j += 16) {                // long rows: whole rounds of 16 fetches first
                    g_issue<8, HALO, 0>(colidx, val, j, rsX, rsH, a.n_own, lane_off, x, w);
                    g_issue<8, HALO, 8>(colidx, val, j + 8, rsX, rsH, a.n_own, lane_off, x, w);
                    wait_vmcnt<0>();
                    g_accum<8, 0>(x, w, acc);
                    g_accum<8, 8>(x, w, acc);
                }
                m = j1 - j;
                g_row_issue<HALO>(colidx, val, j, m, rsX, rsH, a.n_own, lane_off, x, w);
            }
            // the next row's panels are the kEpiFetches youngest fetches: this row's neighbours (and panels) are
            // complete when at most that many are outstanding
            if (MODE != MODE_PLAIN && ep_next) { epi_load(ea, r + kProd, e[(k + 1) & 1]); wait_vmcnt<kEpiFetches>(); }
            else wait_vmcnt<0>();
            if (do_gather) g_row_accum(m, x, w, acc);
            if (ep) {
                if (MODE != MODE_PLAIN) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) tie(e[k & 1].km[i]);
                    tie(e[k & 1].y0v);
                    if (MODE == MODE_ERROR) tie(e[k & 1].y1v);
                }
                epi_finish(ea, r, buf + lr * kLd2, e[k & 1]);
            }
            if (do_gather) {
                *reinterpret_cast<f32x4 *>(buf + lr * kLd2 + 4 * lane) = acc;
                // the row's scale for the fp16 split: a power of two from its largest magnitude (this wave holds the row)
                unsigned sb, ub;
                s16_scale_bits(s16_wave_umax(s16_row_max_bits(acc)), sb, ub);
                if (lane == 0) {
                    const int slot = (buf == s_tile ? 0 : kTile2) + lr;
                    s_sc[slot] = __builtin_bit_cast(float, sb);
                    s_un[slot] = __builtin_bit_cast(float, ub);
                }
            }
        }
    };

    // (Measured and dropped, profiles/r02g_fused_timing.txt: TWO rows in flight per gather wave - the next row's <= 12
    // neighbour fetches, branch-free through null descriptors, issued before the wait for the current row's.  Bit-equal,
    // but slower: plain launch 1.05 -> 1.21 ms, one / two earlier stages 1.38 -> 1.52 / 1.48 -> 1.59 ms.  The gather
    // phase takes the same 1.7 M cycles per launch with the MFMA waves switched off and with twice the fetches in
    // flight: it is bound by the vector-memory path's rate for 1 KiB row fetches (9 per output row = 9 GB per launch
    // L2 -> CU at ~20 B/clk/CU), not by a wave's latency chain.  The lever is FEWER fetches per row - the record plan
    // of spmm_rec.hip stages each distinct neighbour row once per 16-row group - which needs the LDS this kernel spends
    // on its two 64-row S tiles.)

    // ---- consumer -----------------------------------------------------------------------------------
    // wave w owns output columns [64 w, 64 w + 64) (n-tiles 2w, 2w+1) for both 32-row m-tiles
    f32x16 acc00, acc01, acc10, acc11;
    // Ring of weight operands, kRing deep: slot u holds the k-quads q with q % kRing == u.  The weights are the same
    // for every tile, so the ring runs on across tiles: after the MFMAs of quad q its slot is refilled with quad
    // q + kRing of the same tile or, for the last kRing quads, with quad (q % kRing) of the next tile (kRing need not
    // divide 32: the 32 quads are fully unrolled, every slot index is static).
#ifndef NDCN_RING
#define NDCN_RING 4
#endif
    constexpr int kRing = NDCN_RING;
    // packed weights through a buffer descriptor: per-lane offset in one VGPR, the (n-tile, k-quad) block offset on
    // the scalar unit - 64 distinct block addresses per tile would otherwise each cost a 64-bit VGPR pair
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Wp), 0, kH2 * kH2 * 4, 0x00020000);
    const int w_slab = (2 * (wave & 3)) * 32 * 1024;           // bytes; n-tiles 2w, 2w+1 of this wave
    auto ldw = [&](int ntile, int quad) {
        int ws = w_slab;
        asm volatile("" : "+s"(ws));        // one s_add at the use instead of 64 hoisted (and spilled) block offsets
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, lane_off, ws + (ntile * 32 + quad) * 1024, 0));
    };
    f32x4 r0[kRing], r1[kRing];
    auto ring_fill = [&]() {
#pragma unroll
        for (int u = 0; u < kRing; ++u) { r0[u] = ldw(0, u); r1[u] = ldw(1, u); }
    };
    auto mfma_tile = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }
        const float *a0p = src + (lane & 31) * kLd2 + 128 * (lane >> 5);
        const float *a1p = a0p + 32 * kLd2;
        // A operands (LDS) are fetched half a k-quad (2 k-steps = 8 MFMAs = 512 cycles) ahead and ping-pong between two
        // 8-byte register pairs per m-tile.  A weight slot is refilled right AFTER the MFMAs that read it (no operand
        // is ever copied aside), i.e. kRing - 1 quads = 4096 MFMA cycles before it is needed: with twelve gather waves
        // queueing on the same texture path the L2 latency of a weight fetch averaged ~3600 cycles (measured: the
        // MFMA loop ran at 2.09 M cycles per launch without refills, 3.3-3.5 M with a three-quad lead).
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 aX0 = *reinterpret_cast<const f32x2 *>(a0p), aX1 = *reinterpret_cast<const f32x2 *>(a1p);
        f32x2 aY0 = aX0, aY1 = aX1;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int u = q % kRing;
            // first half of the quad on aX while aY (second half) is fetched
            __builtin_amdgcn_sched_barrier(0);      // straight-line code: keep hipcc from hoisting later fetches up here
            aY0 = *reinterpret_cast<const f32x2 *>(a0p + 4 * q + 2);
            aY1 = *reinterpret_cast<const f32x2 *>(a1p + 4 * q + 2);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(aX0[e], r0[u][e], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(aX0[e], r1[u][e], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(aX1[e], r0[u][e], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(aX1[e], r1[u][e], acc11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 32) {
                aX0 = *reinterpret_cast<const f32x2 *>(a0p + 4 * (q + 1));
                aX1 = *reinterpret_cast<const f32x2 *>(a1p + 4 * (q + 1));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(aY0[e], r0[u][2 + e], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(aY0[e], r1[u][2 + e], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(aY1[e], r0[u][2 + e], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(aY1[e], r1[u][2 + e], acc11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(a.dbg & 64)) {                                                 // dbg 64: no weight refills (timing experiment)
                const int qn = q + kRing < 32 ? q + kRing : u;                   // next tile's quad for this slot
                r0[u] = ldw(0, qn);
                r1[u] = ldw(1, qn);
            }
        }
    };
#if NDCN_SPLIT
    // split weights (split16.h): block (n-tile j, k-step s of 16, plane p of 2) = 64 lanes x 16 bytes (8 fp16:
    // W[32 j + (lane & 31)][16 s + 8 (lane >> 5) + 0..7]); ring of kRingQ k-steps x {2 n-tiles of this wave} x 2 planes,
    // refilled right after use
    constexpr int kRingQ = 4;
    constexpr int kPl = kS16Planes;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.Wq), 0, kS16Bytes, 0x00020000);
    const int q_slab = (2 * (wave & 3)) * 16 * kPl * 1024;
    auto ldq = [&](int jj, int ks, int pl) {
        int ws = q_slab;
        asm volatile("" : "+s"(ws));
        return __builtin_amdgcn_raw_buffer_load_b128(rsQ, lane_off, ws + ((jj * 16 + ks) * kPl + pl) * 1024, 0);
    };
    u32x4 Bq[kRingQ][2][kPl];
    auto ring_fill_q = [&]() {
#pragma unroll
        for (int u = 0; u < kRingQ; ++u)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, u, pl);
    };
    // every fp32 S value times its row's power-of-two scale, split error-free into two fp16 pieces; three partial products
    // per k-step, small terms first (split16.h; the same sequence as rhs_fused3.hip: bit-equal results)
    auto mfma_tile_split = [&](const float *src, int tb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }
        const float *ap = src + (lane & 31) * kLd2 + 8 * (lane >> 5);
        const float sc[2] = {s_sc[tb * kTile2 + (lane & 31)], s_sc[tb * kTile2 + 32 + (lane & 31)]};
        f32x4 n0 = *reinterpret_cast<const f32x4 *>(ap), n1 = *reinterpret_cast<const f32x4 *>(ap + 4);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int u = ks % kRingQ;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 r0 = n0, r1 = n1;
                {   // the NEXT block's A values leave LDS while this block's products run
                    const int nk = mt == 0 ? ks : ks + 1, nm = mt ^ 1;
                    if (nk < 16) {
                        n0 = *reinterpret_cast<const f32x4 *>(ap + nm * 32 * kLd2 + 16 * nk);
                        n1 = *reinterpret_cast<const f32x4 *>(ap + nm * 32 * kLd2 + 16 * nk + 4);
                    }
                }
                u32x4 A0, A1;
                s16_split8(r0, r1, sc[mt], A0, A1);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x16 &acc = mt == 0 ? (jj == 0 ? acc00 : acc01) : (jj == 0 ? acc10 : acc11);
                    s16_mfma(acc, A1, Bq[u][jj][0]);
                    s16_mfma(acc, A0, Bq[u][jj][1]);
                    s16_mfma(acc, A0, Bq[u][jj][0]);
                }
            }
            if (!(a.dbg & 64)) {
                const int kn = ks + kRingQ < 16 ? ks + kRingQ : u;     // the last kRingQ k-steps refill their slot with k-step (slot index) of the NEXT tile
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, kn, pl);
            }
        }
    };
#endif
    auto dump_tile = [&](float *dst, int tb) {
        // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]; with the split product column n is multiplied back by
        // 1 / (its weight row's scale), then row m by 1 / (row scale) (16 rows per lane and m-tile: four 16-byte LDS reads)
#if NDCN_SPLIT
        // (a pass of its own: one more live register inside the loop below costs the kernel its third wave per SIMD)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float wu = w_unscale[64 * wave + 32 * n + (lane & 31)];
            if (n == 0) { acc00 *= wu; acc10 *= wu; } else { acc01 *= wu; acc11 *= wu; }
        }
#endif
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 un[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                un[q] = NDCN_SPLIT ? *reinterpret_cast<const f32x4 *>(s_un + tb * kTile2 + 32 * mt + 8 * q + 4 * (lane >> 5))
                                   : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = 64 * wave + 32 * n + (lane & 31);
                const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float o = ((mt == 0 ? (n == 0 ? acc00[r] : acc01[r]) : (n == 0 ? acc10[r] : acc11[r]))) * un[r >> 2][r & 3] + bv;
                    if (a.relu) o = relu_nan(o);
                    dst[m * kLd2 + col] = o;
                }
            }
        }
    };

    // Role-specialised loops (both execute the same sequence of workgroup barriers), so the accumulators of the
    // MFMA waves and the fetch registers of the gather waves never share a live range.
    unsigned long long cyc_work = 0, cyc_wait = 0;
    if (producer) {
        producer_phase(s_tile, 0, t_first, false, true);
        __syncthreads();                                       // S[0] ready
        for (int it = 0; it < my_tiles; ++it) {
            const int t = t_first + it * wgs_per_xcd;
            float *oth = s_tile + ((it & 1) ^ 1) * kTileFloats2;
            const unsigned long long c0 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            // phase A: K of the previous tile sits in `oth`; stream it out, then refill `oth` with the next S
            const bool do_gather = it + 1 < my_tiles && !(a.dbg & 2);
            producer_phase(oth, t - wgs_per_xcd, t + wgs_per_xcd, it > 0 && !(a.dbg & 4), do_gather);
            const unsigned long long c1 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            // phase B: consumers drop K_t into the tile they consumed
            __syncthreads();
            if (a.dbg_cycles) { cyc_work += c1 - c0; cyc_wait += __builtin_readcyclecounter() - c1; }
        }
        if (a.dbg_cycles && lane == 0) {
            a.dbg_cycles[2 * (blockIdx.x * kWaves + wave)] = cyc_work;
            a.dbg_cycles[2 * (blockIdx.x * kWaves + wave) + 1] = cyc_wait;
        }
        const int t_last = t_first + (my_tiles - 1) * wgs_per_xcd;
        producer_phase(s_tile + ((my_tiles - 1) & 1) * kTileFloats2, t_last, 0, true, false);
        if (MODE == MODE_ERROR) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                err_sum += __shfl_down(err_sum, off, 64);
                err_bad += __shfl_down(err_bad, off, 64);
            }
            if (lane == 0) {
                double *partials = epi_args()->partials;
                partials[2 * (blockIdx.x * kProd + p)] = err_sum;
                partials[2 * (blockIdx.x * kProd + p) + 1] = err_bad;
            }
        }
    } else {
#if NDCN_SPLIT
        ring_fill_q();                                         // weight fetches fly while the first tile is gathered
#else
        ring_fill();                                           // weight fetches fly while the first tile is gathered
#endif
        __syncthreads();                                       // S[0] ready
        for (int it = 0; it < my_tiles; ++it) {
            float *cur = s_tile + (it & 1) * kTileFloats2;
            const unsigned long long c0 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
#if NDCN_SPLIT
            if (!(a.dbg & 1)) mfma_tile_split(cur, it & 1);
#else
            if (!(a.dbg & 1)) mfma_tile(cur);
#endif
            const unsigned long long c1 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();                                   // every consumer is done reading `cur`
            const unsigned long long c2 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            dump_tile(cur, it & 1);
            const unsigned long long c3 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            if (a.dbg_cycles) {
                if (a.dbg & 8192) { cyc_work += c1 - c0; cyc_wait += c3 - c2; }     // split: MFMA loop | accumulator dump
                else { cyc_work += (c1 - c0) + (c3 - c2); cyc_wait += (c2 - c1) + (__builtin_readcyclecounter() - c3); }
            }
        }
        // The last refills of the weight ring (they wrap into a tile that does not exist) are still in flight here.
        // hipcc lays the gather-wave region out after this one and merges the two paths' waitcnt state: without this
        // explicit wait it "protects" v64-v95 in the gather code with vmcnt waits that - not counting the asm fetches
        // there - drain those instead.
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        if (a.dbg_cycles && lane == 0) {
            a.dbg_cycles[2 * (blockIdx.x * kWaves + wave)] = cyc_work;
            a.dbg_cycles[2 * (blockIdx.x * kWaves + wave) + 1] = cyc_wait;
        }
    }
}

#if !NDCN_F2_EXACT
// fixed-order sum of the per-producer partials (deterministic accept / reject)
__global__ __launch_bounds__(256) void fused2_finish_kernel(const double *__restrict__ partial, int n, double *__restrict__ out, int accum) {
    __shared__ double sa[256], sb[256];
    double s = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s += partial[2 * i]; bad += partial[2 * i + 1]; }
    sa[threadIdx.x] = s; sb[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { sa[threadIdx.x] += sa[threadIdx.x + w]; sb[threadIdx.x] += sb[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = accum ? out[0] + sa[0] : sa[0]; out[1] = accum ? out[1] + sb[0] : sb[0]; }
}

int partials_finish(const double *partials, int n, double *d_out, hipStream_t st, int accum) {
    hipLaunchKernelGGL(fused2_finish_kernel, dim3(1), dim3(256), 0, st, partials, n, d_out, accum);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

thread_local int g_last_rhs_path = 0;
#endif

static int env_int3(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

#if !NDCN_F2_EXACT
int rhs_fused2_supported(const ndcn_csr *A, int H, uint32_t flags) {
    static const int enabled = env_int3("NDCN_RHS_FUSED2", 1);
    if (!enabled || H != kH2 || !A) return 0;
    if (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) return 0;
    // every panel must fit a buffer descriptor with a 32-bit row offset: < 4 GiB, i.e. < 4 Mi rows of 1 KiB
    return (A->n_cols * (int64_t)kH2 * 4 < (1ll << 32) && A->n_rows * (int64_t)kH2 * 4 < (1ll << 32)) ? 1 : 0;
}

// compiled (mode, n_prev) variants: plain; COMBINE with 0..5 earlier stages; ERROR with dopri5's 5.  Anything else
// is served by the composition path of rhs_rk_f32 (same term order).
int rhs_fused2_variant(int mode, int n_prev) {
    if (mode == MODE_PLAIN) return 1;
    if (mode == MODE_COMBINE) return n_prev >= 0 && n_prev <= kMaxPrev;
    if (mode == MODE_RK4) return n_prev >= 0 && n_prev <= 3;
    return mode == MODE_ERROR && (n_prev == kMaxPrev || n_prev == 1);
}

int64_t rhs_fused2_partials_bytes() { return (int64_t)kCus * 12 * 2 * sizeof(double); }      // (12: the gather waves of the fp32-MFMA build)
#endif

// mode: 0 plain; 1 combine (y_next = y0 + sum c_m k_m, new K last); 2 error (d_out[0..1], d_ws scratch)
int rhs_fused2_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b,
                   float *K, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c,
                   int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st,
                   const RkOpt *opt) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    // Column-sweep plan (struct ndcn_csr; spmm_sweep.hip): S = A X by the sweep into the operator's scratch panel, then this
    // kernel on the IDENTITY operator over S - the Linear, the ReLU and the RK epilogue as on any operator (row i folds
    // fma(1, S[i], 0) = S[i]).  One profiled unit: the two launches are one evaluation of the right-hand side.
    if (!Xh && A->sweep_S && !(opt && (opt->xadd || opt->xmask || opt->s_out)) && spmm_sweep_supported(A, kH2) && aligned16(X)) {
        const double P = 4.0 * kH2 * (double)A->n_rows;
        double bytes = 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * kH2 * (double)(A->n_rows + A->n_cols) + 4.0 * kH2 * kH2;
        if (mode != MODE_PLAIN) bytes += P * (n_prev + 2 + ((mode == MODE_COMBINE && opt && opt->y_aux && opt->c_aux) ? 1 : 0));
        ProfScope prof(PROF_RHS_FUSED, st, bytes, 2.0 * A->nnz * kH2 + 2.0 * (double)A->n_rows * kH2 * kH2);
        const bool was_paused = prof_pause(true);
        int rc = spmm_sweep_f32(A, X, A->sweep_S, st);
        if (!rc) {
            ndcn_csr eye = {};
            eye.n_rows = eye.n_cols = eye.nnz = A->n_rows;
            eye.rowptr = A->sweep_eye_rowptr; eye.colidx = A->sweep_eye_colidx; eye.val = A->sweep_eye_val;
            if (A->sweep_eye_rec) { eye.rec = A->sweep_eye_rec; eye.rec_groups = A->sweep_eye_groups; eye.rec_rows = 16; eye.rec_cap = 40; eye.rec_kib = 2; }
            RkOpt o = opt ? *opt : RkOpt{};
            if (!o.y1) o.y1 = X;                          // ERROR mode: the state whose record is formed is the evaluation's input
            rc = rhs_fused2_f32(&eye, A->sweep_S, nullptr, A->n_rows, Wp, b, K, flags, mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol,
                                d_out, d_ws, st, &o);
            g_last_rhs_path |= NDCN_PATH_SWEEP;
        }
        prof_pause(was_paused);
        return rc;
    }
    // Long-row plan: the hub rows' (A X) rows are formed ahead by two small SpMMs (segments, then their sums) and the
    // kernel runs on the "light" operator that reads them as its second panel (include/ndcn_hip.h, struct ndcn_csr).
    ndcn_csr light;
    const ndcn_csr *A_full = A;
    // With a halo panel the plan still applies when the caller laid the hubs' rows out right BEHIND the halo rows (one
    // allocation: [halo | hub_S], what ndcn_csr_create's n_halo hint + sharding.py do for node-range shards of power-law graphs): the
    // kernel's second panel is then [halo | hubs], and the light operator's hub columns n_cols + h index into it as they are.
    const int64_t n_halo = Xh ? A->n_cols - n_own : 0;
    const bool hub_behind_halo = Xh && A->hub_S == Xh + n_halo * (int64_t)kH2;
    if (A->hub_n > 0 && (!Xh || hub_behind_halo) && A->hub_H == kH2 && A->hub_S && A->hub_Sseg) {
        ndcn_csr seg = {};
        seg.n_rows = A->hub_nseg; seg.n_cols = A->n_cols; seg.nnz = A->hub_nnz;
        seg.rowptr = A->hub_seg_rowptr; seg.colidx = A->hub_colidx; seg.val = A->hub_val;
        int rc = spmm_f32(&seg, X, Xh, Xh ? n_own : A->n_cols, A->hub_Sseg, kH2, 1.f, 0, st);
        if (rc) return rc;
        ndcn_csr cmb = {};
        cmb.n_rows = A->hub_n; cmb.n_cols = A->hub_nseg; cmb.nnz = A->hub_nseg;
        cmb.rowptr = A->hub_cmb_rowptr; cmb.colidx = A->hub_cmb_colidx; cmb.val = A->hub_cmb_val;
        rc = spmm_f32(&cmb, A->hub_Sseg, nullptr, A->hub_nseg, A->hub_S, kH2, 1.f, 0, st);
        if (rc) return rc;
        light = ndcn_csr{};
        light.n_rows = A->n_rows; light.n_cols = A->n_cols + A->hub_n; light.nnz = A->lt_nnz;
        light.rowptr = A->lt_rowptr; light.colidx = A->lt_colidx; light.val = A->lt_val;
        if (!Xh) {
            n_own = A->n_cols;
            Xh = A->hub_S;
        }
        A = &light;
    }
    const int path_bits = (A == &light ? NDCN_PATH_HUB : 0) | (Xh && A_full->n_cols > n_own ? NDCN_PATH_HALO : 0);
    // operators with the 16-row group-record plan (lattices): neighbour rows staged once per 4 x 4 patch (rhs_fused3.hip)
    if (NDCN_SPLIT && rhs_fused3_supported(A) && rhs_fused3_variant(mode, n_prev)) {
        g_last_rhs_path = NDCN_PATH_FUSED3 | path_bits;
        return rhs_fused3_f32(A, X, Xh, n_own, Wp + kH2 * kH2, b, K, flags, mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol, d_out,
                              d_ws, st, opt);
    }
    if (opt && (opt->xadd || opt->xmask || opt->s_out)) { set_error("rhs_fused2: RkOpt::xadd / xmask / s_out need the rhs_fused3 path (rhs_xadd_supported, rhs_adj_supported)"); return NDCN_EINVAL; }
    g_last_rhs_path = (NDCN_F2_EXACT ? NDCN_PATH_EXACT32 : NDCN_PATH_FUSED2) | path_bits;
    if (!rhs_fused2_variant(mode, n_prev)) { set_error("rhs_fused2: no kernel for mode %d with %d previous stages", mode, n_prev); return NDCN_EINVAL; }
    Fused2Args a;
    const int64_t xb = (Xh ? n_own : A->n_cols) * (int64_t)kH2 * 4, xhb = Xh ? (A->n_cols - n_own) * (int64_t)kH2 * 4 : 0;
    if (xb >= (1ll << 32) || xhb >= (1ll << 32) || (int64_t)n_rows * kH2 * 4 >= (1ll << 32)) {
        set_error("rhs_fused2: panel of %lld bytes exceeds the 4 GiB buffer-descriptor range", (long long)(xb > xhb ? xb : xhb));
        return NDCN_EINVAL;
    }
    a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.x_bytes = (unsigned)xb; a.xh_bytes = (unsigned)xhb; a.Wp = Wp; a.Wq = Wp + kH2 * kH2; a.bias = b; a.K = K;
    a.tile_order = A->tile_order;
    a.n_rows = n_rows; a.n_tiles = (n_rows + kTile2 - 1) / kTile2; a.relu = (flags & NDCN_F_RELU) ? 1 : 0;
    EpiArgs ea;
    ea.y0 = y0; ea.n_prev = n_prev; ea.y_next = y_next; ea.rtol = rtol; ea.atol = atol;
    ea.partials = static_cast<double *>(d_ws);
    ea.y1 = (opt && opt->y1) ? opt->y1 : X;
    ea.y_aux = (mode == MODE_COMBINE && opt && opt->y_aux && opt->c_aux) ? opt->y_aux : nullptr;
    for (int m = 0; m <= kMaxPrev; ++m) ea.c2[m] = (ea.y_aux && m <= n_prev) ? opt->c_aux[m] : 0.f;
    static const int dbg = env_int3("NDCN_FUSED_DBG", 0);
    a.dbg = dbg;
    static const int timing = env_int3("NDCN_FUSED_TIMING", 0);
    static unsigned long long *d_cyc = nullptr;
    static int timing_prints = 0;
    if (timing && !d_cyc) (void)hipMalloc(&d_cyc, (size_t)kCus * kWaves * 2 * sizeof(unsigned long long));
    a.dbg_cycles = timing ? d_cyc : nullptr;
    for (int m = 0; m < kMaxPrev; ++m) ea.kprev[m] = (m < n_prev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kMaxPrev; ++m) ea.c[m] = (mode != MODE_PLAIN && mode != MODE_RK4 && m <= n_prev) ? h_c[m] : 0.f;
    if (mode == MODE_RK4) ea.c[0] = h_c[0];                     // the step size
    int per_xcd = kCus / kXcds;
    const int need = (a.n_tiles + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds), block(64 * kWaves);
    const double P = 4.0 * kH2 * (double)A->n_rows;
    double bytes = 8.0 * A_full->nnz + 4.0 * (A_full->n_rows + 1) + 4.0 * kH2 * (double)(A_full->n_rows + A_full->n_cols) + 4.0 * kH2 * kH2;
    if (mode == MODE_COMBINE) bytes += P * (n_prev + 2 + (ea.y_aux ? 1 : 0));   // y0 + earlier stages read, y_next (+ y_aux) written
    if (mode == MODE_ERROR) bytes += P * (n_prev + 2);          // y0 + earlier stages + y1 (row-local re-read)
    if (mode == MODE_RK4) bytes += P * (n_prev + 2);            // y + earlier stages read, next input written
    ProfScope prof(PROF_RHS_FUSED, st, bytes, 2.0 * A_full->nnz * kH2 + 2.0 * (double)A_full->n_rows * kH2 * kH2);
#define NDCN_F2(HALO_, MODE_, NP_) \
    hipLaunchKernelGGL((rhs_fused2_kernel<HALO_, MODE_, NP_>), grid, block, 0, st, A->rowptr, A->colidx, A->val, a, ea)
#define NDCN_F2_DISPATCH(HALO_)                                      \
    do {                                                             \
        if (mode == MODE_PLAIN) NDCN_F2(HALO_, MODE_PLAIN, 0);       \
        else if (mode == MODE_ERROR) { if (n_prev == 1) NDCN_F2(HALO_, MODE_ERROR, 1); else NDCN_F2(HALO_, MODE_ERROR, 5); } \
        else if (mode == MODE_RK4) switch (n_prev) {                 \
            case 0: NDCN_F2(HALO_, MODE_RK4, 0); break;              \
            case 1: NDCN_F2(HALO_, MODE_RK4, 1); break;              \
            case 2: NDCN_F2(HALO_, MODE_RK4, 2); break;              \
            default: NDCN_F2(HALO_, MODE_RK4, 3); break;             \
        }                                                            \
        else switch (n_prev) {                                       \
            case 0: NDCN_F2(HALO_, MODE_COMBINE, 0); break;          \
            case 1: NDCN_F2(HALO_, MODE_COMBINE, 1); break;          \
            case 2: NDCN_F2(HALO_, MODE_COMBINE, 2); break;          \
            case 3: NDCN_F2(HALO_, MODE_COMBINE, 3); break;          \
            case 4: NDCN_F2(HALO_, MODE_COMBINE, 4); break;          \
            default: NDCN_F2(HALO_, MODE_COMBINE, 5); break;         \
        }                                                            \
    } while (0)
    if (Xh) NDCN_F2_DISPATCH(true);
    else NDCN_F2_DISPATCH(false);
#undef NDCN_F2_DISPATCH
#undef NDCN_F2
    NDCN_LAUNCH_CHECK();
    if (mode == MODE_ERROR) {
        int rcf = partials_finish(ea.partials, (int)grid.x * kProd, d_out, st, (opt && opt->accum) ? 1 : 0);
        if (rcf) return rcf;
    }
    if (timing && timing_prints < timing) {                          // debugging aid: s_memtime accounting of block 0 and 100
        (void)hipStreamSynchronize(st);
        unsigned long long h[2 * kWaves];
        for (int bi = 0; bi < 2; ++bi) {
            const int blk = bi == 0 ? 0 : 100;
            (void)hipMemcpy(h, d_cyc + (size_t)blk * 2 * kWaves, sizeof(unsigned long long) * 2 * kWaves, hipMemcpyDeviceToHost);
            double cw = 0, cq = 0, pw = 0, pq = 0;
            for (int w = 0; w < 4; ++w) { cw += h[2 * w] / 4.0; cq += h[2 * w + 1] / 4.0; }
            for (int w = 4; w < 4 + kProd; ++w) { pw += h[2 * w] / (double)kProd; pq += h[2 * w + 1] / (double)kProd; }
            fprintf(stderr, "[fused2 timing] mode %d n_prev %d block %3d: mfma waves work %.0f wait %.0f | gather waves work %.0f wait %.0f\n",
                    mode, n_prev, blk, cw, cq, pw, pq);
        }
        ++timing_prints;
    }
    return NDCN_OK;
}

}  // namespace ndcn
