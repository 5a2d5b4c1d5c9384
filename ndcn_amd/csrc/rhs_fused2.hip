// Fused ODEFunc right-hand side with Runge-Kutta epilogue (H = 256):
//
//     K = relu((A X) W^T + b)          [neural_dynamics.py:27-36]
//   + optionally, in the same pass, the Runge-Kutta algebra that consumes K:
//       COMBINE :  y_next = y0 + sum_m c_m k_m   (k_new last)      [rk_common.py:51 / misc.py:22-25]
//       ERROR   :  sum ((sum_m c_m k_m) / (atol + rtol max(|y0|, |y1|)))^2 and the non-finite count of y1
//                                                                  [rk_common.py:60, misc.py:146-157, dopri5.py:101]
//
// One persistent workgroup per CU: 4 consumer waves (one per SIMD, fp32 MFMA) + 8 producer waves; 64-row tiles;
// LDS = S[2][64][260] fp32 (133 120 B).
//
//   phase A(t): consumers  MFMA on S[t&1]                            -> K_t in registers
//               producers  per pair of rows: request the row-local RK panels of K_{t-1}'s rows (K_{t-1} sits in
//                          S[(t-1)&1]), gather the same two rows of S_{t+1} = (A X)[tile t+1] into registers,
//                          then finish the epilogue (K rows + RK algebra streamed to HBM as whole 1 KiB rows)
//                          and drop the gathered rows into S[(t-1)&1]
//   barrier
//   phase B(t): consumers  K_t (+bias, relu) -> S[t&1], row-major
//   barrier
// A producer wave owns the same rows of every tile (p, p+8, ..., p+56): it first reads K out of them, then
// overwrites them with the gathered S, so phase A needs no producer-to-producer synchronisation.
//
// Measured lessons built in (1M-node grid, MI355X; profiles/r01*):
//   * the first fused kernel ran its MFMA loop at ~50 % because hipcc scheduled the L2 weight fetch of step q+1
//     late inside step q and then waited vmcnt(0): the weight operands now sit in a 4-deep register ring (a
//     fetch is issued right after its slot is consumed, three steps = 3072 MFMA cycles before it is needed);
//   * accumulators leave through LDS as full rows instead of 64 scattered 4-byte stores per lane;
//   * a union-staged gather (fetch each distinct neighbour row of 8 rows once) needed a producer barrier per
//     group and was latency-bound (2.1 ms vs 1.8 ms); the gather here is direct, two rows in flight per wave.
#include <stdlib.h>

#include "kernels.h"

#pragma clang fp contract(off)   // the RK algebra below must round like the reference's separate mul / add ops

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH2 = 256;
constexpr int kTile2 = 64;
constexpr int kLd2 = kH2 + 4;              // +4 floats: row m starts on 16-byte slot (4 m) mod 64 -> conflict-free b128
constexpr int kTileFloats2 = kTile2 * kLd2;
#ifndef NDCN_KPROD
#define NDCN_KPROD 8
#endif
constexpr int kProd = NDCN_KPROD;
constexpr int kRowsPerProd = kTile2 / kProd;
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
};

struct Fused2Args {
    const float *X, *Xh;
    int x_bytes, xh_bytes;                  // sizes of the gathered panels (buffer descriptors: < 2^31)
    int n_own;
    const float *Wp, *bias;
    float *K;                               // relu(...) output panel
    int n_rows, n_tiles, relu;
    unsigned long long *dbg_cycles;         // NDCN_FUSED_TIMING: per (block, wave) {work cycles, barrier-wait cycles}
    int dbg;                                // timing experiments only (NDCN_FUSED_DBG): 1 skip MFMA, 2 skip gather, 4 skip epilogue
};
typedef const __attribute__((address_space(4))) EpiArgs *EpiPtr;
// kernel parameters: rowptr, colidx, val (3 x 8 bytes), Fused2Args, EpiArgs - both structs are 8-aligned
constexpr int kEpiKernargOffset = 24 + (int)((sizeof(Fused2Args) + 7) / 8 * 8);

enum { MODE_PLAIN = 0, MODE_COMBINE = 1, MODE_ERROR = 2 };

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 a) {
    return (f32x4){fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w)};
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Issue U neighbour-row fetches of one output row: entries j .. j+U-1 of the CSR arrays.
// The gather waves share their SIMD with an MFMA wave, and on gfx950 the fp32 MFMA keeps the SIMD's VALU busy
// (measured: the same gather code takes 2x the cycles while the MFMA waves run, independent of memory traffic
// and of wave priorities).  So the per-neighbour work is kept OFF the VALU: column index and value arrive by
// scalar loads (the CSR arrays are __restrict__ kernel arguments, j is wave-uniform), the row address is a
// buffer-load SGPR offset (col << 10) on top of a fixed per-lane offset - the only VALU work left per
// neighbour is the two packed FMAs.
template <int U, bool HALO, int O = 0>
__device__ __forceinline__ void g_issue(const int *__restrict__ colidx, const float *__restrict__ val, int j,
                                        __amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsH, int n_own, int lane_off,
                                        f32x4 (&x)[8], float (&vv)[8]) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
        int cc = colidx[j + q];
        vv[O + q] = val[j + q];
        __amdgpu_buffer_rsrc_t rs = rsX;
        if (HALO && cc >= n_own) { rs = rsH; cc -= n_own; }
        x[O + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, cc << 10, 0));
    }
}
template <int U, int O = 0>
__device__ __forceinline__ void g_accum(const f32x4 (&x)[8], const float (&vv)[8], f32x4 &acc) {
#pragma unroll
    for (int q = 0; q < U; ++q) acc = fma4(vv[O + q], x[O + q], acc);
}

// tails of two rows (fewer than 8 entries left each): pieces of 4 / 2 / 1, the same-size pieces of both rows in
// flight together (one fetch latency per piece size instead of one per row and size); entries are folded in
// ascending order within each row, so the sums round exactly as a row-at-a-time loop would
template <int U, bool HALO>
__device__ __forceinline__ void g_piece2(const int *__restrict__ colidx, const float *__restrict__ val, int &jA, int nA,
                                         int &jB, int nB, __amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsH, int n_own,
                                         int lane_off, f32x4 &accA, f32x4 &accB) {
    f32x4 xA[8], xB[8];
    float wA[8], wB[8];
    const bool hA = nA & U, hB = nB & U;
    if (hA) g_issue<U, HALO>(colidx, val, jA, rsX, rsH, n_own, lane_off, xA, wA);
    if (hB) g_issue<U, HALO>(colidx, val, jB, rsX, rsH, n_own, lane_off, xB, wB);
    if (hA) { g_accum<U>(xA, wA, accA); jA += U; }
    if (hB) { g_accum<U>(xB, wB, accB); jB += U; }
}

// whole batches of 8 of the entries [j, j1) of a row; returns the first entry of the tail
template <bool HALO>
__device__ __forceinline__ int g_batches(const int *__restrict__ colidx, const float *__restrict__ val, int j, int j1,
                                         __amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsH, int n_own, int lane_off,
                                         f32x4 &acc) {
    f32x4 x[8];
    float vv[8];
    for (; j + 8 <= j1; j += 8) { g_issue<8, HALO>(colidx, val, j, rsX, rsH, n_own, lane_off, x, vv); g_accum<8>(x, vv, acc); }
    return j;
}

template <bool HALO, int MODE>
__global__ __launch_bounds__(256 + 64 * kProd) void rhs_fused2_kernel(const int *__restrict__ rowptr,
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

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int p = wave - 4;                                   // producer index 0..7
    const f32x4 *X = reinterpret_cast<const f32x4 *>(a.X);
    const f32x4 *Wp = reinterpret_cast<const f32x4 *>(a.Wp);
    // buffer descriptors of the gathered panels (wave-uniform; byte sizes < 2^31 checked by the launcher)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.X), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HALO ? a.Xh : a.X), 0,
                                                                          HALO ? a.xh_bytes : a.x_bytes, 0x00020000);
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

    // ---- producer: row extents of the wave's 8 rows of a tile by one vector load; they stay in that VGPR (lane
    // 2k / 2k+1 = begin / end of row k) and are read out per row pair - 16 resident SGPRs less
    int ix_rp = 0;
    auto prefetch_index = [&](int t) {
        ix_rp = 0;
        if (lane < 2 * kRowsPerProd) {
            int r = t * kTile2 + p + kProd * (lane >> 1) + (lane & 1);
            ix_rp = rowptr[min(r, a.n_rows)];       // rows past the end: begin == end == rowptr[n_rows]
        }
    };

    // ---- producer: stream K rows of tile `t` out of LDS tile src (+ RK algebra) ----------------------
    // The row-local panels of row k+1 are requested before row k is finished (one row of fetches always in flight).
    struct EpiRow { f32x4 km[kMaxPrev]; f32x4 y0v, y1v; };
    // Row-local panels go through buffer descriptors built on the scalar unit: address = descriptor base + SGPR row
    // offset (r << 10) + the fixed per-lane offset - no VALU address arithmetic per panel (the producers' VALU time
    // is what the MFMA waves squeeze).  Panels are < 2 GiB (launcher check).
    const int panel_bytes = a.n_rows << 10;
    auto ldp = [&](const float *base, int row_off) {
        asm volatile("" : "+s"(base));      // keep the 4-SGPR descriptor transient: hoisted descriptors for 8 panels spill
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, panel_bytes, 0x00020000);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, row_off, 0));
    };
    auto stp = [&](float *base, int row_off, f32x4 v) {
        asm volatile("" : "+s"(base));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, panel_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane_off, row_off, 2 /* nt */);
    };
    auto epi_load = [&](EpiPtr ea, int r, EpiRow &e) {
        const int off = (a.dbg & 2048) ? ((r & 63) << 10) : (r << 10);      // timing experiment: cache-resident panels
        if (a.dbg & 256) {                                     // timing experiment: no row-local fetches
#pragma unroll
            for (int m = 0; m < kMaxPrev; ++m) e.km[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            e.y0v = (f32x4){0.f, 0.f, 0.f, 0.f};
            return;
        }
        const int n_prev = ea->n_prev;
#pragma unroll
        for (int m = 0; m < kMaxPrev; ++m)
            if (m < n_prev) e.km[m] = ldp(ea->kprev[m], off);
        e.y0v = ldp(ea->y0, off);
    };
    // sum of the earlier stages (left to right, as misc.py:22-25 accumulates), into km[0]: frees the fetch registers
    auto epi_reduce = [&](EpiPtr ea, int r, EpiRow &e) {
        const int n_prev = ea->n_prev;
        if (n_prev > 0) {
            f32x4 s = e.km[0] * ea->c[0];
#pragma unroll
            for (int m = 1; m < kMaxPrev; ++m)
                if (m < n_prev) s = s + e.km[m] * ea->c[m];
            e.km[0] = s;
        }
        // ERROR: the input of this evaluation is y1 (own rows); requested only now that the stage registers are free
        if (MODE == MODE_ERROR) e.y1v = (a.dbg & 256) ? e.y0v : ldp(a.X, r << 10);
    };
    auto epi_finish = [&](EpiPtr ea, int r, const float *src_row, const EpiRow &e) {
        const f32x4 kn = *reinterpret_cast<const f32x4 *>(src_row + 4 * lane);
        const int off = r << 10;
        stp(a.K, off, kn);
        if (MODE == MODE_PLAIN) return;
        if (a.dbg & 1024) { if (MODE == MODE_COMBINE) stp(ea->y_next, off, kn); return; }   // timing experiment: no algebra
        const int n_prev = ea->n_prev;
        f32x4 s = kn * ea->c[n_prev];                          // the new stage is the last term of the sum
        if (n_prev > 0) s = e.km[0] + s;
        if (MODE == MODE_COMBINE) {
            if (!(a.dbg & 512)) stp(ea->y_next, off, e.y0v + s);
            else if (s[0] == 1.2345e-30f) stp(ea->y_next, off, e.y0v + s);    // timing experiment: store never taken
        } else {
            const float rtol = ea->rtol, atol = ea->atol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tol = atol + rtol * fmaxf(fabsf(e.y0v[q]), fabsf(e.y1v[q]));
                const float z = s[q] / tol;
                err_sum += (double)(z * z);
                err_bad += (double)(int)(!(fabsf(e.y1v[q]) <= 3.402823466e38f));
            }
        }
    };
    // ---- producer phase A: per pair of the wave's rows -----------------------------------------------------
    //   1. request the row-local panels (y0, earlier stages) of rows (A, B) of tile t_epi, whose K sits in `buf`
    //   2. gather rows (A, B) of the NEXT tile (extents prefetched) into registers - ~10k cycles, which is what
    //      hides the latency of (1): with the epilogue run as its own loop that latency was exposed once per row
    //      (measured 2.3k cycles per row, +1.2..1.7 M cycles per launch); the panels of (1) are folded into one
    //      partial sum per row as soon as the first gather batch has landed (they are older, so they have too)
    //   3. finish the epilogue of rows (A, B): K out of `buf`, RK algebra, stores
    //   4. drop the gathered rows into the same two rows of `buf`
    // A producer wave owns the same rows of every tile, so no producer-to-producer synchronisation is needed.
    auto producer_phase = [&](float *buf, int t_epi, bool do_epi, bool do_gather) {
        // one (laundered) read of the epilogue arguments per tile: re-reading them at every use cost 11 scalar
        // loads + waits per row (SQ_INSTS_SMEM 15.7 M vs 4.0 M per launch)
        EpiPtr ea = nullptr;
        if (MODE != MODE_PLAIN) ea = epi_args();
        const int r0 = t_epi * kTile2 + p;
#pragma unroll
        for (int k = 0; k < kRowsPerProd; k += 2) {
            const int lrA = p + kProd * k, lrB = lrA + kProd;
            const int rA = r0 + kProd * k, rB = rA + kProd;
            const bool epA = do_epi && rA < a.n_rows, epB = do_epi && rB < a.n_rows;
            EpiRow eA, eB;
            // ERROR carries y1 and the fp64 partial sums on top: row B's panels are requested only after row A's
            // have been folded (they then overlap the tail of the gather), which keeps the variant free of spills
            constexpr bool kDeferB = MODE == MODE_ERROR;
            if (MODE != MODE_PLAIN) {
                if (epA) epi_load(ea, rA, eA);
                if (epB && !kDeferB) epi_load(ea, rB, eB);
            }
            f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accB = accA;
            int jA = 0, jB = 0, jA1 = 0, jB1 = 0;
            if (do_gather) {
                jA = __builtin_amdgcn_readlane(ix_rp, 2 * k);
                jA1 = __builtin_amdgcn_readlane(ix_rp, 2 * k + 1);
                jB = __builtin_amdgcn_readlane(ix_rp, 2 * k + 2);
                jB1 = __builtin_amdgcn_readlane(ix_rp, 2 * k + 3);
                // first batches of both rows in flight together (16 x 1 KiB per wave)
                f32x4 xA[8], xB[8];
                float wA[8], wB[8];
                const bool fa = jA + 8 <= jA1, fb = jB + 8 <= jB1;
                if (fa) g_issue<8, HALO>(colidx, val, jA, rsX, rsH, a.n_own, lane_off, xA, wA);
                if (kDeferB) {                 // ERROR: one row's batch at a time (register budget, see above)
                    if (fa) { g_accum<8>(xA, wA, accA); jA += 8; }
                    if (fb) g_issue<8, HALO>(colidx, val, jB, rsX, rsH, a.n_own, lane_off, xA, wA);
                    if (fb) { g_accum<8>(xA, wA, accB); jB += 8; }
                } else {
                    if (fb) g_issue<8, HALO>(colidx, val, jB, rsX, rsH, a.n_own, lane_off, xB, wB);
                    if (fa) { g_accum<8>(xA, wA, accA); jA += 8; }
                    if (fb) { g_accum<8>(xB, wB, accB); jB += 8; }
                }
            }
            if (MODE != MODE_PLAIN) {
                if (epA) epi_reduce(ea, rA, eA);
                if (epB && !kDeferB) epi_reduce(ea, rB, eB);
                if (epB && kDeferB) epi_load(ea, rB, eB);
            }
            if (do_gather) {
                jA = g_batches<HALO>(colidx, val, jA, jA1, rsX, rsH, a.n_own, lane_off, accA);     // rows > 16 entries
                jB = g_batches<HALO>(colidx, val, jB, jB1, rsX, rsH, a.n_own, lane_off, accB);
                const int nA = jA1 - jA, nB = jB1 - jB;
                g_piece2<4, HALO>(colidx, val, jA, nA, jB, nB, rsX, rsH, a.n_own, lane_off, accA, accB);
                g_piece2<2, HALO>(colidx, val, jA, nA, jB, nB, rsX, rsH, a.n_own, lane_off, accA, accB);
                g_piece2<1, HALO>(colidx, val, jA, nA, jB, nB, rsX, rsH, a.n_own, lane_off, accA, accB);
            }
            if (MODE != MODE_PLAIN && kDeferB && epB) epi_reduce(ea, rB, eB);
            if (epA) epi_finish(ea, rA, buf + lrA * kLd2, eA);
            if (epB) epi_finish(ea, rB, buf + lrB * kLd2, eB);
            if (do_gather) {
                *reinterpret_cast<f32x4 *>(buf + lrA * kLd2 + 4 * lane) = accA;
                *reinterpret_cast<f32x4 *>(buf + lrB * kLd2 + 4 * lane) = accB;
            }
        }
    };

    // ---- consumer -----------------------------------------------------------------------------------
    // wave w owns output columns [64 w, 64 w + 64) (n-tiles 2w, 2w+1) for both 32-row m-tiles
    f32x16 acc00, acc01, acc10, acc11;
    // Ring of weight operands, kRing deep: slot u holds k-quad q with q % kRing == u.  The weights are the same
    // for every tile, so the ring simply wraps around - it is already full when the next tile starts.
    constexpr int kRing = 4;
    const f32x4 *b0p = Wp + (size_t)(2 * (wave & 3)) * 32 * 64 + lane;
    const f32x4 *b1p = b0p + 32 * 64;
    f32x4 r0[kRing], r1[kRing];
    auto ring_fill = [&]() {
#pragma unroll
        for (int u = 0; u < kRing; ++u) { r0[u] = b0p[u * 64]; r1[u] = b1p[u * 64]; }
    };
    auto mfma_tile = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }
        const float *a0p = src + (lane & 31) * kLd2 + 128 * (lane >> 5);
        const float *a1p = a0p + 32 * kLd2;
        // A operands ping-pong between an even-quad and an odd-quad register set, and a weight slot is refilled
        // AFTER the MFMAs that read it: no value ever has to be copied to free its register (the first version
        // rotated through temporaries - 116 v_mov per 64 MFMAs, issued on the VALU port the gather waves need)
        f32x4 aE0 = *reinterpret_cast<const f32x4 *>(a0p), aE1 = *reinterpret_cast<const f32x4 *>(a1p);
        f32x4 aO0 = aE0, aO1 = aE1;
#pragma unroll 1
        for (int q0 = 0; q0 < 32; q0 += kRing) {
#pragma unroll
            for (int u = 0; u < kRing; ++u) {
                const int q = q0 + u;
                if ((u & 1) == 0) {
                    aO0 = *reinterpret_cast<const f32x4 *>(a0p + 4 * (q + 1));
                    aO1 = *reinterpret_cast<const f32x4 *>(a1p + 4 * (q + 1));
                } else if (q + 1 < 32) {
                    aE0 = *reinterpret_cast<const f32x4 *>(a0p + 4 * (q + 1));
                    aE1 = *reinterpret_cast<const f32x4 *>(a1p + 4 * (q + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = (u & 1) ? aO0[e] : aE0[e], x1 = (u & 1) ? aO1[e] : aE1[e];
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, r0[u][e], acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, r1[u][e], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, r0[u][e], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, r1[u][e], acc11, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the refill wraps into the next tile's first quads (dbg bit 8 = do not wrap: A/B switch)
                if ((!(a.dbg & 8) || q + kRing < 32) && !(a.dbg & 64)) {       // dbg 64: no weight refills (timing experiment)
                    const int qn = (q + kRing) & 31;
                    r0[u] = b0p[qn * 64];
                    r1[u] = b1p[qn * 64];
                }
            }
        }
    };
    auto dump_tile = [&](float *dst) {
        // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = 64 * wave + 32 * n + (lane & 31);
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float o = (mt == 0 ? (n == 0 ? acc00[r] : acc01[r]) : (n == 0 ? acc10[r] : acc11[r])) + bv;
                    if (a.relu) o = fmaxf(o, 0.f);
                    dst[m * kLd2 + col] = o;
                }
        }
    };

    // Role-specialised loops (both execute the same sequence of workgroup barriers), so the accumulators of the
    // MFMA waves and the fetch registers of the gather waves never share a live range.
    unsigned long long cyc_work = 0, cyc_wait = 0;
    if (producer) {
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(3);         // timing experiment: gather waves win issue arbitration
        prefetch_index(t_first);
        producer_phase(s_tile, 0, false, true);
        __syncthreads();                                       // S[0] ready
        for (int it = 0; it < my_tiles; ++it) {
            const int t = t_first + it * wgs_per_xcd;
            float *oth = s_tile + ((it & 1) ^ 1) * kTileFloats2;
            const unsigned long long c0 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            // phase A: K of the previous tile sits in `oth`; stream it out, then refill `oth` with the next S
            const bool do_gather = it + 1 < my_tiles && !(a.dbg & 2);
            if (do_gather) prefetch_index(t + wgs_per_xcd);
            if (!((a.dbg & 4096) && p >= 4))                   // timing experiment: only half of the gather waves work
                producer_phase(oth, t - wgs_per_xcd, it > 0 && !(a.dbg & 4), do_gather);
            const unsigned long long c1 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            // phase B: consumers drop K_t into the tile they consumed
            __syncthreads();
            if (a.dbg_cycles) { cyc_work += c1 - c0; cyc_wait += __builtin_readcyclecounter() - c1; }
        }
        if (a.dbg_cycles && lane == 0) {
            a.dbg_cycles[2 * (blockIdx.x * 12 + wave)] = cyc_work;
            a.dbg_cycles[2 * (blockIdx.x * 12 + wave) + 1] = cyc_wait;
        }
        const int t_last = t_first + (my_tiles - 1) * wgs_per_xcd;
        producer_phase(s_tile + ((my_tiles - 1) & 1) * kTileFloats2, t_last, true, false);
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
        if (a.dbg & 128) __builtin_amdgcn_s_setprio(3);        // timing experiment: MFMA waves win issue arbitration
        ring_fill();                                           // weight fetches fly while the first tile is gathered
        __syncthreads();                                       // S[0] ready
        for (int it = 0; it < my_tiles; ++it) {
            float *cur = s_tile + (it & 1) * kTileFloats2;
            const unsigned long long c0 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            if ((a.dbg & 8) && it > 0) ring_fill();
            if (!(a.dbg & 1)) mfma_tile(cur);
            const unsigned long long c1 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();                                   // every consumer is done reading `cur`
            const unsigned long long c2 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            dump_tile(cur);
            const unsigned long long c3 = a.dbg_cycles ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            if (a.dbg_cycles) { cyc_work += (c1 - c0) + (c3 - c2); cyc_wait += (c2 - c1) + (__builtin_readcyclecounter() - c3); }
        }
        if (a.dbg_cycles && lane == 0) {
            a.dbg_cycles[2 * (blockIdx.x * 12 + wave)] = cyc_work;
            a.dbg_cycles[2 * (blockIdx.x * 12 + wave) + 1] = cyc_wait;
        }
    }
}

// fixed-order sum of the per-producer partials (deterministic accept / reject)
__global__ __launch_bounds__(256) void fused2_finish_kernel(const double *__restrict__ partial, int n, double *__restrict__ out) {
    __shared__ double sa[256], sb[256];
    double s = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s += partial[2 * i]; bad += partial[2 * i + 1]; }
    sa[threadIdx.x] = s; sb[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { sa[threadIdx.x] += sa[threadIdx.x + w]; sb[threadIdx.x] += sb[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sa[0]; out[1] = sb[0]; }
}

static int env_int3(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

int rhs_fused2_supported(const ndcn_csr *A, int H, uint32_t flags) {
    static const int enabled = env_int3("NDCN_RHS_FUSED2", 1);
    if (!enabled || H != kH2 || !A) return 0;
    if (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) return 0;
    return A->n_cols * (int64_t)kH2 * 4 < (1ll << 31) ? 1 : 0;       // gathered panel must fit a buffer descriptor
}

int64_t rhs_fused2_partials_bytes() { return (int64_t)kCus * kProd * 2 * sizeof(double); }

// mode: 0 plain; 1 combine (y_next = y0 + sum c_m k_m, new K last); 2 error (d_out[0..1], d_ws scratch)
int rhs_fused2_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b,
                   float *K, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c,
                   int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    if (n_prev < 0 || n_prev > kMaxPrev) { set_error("rhs_fused2: at most %d previous stages", kMaxPrev); return NDCN_EINVAL; }
    Fused2Args a;
    const int64_t xb = (Xh ? n_own : A->n_cols) * (int64_t)kH2 * 4, xhb = Xh ? (A->n_cols - n_own) * (int64_t)kH2 * 4 : 0;
    if (xb >= (1ll << 31) || xhb >= (1ll << 31)) {
        set_error("rhs_fused2: panel of %lld bytes exceeds the 2 GiB buffer-descriptor range", (long long)(xb > xhb ? xb : xhb));
        return NDCN_EINVAL;
    }
    a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.x_bytes = (int)xb; a.xh_bytes = (int)xhb; a.Wp = Wp; a.bias = b; a.K = K;
    a.n_rows = n_rows; a.n_tiles = (n_rows + kTile2 - 1) / kTile2; a.relu = (flags & NDCN_F_RELU) ? 1 : 0;
    EpiArgs ea;
    ea.y0 = y0; ea.n_prev = n_prev; ea.y_next = y_next; ea.rtol = rtol; ea.atol = atol;
    ea.partials = static_cast<double *>(d_ws);
    static const int dbg = env_int3("NDCN_FUSED_DBG", 0);
    a.dbg = dbg;
    static const int timing = env_int3("NDCN_FUSED_TIMING", 0);
    static unsigned long long *d_cyc = nullptr;
    static int timing_prints = 0;
    if (timing && !d_cyc) (void)hipMalloc(&d_cyc, (size_t)kCus * 12 * 2 * sizeof(unsigned long long));
    a.dbg_cycles = timing ? d_cyc : nullptr;
    for (int m = 0; m < kMaxPrev; ++m) ea.kprev[m] = (m < n_prev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kMaxPrev; ++m) ea.c[m] = (mode != MODE_PLAIN && m <= n_prev) ? h_c[m] : 0.f;
    int per_xcd = kCus / kXcds;
    const int need = (a.n_tiles + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds), block(256 + 64 * kProd);
    const double P = 4.0 * kH2 * (double)A->n_rows;
    double bytes = 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * kH2 * (double)(A->n_rows + A->n_cols) + 4.0 * kH2 * kH2;
    if (mode == MODE_COMBINE) bytes += P * (n_prev + 2);        // y0 + earlier stages read, y_next written
    if (mode == MODE_ERROR) bytes += P * (n_prev + 2);          // y0 + earlier stages + y1 (row-local re-read)
    ProfScope prof(PROF_RHS_FUSED, st, bytes, 2.0 * A->nnz * kH2 + 2.0 * (double)A->n_rows * kH2 * kH2);
#define NDCN_F2(HALO_, MODE_) \
    hipLaunchKernelGGL((rhs_fused2_kernel<HALO_, MODE_>), grid, block, 0, st, A->rowptr, A->colidx, A->val, a, ea)
    if (Xh) {
        if (mode == MODE_PLAIN) NDCN_F2(true, MODE_PLAIN);
        else if (mode == MODE_COMBINE) NDCN_F2(true, MODE_COMBINE);
        else NDCN_F2(true, MODE_ERROR);
    } else {
        if (mode == MODE_PLAIN) NDCN_F2(false, MODE_PLAIN);
        else if (mode == MODE_COMBINE) NDCN_F2(false, MODE_COMBINE);
        else NDCN_F2(false, MODE_ERROR);
    }
#undef NDCN_F2
    if (mode == MODE_ERROR)
        hipLaunchKernelGGL(fused2_finish_kernel, dim3(1), dim3(256), 0, st, ea.partials, (int)grid.x * kProd, d_out);
    NDCN_LAUNCH_CHECK();
    if (timing && timing_prints < timing) {                          // debugging aid: s_memtime accounting of block 0 and 100
        (void)hipStreamSynchronize(st);
        unsigned long long h[2 * 12 * 2];
        for (int bi = 0; bi < 2; ++bi) {
            const int blk = bi == 0 ? 0 : 100;
            (void)hipMemcpy(h, d_cyc + (size_t)blk * 24, sizeof(unsigned long long) * 24, hipMemcpyDeviceToHost);
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
