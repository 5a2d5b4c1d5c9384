// Gather-side building blocks shared by the fused RHS kernel (rhs_fused2.hip) and the row SpMM (spmm_row.hip):
// buffer-descriptor fetches / stores issued from inline asm with hand-placed waits, and the "whole row in one
// round" neighbour gather with scalar-loaded CSR entries.
#pragma once
#include "common.h"

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 a) {
    return (f32x4){fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w)};
}

// ---- vector-memory fetches with hand-placed waits ---------------------------------------------------------
// A gather wave has three classes of fetches in flight: neighbour rows (L2 hits, needed first), the row-local RK
// panels of the epilogue (HBM, needed later) and its stores.  Vector memory returns in order, so the wait for the
// neighbour rows is exact only as "all but the E youngest fetches" - but how many neighbour fetches a row issues is
// a run-time number, and hipcc's waitcnt insertion then falls back to vmcnt(0), i.e. it also waits for the HBM
// panels (and, with extents read by v_readlane, for the previous row's stores).  The fetches below are therefore
// issued from inline asm (invisible to that pass) and awaited by hand: wait_vmcnt<E>() + tie() of the destination
// registers (an empty asm with the register as in/out operand, ordered after the wait because volatile asms keep
// their order: the first use of the data cannot be scheduled above the wait).
__device__ __forceinline__ u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    return (u32x4){(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ f32x4 fetch128(u32x4 rs, int voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff));
    return v;
}
// streaming (nt) store; issued from asm as well: hipcc guards the data registers of stores it knows about with
// vmcnt waits that - not counting the asm fetches - would drain those instead.  s_nop: the data registers of a
// 16-byte store must not be written in the next wait state.
__device__ __forceinline__ void store128(f32x4 v, u32x4 rs, int voff, unsigned soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// fetch of data that is read exactly once (row-local RK panels): non-temporal, so the stream does not push the
// gathered panel's lines (re-read up to 9 times) out of the XCD's 4 MiB L2
__device__ __forceinline__ f32x4 fetch128_stream(u32x4 rs, int voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }
__device__ __forceinline__ void tie(f32x4 &v) { asm volatile("" : "+v"(v)); }

// Issue U neighbour-row fetches of one output row: entries j .. j+U-1 of the CSR arrays.
// The gather waves share their SIMD with an MFMA wave, and on gfx950 the fp32 MFMA keeps the SIMD's VALU busy
// (measured: the same gather code takes 2x the cycles while the MFMA waves run, independent of memory traffic
// and of wave priorities).  So the per-neighbour work is kept OFF the VALU: column index and value arrive by
// scalar loads (the CSR arrays are __restrict__ kernel arguments, j is wave-uniform), the row address is a
// buffer-load SGPR offset (col << 10) on top of a fixed per-lane offset - the only VALU work left per
// neighbour is the two packed FMAs.
template <int U, bool HALO, int O = 0>
__device__ __forceinline__ void g_issue(const int *__restrict__ colidx, const float *__restrict__ val, int j, u32x4 rsX,
                                        u32x4 rsH, int n_own, int lane_off, f32x4 (&x)[16], float (&vv)[16]) {
    // all scalar loads first: a volatile asm is a scheduling barrier, and U separate s_load_dword + waits (instead of
    // one s_load_dwordx8) would serialise a scalar-cache round trip per neighbour
    int cc[U];
#ifdef NDCN_DBG_NOIDX      // timing experiment: lattice-like indices formed arithmetically, no index loads (results are wrong)
#pragma unroll
    for (int q = 0; q < U; ++q) {
        const int e = O + q, rr = j / 9;
        int c = rr + (e % 3 - 1) + (e / 3 % 3 - 1) * 1000;
        cc[q] = __builtin_amdgcn_readfirstlane(c < 0 ? 0 : (c > 999999 ? 999999 : c));
        vv[O + q] = 0.1f;
    }
#else
#pragma unroll
    for (int q = 0; q < U; ++q) { cc[q] = colidx[j + q]; vv[O + q] = val[j + q]; }
#endif
#pragma unroll
    for (int q = 0; q < U; ++q) {
        u32x4 rs = rsX;
        int c = cc[q];
        if (HALO && c >= n_own) { rs = rsH; c -= n_own; }
        x[O + q] = fetch128(rs, lane_off, (unsigned)c << 10);
    }
}
template <int U, int O = 0>
__device__ __forceinline__ void g_accum(f32x4 (&x)[16], const float (&vv)[16], f32x4 &acc) {
#pragma unroll
    for (int q = 0; q < U; ++q) { tie(x[O + q]); acc = fma4(vv[O + q], x[O + q], acc); }
}

// The last m < 16 entries of a row in ONE round: pieces of 8 / 4 / 2 / 1 in slots 0-7 / 8-11 / 12-13 / 14, all
// issued before any is awaited (a 9-entry grid row is one fetch latency, not two); g_row_accum folds them in
// ascending entry order, so the sum rounds exactly like a sequential loop over the row.
template <bool HALO>
__device__ __forceinline__ void g_row_issue(const int *__restrict__ colidx, const float *__restrict__ val, int j, int m,
                                            u32x4 rsX, u32x4 rsH, int n_own, int lane_off, f32x4 (&x)[16], float (&vv)[16]) {
    if (m & 8) { g_issue<8, HALO, 0>(colidx, val, j, rsX, rsH, n_own, lane_off, x, vv); j += 8; }
    if (m & 4) { g_issue<4, HALO, 8>(colidx, val, j, rsX, rsH, n_own, lane_off, x, vv); j += 4; }
    if (m & 2) { g_issue<2, HALO, 12>(colidx, val, j, rsX, rsH, n_own, lane_off, x, vv); j += 2; }
    if (m & 1) { g_issue<1, HALO, 14>(colidx, val, j, rsX, rsH, n_own, lane_off, x, vv); }
}
// caller has waited for the fetches
__device__ __forceinline__ void g_row_accum(int m, f32x4 (&x)[16], const float (&vv)[16], f32x4 &acc) {
    if (m & 8) g_accum<8, 0>(x, vv, acc);
    if (m & 4) g_accum<4, 8>(x, vv, acc);
    if (m & 2) g_accum<2, 12>(x, vv, acc);
    if (m & 1) g_accum<1, 14>(x, vv, acc);
}

}  // namespace ndcn
