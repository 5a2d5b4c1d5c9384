// Building blocks of the group-record kernels (spmm_rec.hip: SpMM + RK epilogue; rhs_fused3.hip: the whole ODEFunc):
// counted vector-memory waits, the LDS-DMA row copy and the fold of a row's (slot, value) entries in stored order.
#pragma once
#include "common.h"

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void rec_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt takes an immediate: a wave-uniform run-time count goes through a jump table (n > 24: waiting for fewer
// outstanding operations than necessary is always safe)
__device__ __forceinline__ void rec_wait_vmcnt_rt(int n) {
    asm volatile("; ndcn-wait-begin (one of the s_waitcnt below executes)" ::: "memory");     // marker for tools/audit_async_regs.py
    switch (n) {
#define NDCN_W(k) case k: rec_wait_vmcnt<k>(); break;
        NDCN_W(0) NDCN_W(1) NDCN_W(2) NDCN_W(3) NDCN_W(4) NDCN_W(5) NDCN_W(6) NDCN_W(7) NDCN_W(8) NDCN_W(9) NDCN_W(10) NDCN_W(11)
        NDCN_W(12) NDCN_W(13) NDCN_W(14) NDCN_W(15) NDCN_W(16) NDCN_W(17) NDCN_W(18) NDCN_W(19) NDCN_W(20) NDCN_W(21) NDCN_W(22)
        NDCN_W(23)
#undef NDCN_W
        default: rec_wait_vmcnt<24>(); break;
    }
}

// one 1 KiB row, global -> LDS without a register data path: LDS address = M0 (wave-uniform) + lane * 16.  M0 is
// compiler-reserved: written and restored inside the statement.
__device__ __forceinline__ void dma_row(const float *row_base /*uniform*/, unsigned lds_byte /*uniform*/, int lane_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_byte), "s"(row_base) : "memory");
}

__device__ __forceinline__ f32x4 rec_fma4(float s, f32x4 x, f32x4 a) {
    return (f32x4){fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w)};
}

template <int U, class Src>
__device__ __forceinline__ void rec_chunk(int es, float ev, int base, const Src &src, f32x4 &acc) {
    int sl[U];
    float vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        sl[u] = __builtin_amdgcn_readlane(es, base + u);
        vv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), base + u));
    }
    f32x4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = src(sl[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = rec_fma4(vv[u], x[u], acc);
}
// cnt <= 64 entries held one per lane (es = slot / column, ev = value), folded in stored order; WIDE: 8 rows in
// flight per round (32 VGPRs) instead of 4
template <bool WIDE, class Src>
__device__ __forceinline__ void rec_row(int es, float ev, int cnt, const Src &src, f32x4 &acc) {
    int j = 0;
    if (WIDE) for (; cnt - j >= 8; j += 8) rec_chunk<8>(es, ev, j, src, acc);
    else for (; cnt - j >= 4; j += 4) rec_chunk<4>(es, ev, j, src, acc);
    const int m = cnt - j;
    if (WIDE && (m & 4)) { rec_chunk<4>(es, ev, j, src, acc); j += 4; }
    if (m & 2) { rec_chunk<2>(es, ev, j, src, acc); j += 2; }
    if (m & 1) rec_chunk<1>(es, ev, j, src, acc);
}

}  // namespace ndcn
