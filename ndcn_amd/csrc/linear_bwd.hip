// Backward of the Linear on the path:  Y = act(S W^T + b)   (neural_dynamics.py:33,36 under autograd - the reference
// trains by backpropagating through every solver step, heat_dynamics.py:333, dgnn.py:204 - and the encoder / decoder
// Linears :143-148).  With gZ = g (.) [Y > 0] when the ReLU output Y is given (mask fused into the operand loads, no
// separate pass), gZ = g otherwise:
//
//   gS [n, Hi]  = gZ W                fp32 MFMA GEMM, W read transposed while it is staged (no W^T copy); Hi = Ho = 256: the
//                                     forward's two-piece fp16 product with the planes of W^T (linear_gs_256_split_kernel)
//   gW [Ho, Hi] = gZ^T S              "split-K": the reduction runs over the n rows; each workgroup owns a contiguous chunk of
//                                     rows and writes one partial Ho x Hi block, summed afterwards in a FIXED order
//                                     (deterministic gradients, no atomics).  fp32 MFMA with both operands read straight from
//                                     HBM in operand order; Hi = Ho = 256: three bf16 pieces per operand, staged once per
//                                     16 rows in LDS (linear_wgrad_256_split_kernel)
//   gb [Ho]     = column sums of gZ   rides in the gW kernel (the A operand passes through the lanes anyway)
//
// MFMA operand map (32x32x2 f32): A lane l = A[m = l & 31][k = l >> 5]; B lane l = B[k = l >> 5][n = l & 31];
// D reg r = D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31].
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "split16.h"

namespace ndcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGwWaves = 8;          // o-strips of 32 per workgroup (Ho <= 256 per pass)
constexpr int kGwMaxNI = 8;          // i-tiles of 32 per wave (Hi <= 256 per pass)
constexpr int kGwMaxChunks = 256;

__device__ __forceinline__ float masked(const float *__restrict__ g, const float *__restrict__ Y, int64_t idx) {
    const float v = g[idx];
    return (Y && !(Y[idx] > 0.f)) ? 0.f : v;
}

// ---------------------------------------------------------------------------------------------------- gW, gb
// grid.x = row chunks, grid.y = o passes of 256, grid.z = i passes of 256
template <int NI>
__global__ __launch_bounds__(64 * kGwWaves) void linear_wgrad_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                                       const float *__restrict__ S, float *__restrict__ part_w,
                                                                       float *__restrict__ part_b, int64_t n, int Hi, int Ho,
                                                                       int64_t rows_per_chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o0 = blockIdx.y * 256 + wave * 32, i0 = blockIdx.z * 256;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r_hi = r_lo + rows_per_chunk < n ? r_lo + rows_per_chunk : n;
    if (o0 >= Ho) return;
    const int o = o0 + (lane & 31);
    const int kk = lane >> 5;
    const bool o_ok = o < Ho;
    f32x16 acc[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;
    bool i_ok[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) i_ok[t] = i0 + 32 * t + (lane & 31) < Hi;

    for (int64_t r = r_lo; r < r_hi; r += 8) {                   // 4 row pairs per round: their loads fly together
        float a[4], b[4][NI];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = r + 2 * u + kk;
            const bool ok = row < r_hi;
            a[u] = (ok && o_ok) ? masked(g, Y, row * Ho + o) : 0.f;
#pragma unroll
            for (int t = 0; t < NI; ++t) b[u][t] = (ok && i_ok[t]) ? S[row * Hi + i0 + 32 * t + (lane & 31)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum += a[u];
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    float *pw = part_w + (size_t)blockIdx.x * Ho * Hi;
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int i = i0 + 32 * t + (lane & 31);
        if (i >= Hi) continue;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int oo = o0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
            if (oo < Ho) pw[(size_t)oo * Hi + i] = acc[t][rr];
        }
    }
    if (part_b && blockIdx.z == 0) {
        bsum += __shfl_xor(bsum, 32, 64);                        // the two row parities of this column
        if (lane < 32 && o_ok) part_b[(size_t)blockIdx.x * Ho + o] = bsum;
    }
}

// ---- gW, gb for Hi = Ho = 256 on the bf16 matrix cores with fp32-grade results.  The reduction runs over the ROWS, where the
// per-row power-of-two scale of the fp16 pieces (split16.h) cannot follow, and column scales would cost a pass of their own
// (measured: 0.117 ms - the whole budget).  bf16 has fp32's exponent range: every operand is split error-free into THREE bf16
// pieces (8 + 8 + 8 significand bits) and the six partial products with i + j <= 4 are accumulated in fp32, small terms first -
// the product of rounds 1-2 (profiles/r02c_gemm_split_lab.txt: 2.0e-7 of sum |a b| against fp64, fp32 MFMA chain 2.3e-7).
// One workgroup (8 waves) per row chunk, the whole 256 x 256 block: per k-step of 16 rows thread (column c, row half h)
// fetches gZ[8 rows][c] and S[8 rows][c] (a wave's fetch = 64 consecutive columns of one row; a pair of k-steps ahead), splits
// them and writes the 16-byte operand chunks - A[m = c][k = 8 h ..] and B[k = 8 h ..][n = c] have the SAME lane layout - into
// LDS ([k-step of the pair][plane][h][column][16 B]: conflict-free b128 on both sides); wave w then owns the o-strip
// [32 w, 32 w + 32) against all eight i-tiles: 3 + 24 ds_read_b128 and 48 MFMAs per k-step.
typedef __bf16 bf16x8_gw __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned gw_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void gw_split8(const float *x, u32x4_s16 &p1, u32x4_s16 &p2, u32x4_s16 &p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        const unsigned h = gw_cvt_pk_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = gw_cvt_pk_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
        p1[q] = h; p2[q] = m; p3[q] = gw_cvt_pk_bf16(sa, sb);
    }
}
__device__ __forceinline__ void gw_mfma(f32x16 &c, u32x4_s16 a, u32x4_s16 b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_gw, a), __builtin_bit_cast(bf16x8_gw, b), c, 0, 0, 0);
}

constexpr int kWsPlane = 2 * 256 * 16;          // bytes of one operand plane of one k-step
__global__ __launch_bounds__(512) void linear_wgrad_256_split_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                                     const float *__restrict__ S, float *__restrict__ part_w,
                                                                     float *__restrict__ part_b, int64_t n,
                                                                     int64_t rows_per_chunk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 6 * kWsPlane];        // 96 KiB: 2 buffers x {A1..3, B1..3}
    __shared__ float s_b[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = tid & 255, half = tid >> 8;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r_hi = r_lo + rows_per_chunk < n ? r_lo + rows_per_chunk : n;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;
    const int n_steps = r_hi > r_lo ? (int)((r_hi - r_lo + 15) / 16) : 0;
    // a PAIR of k-steps (32 rows) is fetched at a time: 32 + 16 requests per thread in flight under the pair's 96 products
    struct Rows { float gv[2][8], sv[2][8], yv[2][8]; };
    const int n_pairs = (n_steps + 1) / 2;
    // every request is unconditional (rows past the chunk re-read its last row and are zeroed in stage()): a predicated
    // load is waited for where its value merges with the zero - 48 round trips in a row (measured: 0.128 ms for the fetches
    // alone)
    // (round 5: 48 requests per thread and pair, each with a clamped 64-bit row index, were half of the kernel's 615 VALU instructions
    // per pair - SQ_INSTS_VALU, tools/gpu.sh sqk - and the VALU work of the two waves of a SIMD is serial with their MFMA work: pairs
    // that lie inside the chunk - all but the last - take ONE base address per panel and constant offsets)
    auto fetch = [&](int pair, Rows &q) {
        const int64_t p0 = r_lo + 32 * (int64_t)pair;
        if (p0 + 32 <= r_hi) {                                      // wave-uniform
            const size_t base = (size_t)(p0 + 8 * half) * 256 + col;
            const float *gp = g + base, *sp = S + base, *yp = Y ? Y + base : nullptr;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    q.gv[u][e] = gp[(16 * u + e) * 256];
                    q.sv[u][e] = sp[(16 * u + e) * 256];
                    q.yv[u][e] = yp ? yp[(16 * u + e) * 256] : 1.f;
                }
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r0 = p0 + 16 * u + 8 * half;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int64_t r = r0 + e;
                r = r < r_hi ? r : r_hi - 1;
                q.gv[u][e] = g[r * 256 + col];
                q.sv[u][e] = S[r * 256 + col];
                q.yv[u][e] = Y ? Y[r * 256 + col] : 1.f;
            }
        }
    };
    auto stage = [&](const Rows &q, int pair) {
        const bool whole = r_lo + 32 * (int64_t)pair + 32 <= r_hi;    // wave-uniform: no row of the pair lies past the chunk
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r0 = r_lo + 32 * (int64_t)pair + 16 * u + 8 * half;
            float gz[8], sz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = whole || r0 + e < r_hi;
                gz[e] = (ok && q.yv[u][e] > 0.f) ? q.gv[u][e] : 0.f;         // gZ = g where Y > 0 (masked())
                sz[e] = ok ? q.sv[u][e] : 0.f;
                bsum += gz[e];
            }
            u32x4_s16 P[6];
            gw_split8(gz, P[0], P[1], P[2]);
            gw_split8(sz, P[3], P[4], P[5]);
            unsigned char *base = lds + u * 6 * kWsPlane + half * 4096 + col * 16;
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) *reinterpret_cast<u32x4_s16 *>(base + pl * kWsPlane) = P[pl];
        }
    };
    auto products = [&](int buf) {
        const unsigned char *rb = lds + buf * 6 * kWsPlane + (lane >> 5) * 4096 + (lane & 31) * 16;
        u32x4_s16 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const u32x4_s16 *>(rb + pl * kWsPlane + (32 * wave) * 16);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            u32x4_s16 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4_s16 *>(rb + (3 + pl) * kWsPlane + (32 * t) * 16);
            gw_mfma(acc[t], a[2], b[0]);          // pieces (3,1) (1,3) (2,2) (2,1) (1,2) (1,1): small terms first
            gw_mfma(acc[t], a[0], b[2]);
            gw_mfma(acc[t], a[1], b[1]);
            gw_mfma(acc[t], a[1], b[0]);
            gw_mfma(acc[t], a[0], b[1]);
            gw_mfma(acc[t], a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);       // (all 24 operand reads hoisted to the top cost 96 registers: spills)
        }
    };
    Rows q;
    if (n_pairs > 0) fetch(0, q);
    for (int pair = 0; pair < n_pairs; ++pair) {
        stage(q, pair);                                          // (waits for the pair's requests)
        __syncthreads();
        if (pair + 1 < n_pairs) fetch(pair + 1, q);
        products(0);
        products(1);
        __syncthreads();                                         // everyone has read the pair before it is overwritten
    }
    // D[m = o][n = i], o = 32 wave + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), i = 32 t + (lane & 31)
    float *pw = part_w + (size_t)blockIdx.x * 256 * 256;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int i = 32 * t + (lane & 31);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int oo = 32 * wave + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
            pw[(size_t)oo * 256 + i] = acc[t][rr];
        }
    }
    if (part_b) {
        s_b[tid] = bsum;
        __syncthreads();
        if (half == 0) part_b[(size_t)blockIdx.x * 256 + col] = s_b[col] + s_b[256 + col];
    }
}

// narrow shapes (encoder Linear(1, H), decoder Linear(H, 1), H < 16): one thread per output element of the chunk
__global__ __launch_bounds__(256) void linear_wgrad_small_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                                 const float *__restrict__ S, float *__restrict__ part_w,
                                                                 float *__restrict__ part_b, int64_t n, int Hi, int Ho,
                                                                 int64_t rows_per_chunk) {
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r_hi = r_lo + rows_per_chunk < n ? r_lo + rows_per_chunk : n;
    for (int e = threadIdx.x; e < Ho * Hi + Ho; e += 256) {
        float s = 0.f;
        if (e < Ho * Hi) {
            const int o = e / Hi, i = e - o * Hi;
            int64_t r = r_lo;
            for (; r + 8 <= r_hi; r += 8) {                     // eight rows' requests in flight, folded in row order
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = masked(g, Y, (r + u) * Ho + o); b[u] = S[(r + u) * Hi + i]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fmaf(a[u], b[u], s);
            }
            for (; r < r_hi; ++r) s = fmaf(masked(g, Y, r * Ho + o), S[r * Hi + i], s);
            part_w[(size_t)blockIdx.x * Ho * Hi + e] = s;
        } else if (part_b) {
            const int o = e - Ho * Hi;
            for (int64_t r = r_lo; r < r_hi; ++r) s += masked(g, Y, r * Ho + o);
            part_b[(size_t)blockIdx.x * Ho + o] = s;
        }
    }
}

// out[e] = sum over chunks, ascending (fixed order)
// (acc != 0: out[e] = out[e] + scale * sum, product and sum rounded separately - the caller's running total of a reverse pass)
__global__ __launch_bounds__(256) void chunk_sum_kernel(const float *__restrict__ part, float *__restrict__ out, int n_elem,
                                                        int n_chunks, float scale = 1.f, int acc = 0) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= n_chunks; c += 8) {                        // eight requests in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * n_elem + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < n_chunks; ++c) s += part[(size_t)c * n_elem + e];
    out[e] = acc ? __fadd_rn(out[e], __fmul_rn(scale, s)) : s;
}

// g_W and g_b in ONE launch (a training step makes ~30 Linear backwards: one small launch less each)
__global__ __launch_bounds__(256) void chunk_sum2_kernel(const float *__restrict__ part_w, float *__restrict__ out_w, int n_w,
                                                         const float *__restrict__ part_b, float *__restrict__ out_b, int n_b,
                                                         int n_chunks, float scale = 1.f, int acc = 0) {
    int e = blockIdx.x * 256 + threadIdx.x;
    const float *part = part_w;
    float *out = out_w;
    int n_elem = n_w;
    if (e >= n_w) { e -= n_w; part = part_b; out = out_b; n_elem = n_b; }
    if (e >= n_elem) return;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= n_chunks; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * n_elem + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < n_chunks; ++c) s += part[(size_t)c * n_elem + e];
    out[e] = acc ? __fadd_rn(out[e], __fmul_rn(scale, s)) : s;
}

// ---------------------------------------------------------------------------------------------------- gS
// gS[n, Hi] = gZ[n, Ho] W[Ho, Hi]: the forward kernel's tiling (64 rows x BN columns per workgroup, k chunks of 32
// through padded LDS) with the mask applied while the A tile is staged and W staged transposed.
constexpr int kBM2 = 64, kBK2 = 32, kLd2b = kBK2 + 1;

template <int BN>
__global__ __launch_bounds__(256) void linear_gs_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                        const float *__restrict__ W, float *__restrict__ gS, int64_t n,
                                                        int Hi, int Ho) {
    constexpr int NT = BN / 64;
    __shared__ float s_A[kBM2 * kLd2b];
    __shared__ float s_B[BN * kLd2b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * kBM2;
    const int col0 = blockIdx.y * BN;                              // output column = input feature i
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int k0 = 0; k0 < Ho; k0 += kBK2) {                        // reduction over the output features o
        for (int i = tid; i < kBM2 * kBK2; i += 256) {
            const int r = i >> 5, q = i & 31;
            const int64_t gr = row0 + r;
            const int go = k0 + q;
            s_A[r * kLd2b + q] = (gr < n && go < Ho) ? masked(g, Y, gr * Ho + go) : 0.f;
        }
        // B[k = o][col = i] = W[o][i]; stored as s_B[col][k]: threads walk i fastest (coalesced rows of W)
        for (int i = tid; i < BN * kBK2; i += 256) {
            const int q = i / BN, c = i - q * BN;
            const int go = k0 + q, gi = col0 + c;
            s_B[c * kLd2b + q] = (go < Ho && gi < Hi) ? W[(int64_t)go * Hi + gi] : 0.f;
        }
        __syncthreads();
        const float *pa = s_A + (wm * 32 + (lane & 31)) * kLd2b + (lane >> 5);
        const float *pb = s_B + (wn * (BN / 2) + (lane & 31)) * kLd2b + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < kBK2 / 2; ++ks) {
            const float a = pa[ks * 2];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[t * 32 * kLd2b + ks * 2], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int gi = col0 + wn * (BN / 2) + t * 32 + (lane & 31);
        if (gi >= Hi) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr < n) gS[gr * Hi + gi] = acc[t][r];
        }
    }
}

// gS for Hi = Ho = 256 (the ODEFunc's Linear under autograd) on the fp16 matrix cores with fp32-grade results: the product of
// the forward kernels (split16.h: two fp16 pieces per operand behind a per-row power-of-two scale, three partial products in one
// fp32 accumulator) with B = W instead of W^T.  One workgroup = 64 rows: the masked gZ tile is staged in LDS once (each wave
// 16 rows: mask, row maximum by DPP, scale), wave w then owns output columns [64 w, 64 w + 64) of both 32-row m-tiles; the
// packed weights (pack_weight_256_t16: planes of W^T in MFMA B-operand order, 256 KiB, L2-resident) come through a four-k-step
// register ring.  fp32 MFMA version (linear_gs_kernel<256>): 0.44 ms at n = 10^5; this one is bound by its 0.3 GB of HBM traffic.
constexpr int kGsLd = 260;
// MT m-tiles of 32 rows per workgroup.  MT = 2 (64 rows, rounds 3-4) leaves room for two workgroups per CU (67 KB of LDS each): the load
// phase of a workgroup (two round trips to HBM) and its product phase do not overlap, and two workgroups per CU hide little of either
// (0.104 ms for the 0.31 GB of n = 10^5: 0.39 of the HBM peak).  MT = 1 (32 rows, 33.5 KB, 98 registers) runs four to five workgroups per
// CU in different phases; the price - the planes of W^T cross L2 -> CU once per 32 rows instead of 64 - is paid out of the L2's 34 TB/s.
template <int MT>
__global__ __launch_bounds__(256) void linear_gs_256_split_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                                  const void *__restrict__ Wq, float *__restrict__ gS, int64_t n) {
    constexpr int kRows = 32 * MT, kRpw = 8 * MT;             // rows per workgroup / per wave in the load phase
    __shared__ __attribute__((aligned(16))) float s_A[kRows * kGsLd];
    __shared__ float s_sc[kRows], s_un[kRows];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    const float *w_unscale = reinterpret_cast<const float *>(reinterpret_cast<const char *>(Wq) + kS16Bytes);     // [256]: per output column
    // eight rows' requests fly together, unconditionally (rows past n re-read row n - 1 and are zeroed afterwards)
#pragma unroll
    for (int i0 = 0; i0 < kRpw; i0 += 8) {
        f32x4 gv[8], yv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t gr = row0 + kRpw * wave + i0 + i;
            gr = gr < n ? gr : n - 1;
            gv[i] = *reinterpret_cast<const f32x4 *>(g + gr * 256 + 4 * lane);
            yv[i] = Y ? *reinterpret_cast<const f32x4 *>(Y + gr * 256 + 4 * lane) : (f32x4){1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = kRpw * wave + i0 + i;
            const bool ok = row0 + r < n;
            f32x4 v;
            v.x = (ok && yv[i].x > 0.f) ? gv[i].x : 0.f;
            v.y = (ok && yv[i].y > 0.f) ? gv[i].y : 0.f;
            v.z = (ok && yv[i].z > 0.f) ? gv[i].z : 0.f;
            v.w = (ok && yv[i].w > 0.f) ? gv[i].w : 0.f;
            *reinterpret_cast<f32x4 *>(s_A + r * kGsLd + 4 * lane) = v;
            unsigned sb, ub;
            s16_scale_bits(s16_wave_umax(s16_row_max_bits(v)), sb, ub);
            if (lane == 0) {
                s_sc[r] = __builtin_bit_cast(float, sb);
                s_un[r] = __builtin_bit_cast(float, ub);
            }
        }
    }
    constexpr int kRingQ = 4, kPl = kS16Planes;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(Wq), 0, kS16Bytes, 0x00020000);
    const int lane_off = lane * 16;
    const int q_slab = (2 * wave) * 16 * kPl * 1024;
    auto ldq = [&](int jj, int ks, int pl) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsQ, lane_off, q_slab + ((jj * 16 + ks) * kPl + pl) * 1024, 0);
    };
    u32x4_s16 Bq[kRingQ][2][kPl];
#pragma unroll
    for (int u = 0; u < kRingQ; ++u)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, u, pl);
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][jj][i] = 0.f;
    __syncthreads();
    const float *ap = s_A + (lane & 31) * kGsLd + 8 * (lane >> 5);
    float sc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sc[mt] = s_sc[32 * mt + (lane & 31)];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int u = ks % kRingQ;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(ap + mt * 32 * kGsLd + 16 * ks);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(ap + mt * 32 * kGsLd + 16 * ks + 4);
            u32x4_s16 A0, A1;
            s16_split8(r0, r1, sc[mt], A0, A1);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                s16_mfma(acc[mt][jj], A1, Bq[u][jj][0]);
                s16_mfma(acc[mt][jj], A0, Bq[u][jj][1]);
                s16_mfma(acc[mt][jj], A0, Bq[u][jj][0]);
            }
        }
        if (ks + kRingQ < 16) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < kPl; ++pl) Bq[u][jj][pl] = ldq(jj, ks + kRingQ, pl);
        }
    }
    // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31], multiplied back by 1 / (column n's weight-row scale), then 1 / (row scale)
    const float wu[2] = {w_unscale[64 * wave + (lane & 31)], w_unscale[64 * wave + 32 + (lane & 31)]};
    float *out0 = gS + row0 * 256 + 64 * wave + (lane & 31);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row0 + m >= n) continue;
            const float un = s_un[m];
            float *op = out0 + m * 256;                              // (one 64-bit base per lane, 32-bit row offsets)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) op[32 * jj] = (acc[mt][jj][r] * wu[jj]) * un;
        }
}

// gS for Hi = Ho = 256, third form (round 5): PERSISTENT workgroups with the weights resident.  The tile kernels above stream the
// 256 KiB of W^T planes from L2 once per 32 / 64 rows - 0.8 GB per launch at n = 10^5, four fifths of what the vector-memory path of a CU
// carries (TCP_TCC_READ_REQ: 8 M requests against 1.6 M for the panels; SQ_VMEM_TA_*_FIFO_FULL ~ the kernel's whole busy time, tools/gpu.sh
// sqk) - and their load phase and product phase alternate.  Here a workgroup of 8 waves walks row tiles of 32: wave w owns output columns
// [32 w, 32 w + 32) and keeps ALL 16 k-steps x 2 planes of its weights in registers (128 VGPRs) for the whole launch; every wave loads
// 4 rows of the NEXT tile (g and the mask panel: 8 requests in flight under the current tile's products), masks them, forms the row
// maximum by DPP and writes the row as its two fp16 pieces into the other half of a double buffer - the row layout and operand
// addressing of rhs_fused3 (32 x 1040 bytes).  Same pieces, same three products per k-step in the same order: bit-identical to the tile
// kernels.
constexpr int kGrLd = 260;                                   // floats per tile row: [256 fp16 high | 256 fp16 low] + 16 bytes
// MASK: Y is given (gZ = g (.) [Y > 0] formed here); false: g is gZ already (the tape's pull kernel masked it) - no mask registers
template <bool MASK>
__global__ __launch_bounds__(512) void linear_gs_256_res_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                                const void *__restrict__ Wq, float *__restrict__ gS, int64_t n,
                                                                int n_tiles) {
    __shared__ __attribute__((aligned(16))) float s_A[2][32 * kGrLd];
    __shared__ float s_un[2][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *w_unscale = reinterpret_cast<const float *>(reinterpret_cast<const char *>(Wq) + kS16Bytes);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(Wq), 0, kS16Bytes, 0x00020000);
    u32x4_s16 B[16][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            B[ks][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, lane * 16, ((wave * 16 + ks) * 2 + pl) * 1024, 0);
    const float wu = w_unscale[32 * wave + (lane & 31)];
    // TWO tiles of requests in flight (round 6): with one, a CU had 32-64 KiB outstanding - 8-16 MB over the chip, below what the HBM's
    // latency x rate asks for (M: the launch streamed at 4.1 TB/s of its 2 panels); the weights leave 64 registers for the second set
    struct Rows { f32x4 gv[4], yv[MASK ? 4 : 1]; };
    Rows ra, rb;
    // this wave's 4 rows of `tile` - contiguous: through buffer descriptors of the panels (n x 1 KiB < 4 GiB: launcher check) with the
    // row offset on the scalar unit and the lane offset in ONE register for every request of the kernel; rows past n read as zeros
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, (int)(unsigned)(n * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MASK ? Y : g), 0, (int)(unsigned)(n * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(gS, 0, (int)(unsigned)(n * 1024), 0x00020000);
    auto request = [&](int tile, Rows &q) {
        // (the row offset rides in the VECTOR offset: a raw buffer's range check covers vector + immediate offset, not the scalar one)
        const int vo = (tile * 32 + 4 * wave) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q.gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo + 1024 * i, 0, 2));
            if constexpr (MASK) q.yv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, vo + 1024 * i, 0, 2));
        }
    };
    auto stage = [&](int tile, int buf, const Rows &q) {     // mask, scale, split, write the pieces
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * wave + i;
            const bool ok = (int64_t)tile * 32 + r < n;
            f32x4 v;
            if constexpr (MASK) {
                v.x = (ok && q.yv[i].x > 0.f) ? q.gv[i].x : 0.f;
                v.y = (ok && q.yv[i].y > 0.f) ? q.gv[i].y : 0.f;
                v.z = (ok && q.yv[i].z > 0.f) ? q.gv[i].z : 0.f;
                v.w = (ok && q.yv[i].w > 0.f) ? q.gv[i].w : 0.f;
            } else {
                v.x = ok ? q.gv[i].x : 0.f;
                v.y = ok ? q.gv[i].y : 0.f;
                v.z = ok ? q.gv[i].z : 0.f;
                v.w = ok ? q.gv[i].w : 0.f;
            }
            unsigned sb, ub;
            s16_scale_bits(s16_wave_umax(s16_row_max_bits(v)), sb, ub);
            u32x2_s16 h0, h1;
            s16_split4(v, __builtin_bit_cast(float, sb), h0, h1);
            char *hrow = reinterpret_cast<char *>(s_A[buf] + r * kGrLd) + 8 * lane;
            *reinterpret_cast<u32x2_s16 *>(hrow) = h0;
            *reinterpret_cast<u32x2_s16 *>(hrow + 512) = h1;
            if (lane == 0) s_un[buf][r] = __builtin_bit_cast(float, ub);
        }
    };
    const int G = gridDim.x;
    // one tile: the products of `tile` out of buffer `buf`, then the pieces of tile + G (in `cur`) into the other buffer; `nxt` receives
    // the requests of tile + 2 G first (a third set - the unmasked form has the registers - measured no faster: 0.476 vs 0.468 ms at M;
    // per tile the launch is then at its matrix-pipe + split cycles, which add up on a SIMD like the weight gradient's)
    constexpr int kAhead = 2;
    auto one_tile = [&](int tile, int buf, Rows &cur, Rows &nxt) {
        const int next = tile + G;
        if (tile + kAhead * G < n_tiles) request(tile + kAhead * G, nxt);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const char *ap = reinterpret_cast<const char *>(s_A[buf]) + (lane & 31) * (kGrLd * 4) + 16 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const u32x4_s16 A0 = *reinterpret_cast<const u32x4_s16 *>(ap + 32 * ks);
            const u32x4_s16 A1 = *reinterpret_cast<const u32x4_s16 *>(ap + 32 * ks + 512);
            s16_mfma(acc, A1, B[ks][0]);
            s16_mfma(acc, A0, B[ks][1]);
            s16_mfma(acc, A0, B[ks][0]);
            if (ks & 1) __builtin_amdgcn_sched_barrier(0);     // (all 32 operand reads hoisted to the top cost 128 registers: spills)
        }
        // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]: column scale, then row scale (both exact)
        // stores through the output's descriptor: rows past n fall outside its range and are dropped - no branch per store, ONE offset
        // register (the per-store 64-bit addresses used to spill, and their reloads waited for every request in flight)
        const int o0 = (tile * 32 + 4 * (lane >> 5)) * 1024 + (32 * wave + (lane & 31)) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (acc[r] * wu) * s_un[buf][m]), rsS, o0 + 1024 * ((r & 3) + 8 * (r >> 2)), 0, 2);
        }
        if (next < n_tiles) stage(next, buf ^ 1, cur);
        __syncthreads();
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) { request(tile, ra); stage(tile, 0, ra); }
    if (tile + G < n_tiles) request(tile + G, ra);
    __syncthreads();
    // (unrolled by two: the register sets alternate with static names)
    for (; tile < n_tiles; tile += 2 * G) {
        one_tile(tile, 0, ra, rb);
        if (tile + G < n_tiles) one_tile(tile + G, 1, rb, ra);
    }
}

__global__ __launch_bounds__(256) void linear_gs_small_kernel(const float *__restrict__ g, const float *__restrict__ Y,
                                                              const float *__restrict__ W, float *__restrict__ gS, int64_t n,
                                                              int Hi, int Ho) {
    const int64_t total = n * (int64_t)Hi;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / Hi;
        const int i = (int)(e - r * Hi);
        float s = 0.f;
        for (int o = 0; o < Ho; ++o) s = fmaf(masked(g, Y, r * Ho + o), W[(int64_t)o * Hi + i], s);
        gS[e] = s;
    }
}

// narrow shapes (one thread per output element, serial over the chunk's rows) get their parallelism from the number of
// chunks: 4096 instead of 256 (decoder Linear(256, 1) over 10^6 rows: 1.2 -> 0.2 ms); their partial blocks are tiny
static int64_t wgrad_chunks(int64_t n, bool small) {
    // (few rows - the reference's 400 nodes: a chunk is two rounds of dependent loads instead of eight; the launch was 12.9 us, a sixth
    // of the kernel time of a README-sized dopri5 backward pass)
    int64_t c = n <= 4096 ? (n + 15) / 16 : (n + 63) / 64;
    const int64_t cap = small ? 4096 : kGwMaxChunks;
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}

static int64_t wgrad_work_bytes(int64_t n, int Hi, int Ho) {
    return (wgrad_chunks(n, Hi < 16 || Ho < 16) * ((int64_t)Ho * Hi + Ho) * (int64_t)sizeof(float) + 256 + 255) / 256 * 256;
}

// partial gW / gb blocks, then (Hi = Ho = 256) the packed fp16 planes of W^T for the gS product
int64_t linear_bwd_work_bytes(int64_t n, int Hi, int Ho) {
    return wgrad_work_bytes(n, Hi, Ho) + ((Hi == 256 && Ho == 256) ? (int64_t)kS16Bytes + kS16TailBytes + kS16GuardBytes : 0);
}

int linear_bwd_f32(const float *g, const float *Y, const float *S, const float *W, float *gS, float *gW, float *gb, void *work,
                   int64_t n, int Hi, int Ho, hipStream_t st, uint32_t flags, float acc_scale, bool accumulate) {
    // accumulate: gW / gb hold a running total: total + acc_scale * (this call's), each rounded on its own (the native reverse passes
    // of tape.hip: one small launch per evaluation instead of three)
    if (n == 0 && accumulate) return NDCN_OK;
    if (n == 0) {
        if (gW) NDCN_HIP(hipMemsetAsync(gW, 0, (size_t)Ho * Hi * sizeof(float), st));
        if (gb) NDCN_HIP(hipMemsetAsync(gb, 0, (size_t)Ho * sizeof(float), st));
        return NDCN_OK;
    }
    const bool small = Hi < 16 || Ho < 16;
    if (gS) {
        ProfScope prof(PROF_LINEAR_GS, st, 4.0 * n * (double)(Hi + Ho * (Y ? 2 : 1)) + 4.0 * Hi * Ho, 2.0 * n * (double)Hi * Ho);
        static const bool split_on = [] { const char *e = getenv("NDCN_GS_SPLIT"); return !(e && e[0] == '0'); }();
        const bool a16 = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(gS) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
        bool split = !small && split_on && Hi == 256 && Ho == 256 && work && a16;
        void *Wq = split ? static_cast<char *>(work) + wgrad_work_bytes(n, Hi, Ho) : nullptr;
        if (split) {
            int rcp = (flags & NDCN_F_PACKED) ? NDCN_OK : pack_weight_256_t16(W, Wq, st);
            if (rcp) return rcp;
            // range guard (split16.h): a COLUMN of W spanning more than the split product guarantees (the rows of the W^T image) sends
            // gS = gZ W to the fp32 matrix cores below, like the forward's NDCN_PATH_EXACT32
            if (weights_wide_range(Wq)) split = false;
        }
        if (small) {
            hipLaunchKernelGGL(linear_gs_small_kernel, dim3(stream_grid(n * (int64_t)Hi, 256)), dim3(256), 0, st, g, Y, W, gS, n, Hi, Ho);
        } else if (split) {
            // (a caller without scratch - gS only, older bindings - keeps the fp32 MFMA kernel below)
            static const int gs_rows = [] { const char *e = getenv("NDCN_GS_ROWS"); return e ? atoi(e) : 0; }();     // 0: resident weights (default); 32 / 64: the tile kernels
            const int n_tiles = (int)((n + 31) / 32);
            if (gs_rows != 32 && gs_rows != 64 && n * (int64_t)1024 < (1ll << 32)) {
                if (Y) hipLaunchKernelGGL(linear_gs_256_res_kernel<true>, dim3((unsigned)(n_tiles < kCus ? n_tiles : kCus)), dim3(512), 0, st, g, Y, Wq, gS, n, n_tiles);
                else hipLaunchKernelGGL(linear_gs_256_res_kernel<false>, dim3((unsigned)(n_tiles < kCus ? n_tiles : kCus)), dim3(512), 0, st, g, Y, Wq, gS, n, n_tiles);
            }
            else if (gs_rows != 32) hipLaunchKernelGGL(linear_gs_256_split_kernel<2>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, g, Y, Wq, gS, n);
            else hipLaunchKernelGGL(linear_gs_256_split_kernel<1>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, g, Y, Wq, gS, n);
        } else {
            const unsigned gx = (unsigned)((n + kBM2 - 1) / kBM2);
            if (Hi > 128) hipLaunchKernelGGL((linear_gs_kernel<256>), dim3(gx, (unsigned)((Hi + 255) / 256)), dim3(256), 0, st, g, Y, W, gS, n, Hi, Ho);
            else if (Hi > 64) hipLaunchKernelGGL((linear_gs_kernel<128>), dim3(gx, (unsigned)((Hi + 127) / 128)), dim3(256), 0, st, g, Y, W, gS, n, Hi, Ho);
            else hipLaunchKernelGGL((linear_gs_kernel<64>), dim3(gx, (unsigned)((Hi + 63) / 64)), dim3(256), 0, st, g, Y, W, gS, n, Hi, Ho);
        }
        NDCN_LAUNCH_CHECK();
    }
    if (gW || gb) {
        if (!work || !S) { set_error("linear_bwd: scratch of ndcn_linear_bwd_work_bytes() bytes and the forward input are required for gW / gb"); return NDCN_EINVAL; }
        const int64_t chunks = wgrad_chunks(n, small);
        const int64_t rpc0 = (n + chunks - 1) / chunks;
        const int64_t rpc = (rpc0 + 7) / 8 * 8;                    // whole rounds of 8 rows
        const int64_t used = (n + rpc - 1) / rpc;
        float *part_w = static_cast<float *>(work);
        float *part_b = part_w + (size_t)used * Ho * Hi;
        ProfScope prof(PROF_LINEAR_WGRAD, st, 4.0 * n * (double)(Hi + Ho * (Y ? 2 : 1)) + 4.0 * (used + 1) * (double)Hi * Ho, 2.0 * n * (double)Hi * Ho);
        static const bool wsplit_on = [] { const char *e = getenv("NDCN_GW_SPLIT"); return !(e && e[0] == '0'); }();
        if (small) {
            hipLaunchKernelGGL(linear_wgrad_small_kernel, dim3((unsigned)used), dim3(256), 0, st, g, Y, S, part_w, gb ? part_b : nullptr, n, Hi, Ho, rpc);
        } else if (wsplit_on && Hi == 256 && Ho == 256) {
            // (a role-split form - 4 fetch / split waves + 8 product waves - was built and measured in round 5: no faster, because a
            // SIMD's VALU work and MFMA work add up on gfx950 whichever wave issues them: profiles/r05_wgrad_roles.txt)
            hipLaunchKernelGGL(linear_wgrad_256_split_kernel, dim3((unsigned)used), dim3(512), 0, st, g, Y, S, part_w,
                               gb ? part_b : nullptr, n, rpc);
        } else {
            const dim3 grid((unsigned)used, (unsigned)((Ho + 255) / 256), (unsigned)((Hi + 255) / 256));
            const int ni = Hi >= 256 ? 8 : (Hi + 31) / 32;
#define NDCN_GW(NI_) hipLaunchKernelGGL((linear_wgrad_kernel<NI_>), grid, dim3(64 * kGwWaves), 0, st, g, Y, S, part_w, gb ? part_b : nullptr, n, Hi, Ho, rpc)
            switch (ni) {
                case 1: NDCN_GW(1); break;
                case 2: NDCN_GW(2); break;
                case 3: NDCN_GW(3); break;
                case 4: NDCN_GW(4); break;
                case 5: NDCN_GW(5); break;
                case 6: NDCN_GW(6); break;
                case 7: NDCN_GW(7); break;
                default: NDCN_GW(8); break;
            }
#undef NDCN_GW
        }
        NDCN_LAUNCH_CHECK();
        if (gW && gb)
            hipLaunchKernelGGL(chunk_sum2_kernel, dim3((unsigned)((Ho * Hi + Ho + 255) / 256)), dim3(256), 0, st, part_w, gW, Ho * Hi, part_b, gb,
                               Ho, (int)used, acc_scale, accumulate ? 1 : 0);
        else if (gW) hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Ho * Hi + 255) / 256)), dim3(256), 0, st, part_w, gW, Ho * Hi, (int)used,
                                        acc_scale, accumulate ? 1 : 0);
        else hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Ho + 255) / 256)), dim3(256), 0, st, part_b, gb, Ho, (int)used, acc_scale,
                                accumulate ? 1 : 0);
        NDCN_LAUNCH_CHECK();
    }
    return NDCN_OK;
}

}  // namespace ndcn
