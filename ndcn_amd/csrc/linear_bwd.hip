// Backward of the Linear on the path:  Y = act(S W^T + b)   (neural_dynamics.py:33,36 under autograd - the reference
// trains by backpropagating through every solver step, heat_dynamics.py:333, dgnn.py:204 - and the encoder / decoder
// Linears :143-148).  With gZ = g (.) [Y > 0] when the ReLU output Y is given (mask fused into the operand loads, no
// separate pass), gZ = g otherwise:
//
//   gS [n, Hi]  = gZ W                fp32 MFMA GEMM, W read transposed while it is staged (no W^T copy)
//   gW [Ho, Hi] = gZ^T S              fp32 MFMA "split-K": the reduction runs over the n rows; each workgroup owns a
//                                     contiguous chunk of rows and writes one partial Ho x Hi block, summed afterwards in
//                                     a FIXED order (deterministic gradients, no atomics).  Both operands are read
//                                     straight from HBM in MFMA operand order: A[m = o][k = row] = gZ[row][o] and
//                                     B[k = row][n = i] = S[row][i] are 128-byte coalesced per half-wave as they lie.
//   gb [Ho]     = column sums of gZ   rides in the gW kernel (the A operand passes through the lanes anyway)
//
// MFMA operand map (32x32x2 f32): A lane l = A[m = l & 31][k = l >> 5]; B lane l = B[k = l >> 5][n = l & 31];
// D reg r = D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31].
#include "common.h"
#include "kernels.h"

namespace ndcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
            for (int64_t r = r_lo; r < r_hi; ++r) s = fmaf(masked(g, Y, r * Ho + o), S[r * Hi + i], s);
            part_w[(size_t)blockIdx.x * Ho * Hi + e] = s;
        } else if (part_b) {
            const int o = e - Ho * Hi;
            for (int64_t r = r_lo; r < r_hi; ++r) s += masked(g, Y, r * Ho + o);
            part_b[(size_t)blockIdx.x * Ho + o] = s;
        }
    }
}

// out[e] = sum over chunks, ascending (fixed order)
__global__ __launch_bounds__(256) void chunk_sum_kernel(const float *__restrict__ part, float *__restrict__ out, int n_elem,
                                                        int n_chunks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    float s = 0.f;
    for (int c = 0; c < n_chunks; ++c) s += part[(size_t)c * n_elem + e];
    out[e] = s;
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

static int64_t wgrad_chunks(int64_t n) {
    int64_t c = (n + 63) / 64;
    if (c > kGwMaxChunks) c = kGwMaxChunks;
    return c < 1 ? 1 : c;
}

int64_t linear_bwd_work_bytes(int64_t n, int Hi, int Ho) {
    return wgrad_chunks(n) * ((int64_t)Ho * Hi + Ho) * (int64_t)sizeof(float) + 256;
}

int linear_bwd_f32(const float *g, const float *Y, const float *S, const float *W, float *gS, float *gW, float *gb, void *work,
                   int64_t n, int Hi, int Ho, hipStream_t st) {
    if (n == 0) {
        if (gW) NDCN_HIP(hipMemsetAsync(gW, 0, (size_t)Ho * Hi * sizeof(float), st));
        if (gb) NDCN_HIP(hipMemsetAsync(gb, 0, (size_t)Ho * sizeof(float), st));
        return NDCN_OK;
    }
    const bool small = Hi < 16 || Ho < 16;
    if (gS) {
        ProfScope prof(PROF_LINEAR, st, 4.0 * n * (double)(Hi + Ho * (Y ? 2 : 1)) + 4.0 * Hi * Ho, 2.0 * n * (double)Hi * Ho);
        if (small) {
            hipLaunchKernelGGL(linear_gs_small_kernel, dim3(stream_grid(n * (int64_t)Hi, 256)), dim3(256), 0, st, g, Y, W, gS, n, Hi, Ho);
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
        const int64_t chunks = wgrad_chunks(n);
        const int64_t rpc0 = (n + chunks - 1) / chunks;
        const int64_t rpc = (rpc0 + 7) / 8 * 8;                    // whole rounds of 8 rows
        const int64_t used = (n + rpc - 1) / rpc;
        float *part_w = static_cast<float *>(work);
        float *part_b = part_w + (size_t)used * Ho * Hi;
        ProfScope prof(PROF_LINEAR, st, 4.0 * n * (double)(Hi + Ho * (Y ? 2 : 1)) + 4.0 * (used + 1) * (double)Hi * Ho, 2.0 * n * (double)Hi * Ho);
        if (small) {
            hipLaunchKernelGGL(linear_wgrad_small_kernel, dim3((unsigned)used), dim3(256), 0, st, g, Y, S, part_w, gb ? part_b : nullptr, n, Hi, Ho, rpc);
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
        if (gW) hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Ho * Hi + 255) / 256)), dim3(256), 0, st, part_w, gW, Ho * Hi, (int)used);
        if (gb) hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Ho + 255) / 256)), dim3(256), 0, st, part_b, gb, Ho, (int)used);
        NDCN_LAUNCH_CHECK();
    }
    return NDCN_OK;
}

}  // namespace ndcn
