// Y[n, Ho] = S[n, Hi] W^T + b [relu]   -- nn.Linear semantics, W row-major [Ho, Hi].
//
// Replaces self.wt(x) (neural_dynamics.py:33) and the encoder / decoder Linears (:143-148).
// fp32 in, fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise an fmaf
// chain over k) at the fp32 peak of 157 TFLOP/s; gfx950 has no TF32-like shortcut.
//
// MFMA operand map (32x32x2, MI355X_MICROARCH / cdna_hip_programming 3):
//   A: lane l holds A[m = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][n = l & 31]
//   D: 16 regs, D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31]
// Here m = node row, n = output feature, k = input feature; A = S tile, B[k][n] = W[n][k].
#include "common.h"

namespace ndcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 64;      // node rows per workgroup
constexpr int kBK = 32;      // k chunk staged per iteration
constexpr int kLd = kBK + 1; // padded leading dimension: column-ish ds_read_b32 is conflict-free at stride 33

// BN: output features per workgroup (multiple of 64, <= 256).  4 waves as 2 (m) x 2 (n):
// each wave owns one 32-row m-tile and BN/64 n-tiles of 32.
template <int BN, bool VEC>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float *__restrict__ S, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ Y,
                                                          int64_t n, int Hi, int Ho, int relu) {
    constexpr int NT = BN / 64;                   // n-tiles per wave
    __shared__ float s_S[kBM * kLd];
    __shared__ float s_W[BN * kLd];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * kBM;
    const int col0 = blockIdx.y * BN;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int k0 = 0; k0 < Hi; k0 += kBK) {
        // ---- stage S[row0 .. row0+64) x [k0 .. k0+32) and W[col0 .. col0+BN) x [k0 .. k0+32), zero-filled
        if (VEC) {
            // 8 float4 per row
            for (int i = tid; i < kBM * (kBK / 4); i += 256) {
                const int r = i >> 3, q = i & 7;
                const int64_t gr = row0 + r;
                const int gk = k0 + q * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < n && gk < Hi) v = *reinterpret_cast<const float4 *>(S + gr * Hi + gk);
                float *d = s_S + r * kLd + q * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            for (int i = tid; i < BN * (kBK / 4); i += 256) {
                const int r = i >> 3, q = i & 7;
                const int go = col0 + r;
                const int gk = k0 + q * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (go < Ho && gk < Hi) v = *reinterpret_cast<const float4 *>(W + (int64_t)go * Hi + gk);
                float *d = s_W + r * kLd + q * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int i = tid; i < kBM * kBK; i += 256) {
                const int r = i >> 5, q = i & 31;
                const int64_t gr = row0 + r;
                const int gk = k0 + q;
                s_S[r * kLd + q] = (gr < n && gk < Hi) ? S[gr * Hi + gk] : 0.f;
            }
            for (int i = tid; i < BN * kBK; i += 256) {
                const int r = i >> 5, q = i & 31;
                const int go = col0 + r;
                const int gk = k0 + q;
                s_W[r * kLd + q] = (go < Ho && gk < Hi) ? W[(int64_t)go * Hi + gk] : 0.f;
            }
        }
        __syncthreads();
        // ---- 16 k-steps of 2
        const float *pa = s_S + (wm * 32 + (lane & 31)) * kLd + (lane >> 5);
        const float *pb = s_W + (wn * (BN / 2) + (lane & 31)) * kLd + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < kBK / 2; ++ks) {
            const float a = pa[ks * 2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = pb[t * 32 * kLd + ks * 2];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: + bias, relu, store.  lane holds column n = lane & 31 of rows (r&3) + 8 (r>>2) + 4 (lane>>5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int go = col0 + wn * (BN / 2) + t * 32 + (lane & 31);
        if (go >= Ho) continue;
        const float bv = bias ? bias[go] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr < n) {
                float v = acc[t][r] + bv;
                if (relu) v = relu_nan(v);
                Y[gr * Ho + go] = v;
            }
        }
    }
}

// Narrow shapes (encoder Linear(1,H), decoder Linear(H,1), H < 16): one thread per output element.
__global__ __launch_bounds__(256) void linear_small_kernel(const float *__restrict__ S, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ Y,
                                                           int64_t n, int Hi, int Ho, int relu) {
    const int64_t total = n * (int64_t)Ho;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Ho;
        const int o = (int)(i - r * Ho);
        const float *s = S + r * Hi;
        const float *w = W + (int64_t)o * Hi;
        float acc = 0.f;
        for (int k = 0; k < Hi; ++k) acc = fmaf(s[k], w[k], acc);
        if (bias) acc += bias[o];
        if (relu) acc = relu_nan(acc);
        Y[i] = acc;
    }
}

// Wide-in, narrow-out (the decoder Linear(H, 1), neural_dynamics.py:148, over every tick of the solution): a wave per row -
// the row is read once, coalesced (one thread per output walks its own 1 KiB row: 0.67 ms for 490 k rows of H = 256) - each
// lane folds its strided share of the dot product as an fma chain, the wave sums the 64 partials in a fixed butterfly.
template <int NV>          // NV = ceil(Hi / 64) <= 8
__global__ __launch_bounds__(256) void linear_rowdot_kernel(const float *__restrict__ S, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ Y,
                                                            int64_t n, int Hi, int Ho, int relu) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < n; r += n_waves) {
        float x[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = lane + 64 * v;
            x[v] = S[r * Hi + (k < Hi ? k : Hi - 1)];              // unconditional request; the surplus lanes are zeroed
            x[v] = k < Hi ? x[v] : 0.f;
        }
        for (int o = 0; o < Ho; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = lane + 64 * v;
                acc = fmaf(x[v], W[(int64_t)o * Hi + (k < Hi ? k : Hi - 1)], acc);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                if (bias) acc += bias[o];
                Y[r * Ho + o] = relu ? relu_nan(acc) : acc;
            }
        }
    }
}

int linear_f32(const float *S, const float *W, const float *b, float *Y, int64_t n, int Hi, int Ho, uint32_t flags,
               hipStream_t st) {
    if (n == 0) return NDCN_OK;
    const int relu = (flags & NDCN_F_RELU) ? 1 : 0;
    ProfScope prof(PROF_LINEAR, st, 4.0 * n * (double)(Hi + Ho) + 4.0 * Hi * Ho, 2.0 * n * (double)Hi * Ho);
    if (Ho < 16 && Hi >= 64 && Hi <= 512) {
        const int grid = stream_grid(n * 64, 256);
#define NDCN_RD(NV_) hipLaunchKernelGGL((linear_rowdot_kernel<NV_>), dim3(grid), dim3(256), 0, st, S, W, b, Y, n, Hi, Ho, relu)
        switch ((Hi + 63) / 64) {
            case 1: NDCN_RD(1); break;
            case 2: NDCN_RD(2); break;
            case 3: NDCN_RD(3); break;
            case 4: NDCN_RD(4); break;
            case 5: NDCN_RD(5); break;
            case 6: NDCN_RD(6); break;
            case 7: NDCN_RD(7); break;
            default: NDCN_RD(8); break;
        }
#undef NDCN_RD
        NDCN_LAUNCH_CHECK();
        return NDCN_OK;
    }
    if (Hi < 16 || Ho < 16) {
        const int64_t total = n * (int64_t)Ho;
        hipLaunchKernelGGL(linear_small_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, S, W, b, Y, n, Hi, Ho,
                           relu);
        NDCN_LAUNCH_CHECK();
        return NDCN_OK;
    }
    const bool vec = (Hi % 4 == 0) && aligned16(S) && aligned16(W);
    const dim3 block(256);
    const unsigned gx = (unsigned)((n + kBM - 1) / kBM);
#define NDCN_LIN(BN_)                                                                                          \
    do {                                                                                                       \
        const dim3 grid(gx, (unsigned)((Ho + (BN_)-1) / (BN_)));                                               \
        if (vec)                                                                                               \
            hipLaunchKernelGGL((linear_mfma_kernel<BN_, true>), grid, block, 0, st, S, W, b, Y, n, Hi, Ho, relu);  \
        else                                                                                                   \
            hipLaunchKernelGGL((linear_mfma_kernel<BN_, false>), grid, block, 0, st, S, W, b, Y, n, Hi, Ho, relu); \
    } while (0)
    if (Ho > 128) NDCN_LIN(256);
    else if (Ho > 64) NDCN_LIN(128);
    else NDCN_LIN(64);
#undef NDCN_LIN
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
