// Row-wise L1 normalisation of a panel: Y[r, :] = X[r, :] / max(sum_j |X[r, j]|, 1e-12), infinities zeroed
// (RowNorm / row_normalization of the reference's ode_gcn.py:9-26: F.normalize(X, p=1, dim=1) followed by
// X[isinf(X)] = 0).  HBM-bound: one read + one write of the panel; a wave owns a row (rows of H <= 1024 floats
// stay in registers between the reduction and the division, longer rows are read twice).
#include "common.h"

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float norm1(float x, float inv_or_den, bool) {
    const float y = x / inv_or_den;
    return (fabsf(y) == INFINITY) ? 0.f : y;
}

// H % 4 == 0, H <= 1024, 16-byte aligned panels: up to 4 float4 per lane held in registers
__global__ __launch_bounds__(256) void rownorm_vec_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t n_rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int h4 = H >> 2;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const f32x4 *xr = reinterpret_cast<const f32x4 *>(X + r * H);
        f32x4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < h4 ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += (fabsf(v[i].x) + fabsf(v[i].y)) + (fabsf(v[i].z) + fabsf(v[i].w));
        }
        const float den = fmaxf(wave_sum(s), 1e-12f);
        f32x4 *yr = reinterpret_cast<f32x4 *>(Y + r * H);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < h4)
                yr[c] = (f32x4){norm1(v[i].x, den, true), norm1(v[i].y, den, true), norm1(v[i].z, den, true), norm1(v[i].w, den, true)};
        }
    }
}

// any H: two passes over the row
__global__ __launch_bounds__(256) void rownorm_any_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t n_rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const float *xr = X + r * H;
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += fabsf(xr[c]);
        const float den = fmaxf(wave_sum(s), 1e-12f);
        float *yr = Y + r * H;
        for (int c = lane; c < H; c += 64) yr[c] = norm1(xr[c], den, true);
    }
}

// VJP of the above (training through ResBlock(normalize=True) / RowNorm, ode_gcn.py:50-57): with s = sum_j |x_j|,
// den = max(s, 1e-12), gm = g where the output is finite (the masked_fill entries pass no gradient):
//   gx_j = gm_j / den - sign(x_j) (sum_i gm_i x_i) / den^2      (second term only where the clamp is inactive, s >= 1e-12)
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float *__restrict__ G, const float *__restrict__ X,
                                                          float *__restrict__ GX, int64_t n_rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const float *xr = X + r * H, *gr = G + r * H;
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += fabsf(xr[c]);
        s = wave_sum(s);
        const float den = fmaxf(s, 1e-12f);
        float t = 0.f;
        for (int c = lane; c < H; c += 64) {
            const float x = xr[c];
            if (fabsf(x / den) != INFINITY) t += gr[c] * x;
        }
        t = (s >= 1e-12f) ? wave_sum(t) / (den * den) : 0.f;
        float *o = GX + r * H;
        for (int c = lane; c < H; c += 64) {
            const float x = xr[c];
            const float gm = (fabsf(x / den) != INFINITY) ? gr[c] : 0.f;
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            o[c] = gm / den - sg * t;
        }
    }
}

int row_l1_normalize_bwd_f32(const float *G, const float *X, float *GX, int64_t n_rows, int H, hipStream_t st) {
    if (n_rows == 0 || H == 0) return NDCN_OK;
    int64_t g = (n_rows + 3) / 4;
    if (g > (int64_t)kCus * 16) g = (int64_t)kCus * 16;
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)g), dim3(256), 0, st, G, X, GX, n_rows, H);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int row_l1_normalize_f32(const float *X, float *Y, int64_t n_rows, int H, hipStream_t st) {
    if (n_rows == 0 || H == 0) return NDCN_OK;
    int64_t g = (n_rows + 3) / 4;
    if (g > (int64_t)kCus * 16) g = (int64_t)kCus * 16;
    const bool vec = (H % 4 == 0) && H <= 1024 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0;
    if (vec) hipLaunchKernelGGL(rownorm_vec_kernel, dim3((unsigned)g), dim3(256), 0, st, X, Y, n_rows, H);
    else hipLaunchKernelGGL(rownorm_any_kernel, dim3((unsigned)g), dim3(256), 0, st, X, Y, n_rows, H);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
