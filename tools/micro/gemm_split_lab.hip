// GEMM laboratory (tuning aid, not product code): Y[N x 256] = S[N x 256] W^T with fp32 operands split into bf16 pieces
// (x = x1 + x2 + x3 exactly, 8 + 8 + 8 significand bits) and the partial products a_i b_j summed on the bf16 matrix
// cores (16x the fp32 MFMA rate): NT = 6 keeps the terms with i + j <= 4 (dropped: ~2^-24 |a b| per product), NT = 8
// drops only a_3 b_3 (~2^-32), NT = 9 is exact in the products.  Reports time and the error against fp64 next to the
// fp32-MFMA chain of the product library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/gemm_split_lab tools/micro/gemm_split_lab.hip \
//         -Lndcn_amd -l:libndcn_hip.so -Wl,-rpath,'$ORIGIN/../../ndcn_amd'
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../include/ndcn_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 consecutive fp32 -> three bf16x8 planes
__device__ __forceinline__ void split8(const float *x, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        const unsigned h = cvt_pk_bf16(a, b);
        const float ra = a - lo_f(h), rb = b - hi_f(h);
        const unsigned m = cvt_pk_bf16(ra, rb);
        const float sa = ra - lo_f(m), sb = rb - hi_f(m);
        p1[q] = h; p2[q] = m; p3[q] = cvt_pk_bf16(sa, sb);
    }
}

constexpr int kLd = 260;

// one 64-row tile per workgroup (4 waves, wave w owns output columns [64 w, 64 w + 64)); S tile staged in LDS
template <int NT>
__global__ __launch_bounds__(256) void gemm_split(const float *__restrict__ S, const u32x4 *__restrict__ Wp, float *__restrict__ Y, int n) {
    __shared__ __attribute__((aligned(16))) float s_S[64 * kLd];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (row0 + r < n) v = reinterpret_cast<const f32x4 *>(S)[(size_t)(row0 + r) * 64 + c];
        *reinterpret_cast<f32x4 *>(s_S + r * kLd + 4 * c) = v;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // packed weights: block (n-tile j, k-step s, plane p) = 64 lanes x 16 bytes
    const u32x4 *wb = Wp + (size_t)(2 * wave) * 16 * 3 * 64 + lane;
    for (int s = 0; s < 16; ++s) {
        u32x4 B[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) B[j][p] = wb[((size_t)j * 16 + s) * 3 * 64 + p * 64];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float raw[8];
            const float *src = s_S + (32 * mt + (lane & 31)) * kLd + 16 * s + 8 * (lane >> 5);
            *reinterpret_cast<f32x4 *>(raw) = *reinterpret_cast<const f32x4 *>(src);
            *reinterpret_cast<f32x4 *>(raw + 4) = *reinterpret_cast<const f32x4 *>(src + 4);
            u32x4 A[3];
            split8(raw, A[0], A[1], A[2]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // small terms first
                auto mm = [&](int ia, int ib) {
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[ia]), __builtin_bit_cast(bf16x8, B[j][ib]),
                                                                        acc[mt][j], 0, 0, 0);
                };
                if (NT >= 9) mm(2, 2);
                if (NT >= 8) { mm(1, 2); mm(2, 1); }
                mm(0, 2); mm(2, 0); mm(1, 1);
                mm(0, 1); mm(1, 0);
                mm(0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < n) Y[(size_t)row * 256 + 64 * wave + 32 * j + (lane & 31)] = acc[mt][j][r];
            }
}

// ---- fp16 x 2 pieces, 3 products (round-3 experiment): x = h0 + h1 + r, 11 + 11 significand bits.  fp16 has 5 exponent bits:
// rows are scaled by a power of two (exact) so that max |s| <= 1, weights likewise by one global power of two; the LOW pieces
// are stored scaled up by 2^11 (normal numbers wherever the high piece is) and summed in an accumulator of their own.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <bool SCALED>
__device__ __forceinline__ void split8_h(const float *x, float sc, u32x4 &p0, u32x4 &p1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q] * sc, b = x[2 * q + 1] * sc;
        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
        const float ra = (a - (float)h.x) * (SCALED ? 2048.f : 1.f), rb = (b - (float)h.y) * (SCALED ? 2048.f : 1.f);
        const f16x2 l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ra, rb));
        p0[q] = __builtin_bit_cast(unsigned, h);
        p1[q] = __builtin_bit_cast(unsigned, l);
    }
}

// Wh: [n-tile j][k-step s][plane p (2)][lane][8 fp16]; w_unscale = 1 / (global weight scale)
template <bool SCALED>
__global__ __launch_bounds__(256) void gemm_split_f16(const float *__restrict__ S, const u32x4 *__restrict__ Wh, float *__restrict__ Y, int n,
                                                      float w_unscale) {
    __shared__ __attribute__((aligned(16))) float s_S[64 * kLd];
    __shared__ float s_scale[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (row0 + r < n) v = reinterpret_cast<const f32x4 *>(S)[(size_t)(row0 + r) * 64 + c];
        *reinterpret_cast<f32x4 *>(s_S + r * kLd + 4 * c) = v;
    }
    __syncthreads();
    // row scale: 2^-e with 2^e > max |s| (one wave per 16 rows)
    for (int r = wave * 16; r < wave * 16 + 16; ++r) {
        float m = 0.f;
        for (int c = lane; c < 256; c += 64) m = fmaxf(m, fabsf(s_S[r * kLd + c]));
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0) {
            int e = 0;
            if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);       // m = f * 2^e, f in [0.5, 1)
            s_scale[r] = ldexpf(1.f, -e);
        }
    }
    __syncthreads();
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) { hi[a][b][i] = 0.f; lo[a][b][i] = 0.f; }
    const u32x4 *wb = Wh + (size_t)(2 * wave) * 16 * 2 * 64 + lane;
    for (int s = 0; s < 16; ++s) {
        u32x4 B[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) B[j][p] = wb[((size_t)j * 16 + s) * 2 * 64 + p * 64];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float raw[8];
            const int row = 32 * mt + (lane & 31);
            const float *src = s_S + row * kLd + 16 * s + 8 * (lane >> 5);
            *reinterpret_cast<f32x4 *>(raw) = *reinterpret_cast<const f32x4 *>(src);
            *reinterpret_cast<f32x4 *>(raw + 4) = *reinterpret_cast<const f32x4 *>(src + 4);
            u32x4 A0, A1;
            split8_h<SCALED>(raw, s_scale[row], A0, A1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                lo[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A1), __builtin_bit_cast(f16x8, B[j][0]), lo[mt][j], 0, 0, 0);
                lo[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), __builtin_bit_cast(f16x8, B[j][1]), lo[mt][j], 0, 0, 0);
                if (!SCALED) hi[mt][j] = lo[mt][j];                 // one accumulator: small terms first, then the big one on top
                hi[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), __builtin_bit_cast(f16x8, B[j][0]), hi[mt][j], 0, 0, 0);
                if (!SCALED) { lo[mt][j] = hi[mt][j]; }
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int row = row0 + lr;
                const float un = w_unscale / s_scale[lr];
                if (row < n) Y[(size_t)row * 256 + 64 * wave + 32 * j + (lane & 31)] = (SCALED ? (hi[mt][j][r] + lo[mt][j][r] * (1.f / 2048.f)) : hi[mt][j][r]) * un;
            }
}

static unsigned short bf16_rne(float x) {
    unsigned b; memcpy(&b, &x, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}
static float bf16_f(unsigned short h) { unsigned b = (unsigned)h << 16; float f; memcpy(&f, &b, 4); return f; }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;
    const int reps = 10;
    std::vector<float> S((size_t)n * 256), W(256 * 256);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (auto &x : S) x = rnd() * 4.f - 1.f;                       // like A X of U(0,1) states: O(1), mixed sign
    if (argc > 2) {                                                // wide dynamic range: rows scaled by 1e-8 .. 1e8, 1 element in 8 shrunk by 1e-5
        for (int r = 0; r < n; ++r) {
            const float rs = powf(10.f, (float)((r * 7) % 17) - 8.f);
            for (int k = 0; k < 256; ++k) S[(size_t)r * 256 + k] *= rs * (((r + k) % 8 == 0) ? 1e-5f : 1.f);
        }
    }
    for (auto &x : W) x = (rnd() - 0.5f) / 8.f;                    // nn.Linear default init range 1/sqrt(256)
    // packed split weights: [n-tile j][k-step s][plane p][lane][8 bf16]
    std::vector<unsigned short> Wp((size_t)8 * 16 * 3 * 64 * 8);
    for (int j = 0; j < 8; ++j) for (int st = 0; st < 16; ++st) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const float w = W[(size_t)(32 * j + (l & 31)) * 256 + 16 * st + 8 * (l >> 5) + e];
        const unsigned short h = bf16_rne(w);
        const float r1 = w - bf16_f(h);
        const unsigned short m = bf16_rne(r1);
        const float r2 = r1 - bf16_f(m);
        const unsigned short lo = bf16_rne(r2);
        const unsigned short pl[3] = {h, m, lo};
        for (int p = 0; p < 3; ++p) Wp[((((size_t)j * 16 + st) * 3 + p) * 64 + l) * 8 + e] = pl[p];
    }
    // fp16 split weights: one global power-of-two scale, low piece x 2^11
    float wmax = 0.f;
    for (auto x : W) wmax = std::fmax(wmax, std::fabs(x));
    int we = 0; (void)frexpf(wmax, &we);
    const float wscale = ldexpf(1.f, -we);
    std::vector<_Float16> Wh((size_t)8 * 16 * 2 * 64 * 8);
    for (int j = 0; j < 8; ++j) for (int st = 0; st < 16; ++st) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const float w = W[(size_t)(32 * j + (l & 31)) * 256 + 16 * st + 8 * (l >> 5) + e] * wscale;
        const _Float16 g0 = (_Float16)w;
        const _Float16 g1 = (_Float16)((w - (float)g0) * 2048.f);
        Wh[((((size_t)j * 16 + st) * 2 + 0) * 64 + l) * 8 + e] = g0;
        Wh[((((size_t)j * 16 + st) * 2 + 1) * 64 + l) * 8 + e] = g1;
    }
    void *dWh, *dWh1;
    HIPCHECK(hipMalloc(&dWh, Wh.size() * 2));
    HIPCHECK(hipMemcpy(dWh, Wh.data(), Wh.size() * 2, hipMemcpyHostToDevice));
    std::vector<_Float16> Wh1(Wh.size());
    for (int j = 0; j < 8; ++j) for (int st = 0; st < 16; ++st) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const float w = W[(size_t)(32 * j + (l & 31)) * 256 + 16 * st + 8 * (l >> 5) + e] * wscale;
        const _Float16 g0 = (_Float16)w;
        Wh1[((((size_t)j * 16 + st) * 2 + 0) * 64 + l) * 8 + e] = g0;
        Wh1[((((size_t)j * 16 + st) * 2 + 1) * 64 + l) * 8 + e] = (_Float16)(w - (float)g0);
    }
    HIPCHECK(hipMalloc(&dWh1, Wh1.size() * 2));
    HIPCHECK(hipMemcpy(dWh1, Wh1.data(), Wh1.size() * 2, hipMemcpyHostToDevice));
    float *dS, *dW, *dY, *dYref;
    void *dWp;
    HIPCHECK(hipMalloc(&dS, S.size() * 4)); HIPCHECK(hipMalloc(&dW, W.size() * 4)); HIPCHECK(hipMalloc(&dY, S.size() * 4));
    HIPCHECK(hipMalloc(&dYref, S.size() * 4)); HIPCHECK(hipMalloc(&dWp, Wp.size() * 2));
    HIPCHECK(hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dWp, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice));

    auto timeit = [&](const std::function<void()> &fn) {
        for (int i = 0; i < 2; ++i) fn();
        HIPCHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
        HIPCHECK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) fn();
        HIPCHECK(hipEventRecord(b, 0)); HIPCHECK(hipEventSynchronize(b));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b)); HIPCHECK(hipGetLastError());
        return ms / reps;
    };
    // fp64 reference on sampled rows
    const int step = n / 997 > 0 ? n / 997 : 1;
    std::vector<int> rows; for (int r = 0; r < n; r += step) rows.push_back(r);
    std::vector<double> ref(rows.size() * 256), mag(rows.size() * 256);
    for (size_t i = 0; i < rows.size(); ++i) for (int o = 0; o < 256; ++o) {
        double a = 0, m = 0;
        for (int k = 0; k < 256; ++k) { const double p = (double)S[(size_t)rows[i] * 256 + k] * (double)W[(size_t)o * 256 + k]; a += p; m += std::fabs(p); }
        ref[i * 256 + o] = a; mag[i * 256 + o] = m;
    }
    auto err = [&](const float *d, double &mx, double &rel) {
        std::vector<float> row(256);
        mx = rel = 0;
        for (size_t i = 0; i < rows.size(); ++i) {
            HIPCHECK(hipMemcpy(row.data(), d + (size_t)rows[i] * 256, 1024, hipMemcpyDeviceToHost));
            for (int o = 0; o < 256; ++o) {
                const double e = std::fabs((double)row[o] - ref[i * 256 + o]);
                if (e > mx) mx = e;
                if (e / mag[i * 256 + o] > rel) rel = e / mag[i * 256 + o];
            }
        }
    };
    double mx, rel;
    float ms = timeit([&] { ndcn_linear_f32(dS, dW, nullptr, dYref, n, 256, 256, 0, nullptr); });
    err(dYref, mx, rel);
    printf("{\"variant\": \"lib_linear_fp32_mfma\", \"ms\": %.4f, \"TFLOPs\": %.1f, \"max_abs_err\": %.3e, \"max_err_over_sum_abs\": %.3e}\n", ms, 2.0 * n * 65536 / ms / 1e9, mx, rel);
#define RUN(NT_)                                                                                                         \
    do {                                                                                                                 \
        ms = timeit([&] { hipLaunchKernelGGL((gemm_split<NT_>), dim3((n + 63) / 64), dim3(256), 0, 0, dS, (const u32x4 *)dWp, dY, n); }); \
        err(dY, mx, rel);                                                                                                \
        printf("{\"variant\": \"split_bf16_x%d\", \"ms\": %.4f, \"TFLOPs_equiv\": %.1f, \"max_abs_err\": %.3e, \"max_err_over_sum_abs\": %.3e}\n", NT_, ms, 2.0 * n * 65536 / ms / 1e9, mx, rel); \
    } while (0)
    RUN(6);
    RUN(8);
    RUN(9);
    ms = timeit([&] { hipLaunchKernelGGL(gemm_split_f16<false>, dim3((n + 63) / 64), dim3(256), 0, 0, dS, (const u32x4 *)dWh1, dY, n, 1.f / wscale); });
    err(dY, mx, rel);
    printf("{\"variant\": \"split_fp16_x3_unscaled_low_one_acc\", \"ms\": %.4f, \"max_abs_err\": %.3e, \"max_err_over_sum_abs\": %.3e}\n", ms, mx, rel);
    ms = timeit([&] { hipLaunchKernelGGL(gemm_split_f16<true>, dim3((n + 63) / 64), dim3(256), 0, 0, dS, (const u32x4 *)dWh, dY, n, 1.f / wscale); });
    err(dY, mx, rel);
    printf("{\"variant\": \"split_fp16_x3_scaled\", \"ms\": %.4f, \"TFLOPs_equiv\": %.1f, \"max_abs_err\": %.3e, \"max_err_over_sum_abs\": %.3e}\n", ms, 2.0 * n * 65536 / ms / 1e9, mx, rel);
    return 0;
}
