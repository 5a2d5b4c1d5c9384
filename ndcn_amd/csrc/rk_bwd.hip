// Vector-Jacobian products of the dopri5 panel operations.  The reference's (old) torchdiffeq differentiates THROUGH its
// step-size controller: dt, the stage coefficients dt * beta, the error ratio, the initial-step norms and the
// interpolation abscissa are tensors with autograd history (rk_common.py:41-61, misc.py:84-170, interp.py:38-65), and
// the drivers train by plain backprop through the solver (heat_dynamics.py:333, dgnn.py:204).  Each kernel below is ONE
// pass that produces every panel gradient of its operation plus the inner products that become the gradients of the
// scalar inputs (fp64 partials per workgroup, summed in a fixed order: deterministic):
//
//   combine     y = y0 + sum_j c_j k_j                                   g_kj = c_j g ;  g_cj = <g, k_j> ;  (g_y0 = g)
//   error ratio r = mean(((sum_j c_j k_j) / tol)^2), tol = atol + rtol max(|y0|, |y1|)      (misc.py:146-157)
//   rms         o = ||(a - b) / (atol + |y| rtol)||_2 / sqrt(N)                               (misc.py:71-76,121-138)
//   dense out   o = a x^4 + b x^3 + c x^2 + d x + y0 with the dopri5 fit                      (dopri5.py:39-45, interp.py:21-65)
#include "common.h"
#include "kernels.h"

namespace ndcn {

constexpr int kBwdMaxK = 8;
constexpr int kBwdBlocks = 2048;
constexpr int kBwdDots = 8;

// acc (nullable each): a gradient the SAME tensor already received from the operations that consumed it later; the kernel
// writes gk = acc + (its own contribution) - the sum autograd would otherwise form with a pass of its own per consumer
// (torch `add` kernels were 24 % of a dopri5 training step).  gk may alias acc.
struct BwdTerms {
    const float *k[kBwdMaxK];
    float *gk[kBwdMaxK];           // nullable each
    const float *acc[kBwdMaxK];    // nullable each
    float c[kBwdMaxK];
    int n;
};

typedef float bw_f4 __attribute__((ext_vector_type(4)));
// (non-temporal: a backward pass streams dozens of panels once each - 100k-node Adam step 33.6 -> 32.5 ms in alternating runs)
__device__ __forceinline__ bw_f4 ld4(const float *p, int64_t i) { return __builtin_nontemporal_load(reinterpret_cast<const bw_f4 *>(p) + i); }
__device__ __forceinline__ void st4(float *p, int64_t i, bw_f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<bw_f4 *>(p) + i); }

// per-block sums of kBwdDots doubles -> partial[block][kBwdDots]
__device__ __forceinline__ void block_store_dots(double (&d)[kBwdDots], double *__restrict__ partial) {
    __shared__ double sh[4][kBwdDots];
#pragma unroll
    for (int q = 0; q < kBwdDots; ++q) {
        double v = d[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        d[q] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < kBwdDots; ++q) sh[w][q] = d[q];
    __syncthreads();
    if (threadIdx.x < kBwdDots)
        partial[(size_t)blockIdx.x * kBwdDots + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// one workgroup, the kBwdDots quantities side by side: 32 lanes each, lane l adds partials l, l + 32, .. in ascending order, then the 32
// lane sums meet through shuffles - a fixed order (deterministic), one pass.  (The first form took the quantities one after the other
// with a 256-thread tree each: 7.7 us per launch, 12 % of the kernel time of a README-sized dopri5 training step with its ~1 200 launches.)
__global__ __launch_bounds__(256) void dots_finish_kernel(const double *__restrict__ partial, int n_blocks, double *__restrict__ out) {
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    double s = 0.0;
    for (int i = l; i < n_blocks; i += 32 * 8) {                  // eight requests in flight, added in ascending order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + 32 * u < n_blocks ? partial[(size_t)(i + 32 * u) * kBwdDots + q] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
    if (l == 0) out[q] = s;
}

// (the partial-sum scratch holds kBwdBlocks slots; the streaming VJPs run best on half of them - measured on the 100k-node training
// step: combine / pull 3.6 ms against 4.15 with 2048 workgroups, the dense-output VJP, which reads 9 + nt panels and writes 9, the other
// way round: 2.30 -> 1.95 ms)
static int bwd_grid(int64_t n, int cap = kBwdBlocks / 2) {
    int g = stream_grid(n, 256);
    return g > cap ? cap : g;
}

// ------------------------------------------------------------------------------------------------ combine
// VEC: 16 bytes per lane (n counts float4 items then).  gy0 = acc_y0 + g when both are given (the identity branch of the sum).
template <bool VEC>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const float *__restrict__ g, BwdTerms t, float *gy0, const float *acc_y0,
                                                          int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const bw_f4 gv = ld4(g, i);
            if (gy0) st4(gy0, i, ld4(acc_y0, i) + gv);
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < t.n) {
                    const bw_f4 kv = ld4(t.k[j], i);
                    d[j] += (double)(gv.x * kv.x) + (double)(gv.y * kv.y) + (double)(gv.z * kv.z) + (double)(gv.w * kv.w);
                    if (t.gk[j]) {
                        bw_f4 o = t.c[j] * gv;
                        if (t.acc[j]) o = ld4(t.acc[j], i) + o;
                        st4(t.gk[j], i, o);
                    }
                }
        } else {
            const float gv = g[i];
            if (gy0) gy0[i] = acc_y0[i] + gv;
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < t.n) {
                    d[j] += (double)(gv * t.k[j][i]);
                    if (t.gk[j]) {
                        const float o = t.c[j] * gv;
                        t.gk[j][i] = t.acc[j] ? t.acc[j][i] + o : o;
                    }
                }
        }
    }
    block_store_dots(d, partial);
}

// ------------------------------------------------------------------------------------------------ error ratio
struct ErrBwdArgs {
    const float *y0, *y1;
    const float *acc_y0, *acc_y1;  // nullable
    float *gy0, *gy1;              // nullable
    BwdTerms t;
    float rtol, atol, g_r, inv_n;  // inv_n = 1 / numel (global count when the mean spans ranks)
};

// one element of the error-ratio VJP: s = d r / d e (the stage gradients are g_r c_j s, formed where they are stored: keeping
// them per component cost 32 registers of a 250-register kernel that ran at ONE wave per SIMD, 1.65 TB/s), the two state
// gradients through references (accumulators applied by the caller)
__device__ __forceinline__ float error_bwd_elem(const ErrBwdArgs &p, const float (&kv)[kBwdMaxK], float a0, float a1, double (&d)[kBwdDots],
                                                float &o0, float &o1) {
    float e = p.t.c[0] * kv[0];
#pragma unroll
    for (int j = 1; j < kBwdMaxK; ++j)
        if (j < p.t.n) e = e + p.t.c[j] * kv[j];
    const float m0 = fabsf(a0), m1 = fabsf(a1);
    const float tol = p.atol + p.rtol * max_nan(m0, m1);
    const float q = e / tol;
    const float s = 2.f * q * p.inv_n / tol;             // d r / d e
#pragma unroll
    for (int j = 0; j < kBwdMaxK; ++j)
        if (j < p.t.n) d[j] += (double)(s * kv[j]);       // d r / d c_j (times g_r on the host)
    // d r / d tol = -q s ; tol = atol + rtol max(|y0|, |y1|) ; torch.max splits the gradient evenly on ties
    const float gm = p.g_r * (-q * s) * p.rtol;
    const float w0 = m0 > m1 ? 1.f : (m0 == m1 ? 0.5f : 0.f);
    o0 = gm * w0 * (a0 > 0.f ? 1.f : (a0 < 0.f ? -1.f : 0.f));
    o1 = gm * (1.f - w0) * (a1 > 0.f ? 1.f : (a1 < 0.f ? -1.f : 0.f));
    return s;
}

// (waves_per_eu: with launch_bounds alone the scheduler hoists every load and takes 250 + 98 registers - one wave per SIMD)
template <bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void error_bwd_kernel(ErrBwdArgs p, int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int W = VEC ? 4 : 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float kv[W][kBwdMaxK], a0[W], a1[W], sv[W], o0[W], o1[W];
        if (VEC) {
            const bw_f4 v0 = ld4(p.y0, i), v1 = ld4(p.y1, i);
            a0[0] = v0.x; a0[W > 1 ? 1 : 0] = v0.y; a0[W > 1 ? 2 : 0] = v0.z; a0[W > 1 ? 3 : 0] = v0.w;
            a1[0] = v1.x; a1[W > 1 ? 1 : 0] = v1.y; a1[W > 1 ? 2 : 0] = v1.z; a1[W > 1 ? 3 : 0] = v1.w;
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < p.t.n) {
                    const bw_f4 k4 = ld4(p.t.k[j], i);
                    kv[0][j] = k4.x; kv[W > 1 ? 1 : 0][j] = k4.y; kv[W > 1 ? 2 : 0][j] = k4.z; kv[W > 1 ? 3 : 0][j] = k4.w;
                }
        } else {
            a0[0] = p.y0[i]; a1[0] = p.y1[i];
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < p.t.n) kv[0][j] = p.t.k[j][i];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) sv[u] = error_bwd_elem(p, kv[u], a0[u], a1[u], d, o0[u], o1[u]);
        if (VEC) {
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < p.t.n && p.t.gk[j]) {
                    const float cj = p.t.c[j];
                    bw_f4 o = {p.g_r * (cj * sv[0]), p.g_r * (cj * sv[W > 1 ? 1 : 0]), p.g_r * (cj * sv[W > 1 ? 2 : 0]), p.g_r * (cj * sv[W > 1 ? 3 : 0])};
                    if (p.t.acc[j]) o = ld4(p.t.acc[j], i) + o;
                    st4(p.t.gk[j], i, o);
                }
            if (p.gy0) {
                bw_f4 o = {o0[0], o0[W > 1 ? 1 : 0], o0[W > 1 ? 2 : 0], o0[W > 1 ? 3 : 0]};
                if (p.acc_y0) o = ld4(p.acc_y0, i) + o;
                st4(p.gy0, i, o);
            }
            if (p.gy1) {
                bw_f4 o = {o1[0], o1[W > 1 ? 1 : 0], o1[W > 1 ? 2 : 0], o1[W > 1 ? 3 : 0]};
                if (p.acc_y1) o = ld4(p.acc_y1, i) + o;
                st4(p.gy1, i, o);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kBwdMaxK; ++j)
                if (j < p.t.n && p.t.gk[j]) {
                    const float o = p.g_r * (p.t.c[j] * sv[0]);
                    p.t.gk[j][i] = p.t.acc[j] ? p.t.acc[j][i] + o : o;
                }
            if (p.gy0) p.gy0[i] = p.acc_y0 ? p.acc_y0[i] + o0[0] : o0[0];
            if (p.gy1) p.gy1[i] = p.acc_y1 ? p.acc_y1[i] + o1[0] : o1[0];
        }
    }
    block_store_dots(d, partial);
}

// ------------------------------------------------------------------------------------------------ rms
// o = ||v|| / sqrt(N), v = (a - b) / scale, scale = atol + |y| rtol ; coef = g_o / (||v|| sqrt(N))
__global__ __launch_bounds__(256) void rms_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ y,
                                                      float rtol, float atol, float coef, float *__restrict__ ga, float *__restrict__ gb,
                                                      float *__restrict__ gy, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float yv = y[i];
        const float scale = atol + fabsf(yv) * rtol;
        const float v = (b ? a[i] - b[i] : a[i]) / scale;
        const float gv = coef * v;
        const float g_a = gv / scale;
        const float g_y = -(gv * v / scale) * rtol * (yv > 0.f ? 1.f : (yv < 0.f ? -1.f : 0.f));
        if (ga) ga[i] = g_a;
        if (gy) gy[i] = g_y;                                 // (a may BE y, misc.py:123: the caller's autograd adds the two)
        if (gb) gb[i] = -g_a;
    }
}

// ------------------------------------------------------------------------------------------------ dense output
struct DenseBwdArgs {
    const float *g, *y0, *y1;
    const float *k[7];
    float *gy0, *gy1, *gk[7];      // nullable each
    const float *acc_y0, *acc_y1, *acc[7];   // nullable each: gradients already received (see BwdTerms)
    float cm[7];                   // dt * DPS_C_MID (fp32, as the forward)
    float cmid[7];                 // DPS_C_MID
    float dt, x;
};

__global__ __launch_bounds__(256) void dense_bwd_kernel(DenseBwdArgs p, int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float x = p.x, x2 = x * x, x3 = x2 * x, x4 = x3 * x, dt = p.dt;
    const float w_ym = 16.f * x4 - 32.f * x3 + 16.f * x2;
    const float w_y0 = (-8.f * x4 + 18.f * x3 - 11.f * x2 + 1.f) + w_ym;
    const float w_y1 = -8.f * x4 + 14.f * x3 - 5.f * x2;
    const float w_f0 = dt * (-2.f * x4 + 5.f * x3 - 4.f * x2 + x);
    const float w_f1 = dt * (2.f * x4 - 3.f * x3 + x2);
    float wk[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) wk[j] = w_ym * p.cm[j];
    wk[0] += w_f0;
    wk[6] += w_f1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = p.g[i];
        const float y0 = p.y0[i], y1 = p.y1[i];
        float kv[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) kv[j] = p.k[j][i];
        float sm = 0.f, sc = 0.f;                            // sum cm_j k_j (= ymid - y0) and sum cmid_j k_j (= d ymid / d dt)
#pragma unroll
        for (int j = 0; j < 7; ++j) { sm += p.cm[j] * kv[j]; sc += p.cmid[j] * kv[j]; }
        const float ym = y0 + sm, f0 = kv[0], f1 = kv[6];
        const float ca = 2.f * dt * (f1 - f0) - 8.f * y0 - 8.f * y1 + 16.f * ym;
        const float cb = dt * (5.f * f0 - 3.f * f1) + 18.f * y0 + 14.f * y1 - 32.f * ym;
        const float cc = dt * (f1 - 4.f * f0) - 11.f * y0 - 5.f * y1 + 16.f * ym;
        const float cd = dt * f0;
        d[0] += (double)(gv * (4.f * ca * x3 + 3.f * cb * x2 + 2.f * cc * x + cd));                      // d o / d x
        d[1] += (double)(gv * (x4 * (2.f * (f1 - f0) + 16.f * sc) + x3 * (5.f * f0 - 3.f * f1 - 32.f * sc) +
                               x2 * (f1 - 4.f * f0 + 16.f * sc) + x * f0));                              // d o / d dt
        if (p.gy0) p.gy0[i] = p.acc_y0 ? p.acc_y0[i] + w_y0 * gv : w_y0 * gv;
        if (p.gy1) p.gy1[i] = p.acc_y1 ? p.acc_y1[i] + w_y1 * gv : w_y1 * gv;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (p.gk[j]) p.gk[j][i] = p.acc[j] ? p.acc[j][i] + wk[j] * gv : wk[j] * gv;
    }
    block_store_dots(d, partial);
}

// Several ticks of ONE accepted step in one pass (the drivers sample 16-120 ticks over a handful of steps: heat_dynamics.py:35,123;
// dgnn.py:173-182): the step's nine panels and their received gradients are read once instead of once per tick.
//   gk_j = acc_j + sum_t wk_j(x_t) g_t,  gy0 / gy1 likewise;  dots: d[t] = <g_t, d o / d x_t> (t < nt <= 7), d[7] = sum_t <g_t, d o / d dt>
constexpr int kDenseMulti = 7;
struct DenseMultiArgs {
    const float *g[kDenseMulti];
    const float *y0, *y1;
    const float *k[7];
    float *gy0, *gy1, *gk[7];
    const float *acc_y0, *acc_y1, *acc[7];
    float cm[7], cmid[7];
    float dt, x[kDenseMulti];
    int nt;
};

__global__ __launch_bounds__(256) void dense_bwd_multi_kernel(DenseMultiArgs p, int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float dt = p.dt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float y0 = p.y0[i], y1 = p.y1[i];
        float kv[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) kv[j] = p.k[j][i];
        float sm = 0.f, sc = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) { sm += p.cm[j] * kv[j]; sc += p.cmid[j] * kv[j]; }
        const float ym = y0 + sm, f0 = kv[0], f1 = kv[6];
        const float ca = 2.f * dt * (f1 - f0) - 8.f * y0 - 8.f * y1 + 16.f * ym;
        const float cb = dt * (5.f * f0 - 3.f * f1) + 18.f * y0 + 14.f * y1 - 32.f * ym;
        const float cc = dt * (f1 - 4.f * f0) - 11.f * y0 - 5.f * y1 + 16.f * ym;
        const float cd = dt * f0;
        float o_y0 = 0.f, o_y1 = 0.f, o_k[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < kDenseMulti; ++t)
            if (t < p.nt) {
                const float x = p.x[t], x2 = x * x, x3 = x2 * x, x4 = x3 * x;
                const float w_ym = 16.f * x4 - 32.f * x3 + 16.f * x2;
                const float w_y0 = (-8.f * x4 + 18.f * x3 - 11.f * x2 + 1.f) + w_ym;
                const float w_y1 = -8.f * x4 + 14.f * x3 - 5.f * x2;
                const float w_f0 = dt * (-2.f * x4 + 5.f * x3 - 4.f * x2 + x);
                const float w_f1 = dt * (2.f * x4 - 3.f * x3 + x2);
                const float gv = p.g[t][i];
                d[t] += (double)(gv * (4.f * ca * x3 + 3.f * cb * x2 + 2.f * cc * x + cd));
                d[7] += (double)(gv * (x4 * (2.f * (f1 - f0) + 16.f * sc) + x3 * (5.f * f0 - 3.f * f1 - 32.f * sc) +
                                       x2 * (f1 - 4.f * f0 + 16.f * sc) + x * f0));
                o_y0 += w_y0 * gv;
                o_y1 += w_y1 * gv;
#pragma unroll
                for (int j = 0; j < 7; ++j) o_k[j] += (w_ym * p.cm[j]) * gv;
                o_k[0] += w_f0 * gv;
                o_k[6] += w_f1 * gv;
            }
        if (p.gy0) p.gy0[i] = p.acc_y0 ? p.acc_y0[i] + o_y0 : o_y0;
        if (p.gy1) p.gy1[i] = p.acc_y1 ? p.acc_y1[i] + o_y1 : o_y1;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (p.gk[j]) p.gk[j][i] = p.acc[j] ? p.acc[j][i] + o_k[j] : o_k[j];
    }
    block_store_dots(d, partial);
}

// ------------------------------------------------------------------------------------------------ host wrappers
int64_t rk_bwd_ws_bytes() { return (int64_t)kBwdBlocks * kBwdDots * sizeof(double); }

static int fill_bwd(BwdTerms &t, const float *const *h_k, float *const *h_gk, const float *h_c, int n_k, const float *const *h_acc) {
    if (n_k < 1 || n_k > kBwdMaxK) { set_error("rk backward: 1..%d terms", kBwdMaxK); return NDCN_EINVAL; }
    t.n = n_k;
    for (int j = 0; j < kBwdMaxK; ++j) {
        t.k[j] = j < n_k ? h_k[j] : h_k[0];
        t.gk[j] = (j < n_k && h_gk) ? h_gk[j] : nullptr;
        t.acc[j] = (j < n_k && h_acc && t.gk[j]) ? h_acc[j] : nullptr;
        t.c[j] = j < n_k ? h_c[j] : 0.f;
        if (j < n_k && !h_k[j]) { set_error("rk backward: null stage pointer"); return NDCN_EINVAL; }
    }
    return NDCN_OK;
}

int rk_combine_bwd_f32(const float *g, const float *const *h_k, const float *h_c, int n_k, float *const *h_gk, const float *const *h_acc,
                       float *gy0, const float *acc_y0, double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    BwdTerms t;
    int rc = fill_bwd(t, h_k, h_gk, h_c, n_k, h_acc);
    if (rc) return rc;
    if (!acc_y0) gy0 = nullptr;                              // without a received gradient g_y0 IS g: nothing to write
    bool vec = n % 4 == 0 && aligned16(g) && (!gy0 || (aligned16(gy0) && aligned16(acc_y0)));
    for (int j = 0; j < n_k; ++j) vec = vec && aligned16(t.k[j]) && (!t.gk[j] || aligned16(t.gk[j])) && (!t.acc[j] || aligned16(t.acc[j]));
    int n_out = gy0 ? 2 : 0;
    for (int j = 0; j < n_k; ++j) n_out += t.gk[j] ? (t.acc[j] ? 2 : 1) : 0;
    ProfScope prof(PROF_COMBINE_BWD, st, 4.0 * n * (n_k + 1 + n_out), 2.0 * n * n_k);
    const int grid = bwd_grid(vec ? n / 4 : n);
    if (vec) hipLaunchKernelGGL(combine_bwd_kernel<true>, dim3(grid), dim3(256), 0, st, g, t, gy0, acc_y0, n / 4, static_cast<double *>(d_ws));
    else hipLaunchKernelGGL(combine_bwd_kernel<false>, dim3(grid), dim3(256), 0, st, g, t, gy0, acc_y0, n, static_cast<double *>(d_ws));
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// ------------------------------------------------------------------------------------------------ <g, a - b>
// The gradient of a step size through ONE stage sum u = y0 + dt sum_j beta_j k_j is <g_u, sum_j beta_j k_j> = <g_u, u - y0> / dt:
// three panels read whatever the number of terms (the per-coefficient products of combine_bwd read one panel per term).
template <bool VEC>
__global__ __launch_bounds__(256) void dot_diff_kernel(const float *__restrict__ g, const float *__restrict__ a, const float *__restrict__ b,
                                                       int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const bw_f4 gv = ld4(g, i);
            bw_f4 e = ld4(a, i);
            if (b) e = e - ld4(b, i);
            d[0] += (double)(gv.x * e.x) + (double)(gv.y * e.y) + (double)(gv.z * e.z) + (double)(gv.w * e.w);
        } else {
            const float e = b ? a[i] - b[i] : a[i];
            d[0] += (double)(g[i] * e);
        }
    }
    block_store_dots(d, partial);
}

// ------------------------------------------------------------------------------------------------ pull (tape.hip)
// The total gradient of a stage derivative in the reverse pass of an attempted step, as ONE pass:
//   out = [mask > 0 ?] base + ((c_0 p_0 + c_1 p_1) + ...)        (rk_combine's order and rounding; base nullable; mask nullable: the ReLU
//                                                                 output of the evaluation this gradient enters - its VJP then reads no mask)
//   d_dots[0] = <p_0, ua - ub>                                    (ua nullable: no product; the step size's gradient through the stage sum
//                                                                 whose input's gradient p_0 is: rk_dot_diff_f32's sum)
// instead of a combine pass, a dot_diff pass and two reads of the mask by the Linear backward's kernels.
struct PullArgs {
    const float *base, *mask, *ua, *ub;
    const float *p[kBwdMaxK];
    float c[kBwdMaxK];
    int n;
    float *out;
};

template <bool VEC>
__global__ __launch_bounds__(256) void pull_kernel(PullArgs a, int64_t n, double *__restrict__ partial) {
    double d[kBwdDots] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const bw_f4 p0 = ld4(a.p[0], i);
            bw_f4 s;
            s.x = __fmul_rn(a.c[0], p0.x); s.y = __fmul_rn(a.c[0], p0.y); s.z = __fmul_rn(a.c[0], p0.z); s.w = __fmul_rn(a.c[0], p0.w);
#pragma unroll
            for (int j = 1; j < kBwdMaxK; ++j)
                if (j < a.n) {
                    const bw_f4 k = ld4(a.p[j], i);
                    const float c = a.c[j];
                    s.x = __fadd_rn(s.x, __fmul_rn(c, k.x)); s.y = __fadd_rn(s.y, __fmul_rn(c, k.y));
                    s.z = __fadd_rn(s.z, __fmul_rn(c, k.z)); s.w = __fadd_rn(s.w, __fmul_rn(c, k.w));
                }
            if (a.base) {
                const bw_f4 y = ld4(a.base, i);
                s.x = __fadd_rn(y.x, s.x); s.y = __fadd_rn(y.y, s.y); s.z = __fadd_rn(y.z, s.z); s.w = __fadd_rn(y.w, s.w);
            }
            if (a.mask) {
                const bw_f4 m = ld4(a.mask, i);
                s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f; s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
            }
            st4(a.out, i, s);
            if (a.ua) {
                bw_f4 e = ld4(a.ua, i);
                if (a.ub) e = e - ld4(a.ub, i);
                d[0] += (double)__fmul_rn(p0.x, e.x) + (double)__fmul_rn(p0.y, e.y) + (double)__fmul_rn(p0.z, e.z) + (double)__fmul_rn(p0.w, e.w);
            }
        } else {
            const float p0 = a.p[0][i];
            float s = __fmul_rn(a.c[0], p0);
            for (int j = 1; j < a.n; ++j) s = __fadd_rn(s, __fmul_rn(a.c[j], a.p[j][i]));
            if (a.base) s = __fadd_rn(a.base[i], s);
            if (a.mask && !(a.mask[i] > 0.f)) s = 0.f;
            a.out[i] = s;
            if (a.ua) d[0] += (double)__fmul_rn(p0, a.ub ? a.ua[i] - a.ub[i] : a.ua[i]);
        }
    }
    if (a.ua) block_store_dots(d, partial);
}

int rk_pull_f32(float *out, const float *base, const float *const *h_p, const float *h_c, int n_p, const float *mask, const float *ua,
                const float *ub, double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    if (!out || !h_p || !h_c || n_p < 1 || n_p > kBwdMaxK || n < 0 || (ua && (!d_dots || !d_ws))) { set_error("rk pull: bad argument"); return NDCN_EINVAL; }
    PullArgs a;
    a.base = base; a.mask = mask; a.ua = ua; a.ub = ub; a.n = n_p; a.out = out;
    bool vec = n % 4 == 0 && aligned16(out) && (!base || aligned16(base)) && (!mask || aligned16(mask)) && (!ua || aligned16(ua)) && (!ub || aligned16(ub));
    for (int j = 0; j < kBwdMaxK; ++j) {
        a.p[j] = j < n_p ? h_p[j] : h_p[0];
        a.c[j] = j < n_p ? h_c[j] : 0.f;
        if (j < n_p && !h_p[j]) { set_error("rk pull: null term"); return NDCN_EINVAL; }
        if (j < n_p) vec = vec && aligned16(h_p[j]);
    }
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_COMBINE_BWD, st, 4.0 * n * (n_p + 1 + (base ? 1 : 0) + (mask ? 1 : 0) + (ua ? (ub ? 2 : 1) : 0)), 2.0 * n * (n_p + 1));
    const int grid = bwd_grid(vec ? n / 4 : n);
    if (vec) hipLaunchKernelGGL(pull_kernel<true>, dim3(grid), dim3(256), 0, st, a, n / 4, static_cast<double *>(d_ws));
    else hipLaunchKernelGGL(pull_kernel<false>, dim3(grid), dim3(256), 0, st, a, n, static_cast<double *>(d_ws));
    if (ua) hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rk_dot_diff_f32(const float *g, const float *a, const float *b, double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    if (!g || !a || !d_dots || !d_ws || n < 0) { set_error("rk dot_diff: null pointer"); return NDCN_EINVAL; }
    const bool vec = n % 4 == 0 && aligned16(g) && aligned16(a) && (!b || aligned16(b));
    ProfScope prof(PROF_COMBINE_BWD, st, 4.0 * n * (b ? 3 : 2), 2.0 * n);
    const int grid = bwd_grid(vec ? n / 4 : n);
    if (vec) hipLaunchKernelGGL(dot_diff_kernel<true>, dim3(grid), dim3(256), 0, st, g, a, b, n / 4, static_cast<double *>(d_ws));
    else hipLaunchKernelGGL(dot_diff_kernel<false>, dim3(grid), dim3(256), 0, st, g, a, b, n, static_cast<double *>(d_ws));
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rk_error_bwd_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol, float atol,
                     float g_r, double inv_n, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1,
                     const float *const *h_acc, double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    ErrBwdArgs p;
    int rc = fill_bwd(p.t, h_k, h_gk, h_c, n_k, h_acc);
    if (rc) return rc;
    p.y0 = y0; p.y1 = y1; p.gy0 = gy0; p.gy1 = gy1; p.rtol = rtol; p.atol = atol; p.g_r = g_r; p.inv_n = (float)inv_n;
    p.acc_y0 = gy0 ? acc_y0 : nullptr; p.acc_y1 = gy1 ? acc_y1 : nullptr;
    bool vec = n % 4 == 0 && aligned16(y0) && aligned16(y1) && (!gy0 || aligned16(gy0)) && (!gy1 || aligned16(gy1)) &&
               (!p.acc_y0 || aligned16(p.acc_y0)) && (!p.acc_y1 || aligned16(p.acc_y1));
    for (int j = 0; j < n_k; ++j) vec = vec && aligned16(p.t.k[j]) && (!p.t.gk[j] || aligned16(p.t.gk[j])) && (!p.t.acc[j] || aligned16(p.t.acc[j]));
    const int grid = bwd_grid(vec ? n / 4 : n);
    ProfScope prof(PROF_ERROR_BWD, st, 4.0 * n * (2 * n_k + 4), 2.0 * n * (3 * n_k + 12));
    if (vec) hipLaunchKernelGGL(error_bwd_kernel<true>, dim3(grid), dim3(256), 0, st, p, n / 4, static_cast<double *>(d_ws));
    else hipLaunchKernelGGL(error_bwd_kernel<false>, dim3(grid), dim3(256), 0, st, p, n, static_cast<double *>(d_ws));
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rk_rms_bwd_f32(const float *a, const float *b, const float *y, float rtol, float atol, float coef, float *ga, float *gb,
                   float *gy, int64_t n, hipStream_t st) {
    ProfScope prof(PROF_SUMSQ_BWD, st, 4.0 * n * 5, 10.0 * n);
    hipLaunchKernelGGL(rms_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, a, b, y, rtol, atol, coef, ga, gb, gy, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

static const double kCMidBwd[7] = {
    6025192743. / 30085553152. / 2, 0, 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
    187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2,
};

int rk_dense_bwd_f32(const float *g, const float *y0, const float *y1, const float *const *h_k, float dt, float x, float *gy0,
                     float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1, const float *const *h_acc,
                     double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    DenseBwdArgs p;
    p.g = g; p.y0 = y0; p.y1 = y1; p.gy0 = gy0; p.gy1 = gy1; p.dt = dt; p.x = x;
    p.acc_y0 = gy0 ? acc_y0 : nullptr; p.acc_y1 = gy1 ? acc_y1 : nullptr;
    for (int j = 0; j < 7; ++j) {
        if (!h_k[j]) { set_error("dense backward: null stage pointer"); return NDCN_EINVAL; }
        p.k[j] = h_k[j];
        p.gk[j] = h_gk ? h_gk[j] : nullptr;
        p.acc[j] = (p.gk[j] && h_acc) ? h_acc[j] : nullptr;
        p.cmid[j] = (float)kCMidBwd[j];
        p.cm[j] = dt * (float)kCMidBwd[j];
    }
    const int grid = bwd_grid(n, kBwdBlocks);
    ProfScope prof(PROF_DENSE_BWD, st, 4.0 * n * 19, 2.0 * n * 60);
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(grid), dim3(256), 0, st, p, n, static_cast<double *>(d_ws));
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rk_dense_bwd_multi_f32(const float *const *h_g, int nt, const float *y0, const float *y1, const float *const *h_k, float dt,
                           const float *h_x, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1,
                           const float *const *h_acc, double *d_dots, void *d_ws, int64_t n, hipStream_t st) {
    if (nt < 1 || nt > kDenseMulti) { set_error("dense backward: 1..%d ticks per launch", kDenseMulti); return NDCN_EINVAL; }
    DenseMultiArgs p;
    p.nt = nt; p.y0 = y0; p.y1 = y1; p.gy0 = gy0; p.gy1 = gy1; p.dt = dt;
    p.acc_y0 = gy0 ? acc_y0 : nullptr; p.acc_y1 = gy1 ? acc_y1 : nullptr;
    for (int t = 0; t < kDenseMulti; ++t) {
        p.g[t] = h_g[t < nt ? t : 0];
        p.x[t] = t < nt ? h_x[t] : 0.f;
        if (t < nt && !h_g[t]) { set_error("dense backward: null gradient panel"); return NDCN_EINVAL; }
    }
    for (int j = 0; j < 7; ++j) {
        if (!h_k[j]) { set_error("dense backward: null stage pointer"); return NDCN_EINVAL; }
        p.k[j] = h_k[j];
        p.gk[j] = h_gk ? h_gk[j] : nullptr;
        p.acc[j] = (p.gk[j] && h_acc) ? h_acc[j] : nullptr;
        p.cmid[j] = (float)kCMidBwd[j];
        p.cm[j] = dt * (float)kCMidBwd[j];
    }
    const int grid = bwd_grid(n, kBwdBlocks);
    ProfScope prof(PROF_DENSE_BWD, st, 4.0 * n * (27 + nt), 2.0 * n * (30 + 30 * nt));
    hipLaunchKernelGGL(dense_bwd_multi_kernel, dim3(grid), dim3(256), 0, st, p, n, static_cast<double *>(d_ws));
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, st, static_cast<const double *>(d_ws), grid, d_dots);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
