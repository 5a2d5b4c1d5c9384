// Runge-Kutta bookkeeping kernels: every one is a single streaming pass (HBM-bound) that replaces a
// chain of separate ATen elementwise ops in the reference's integrator.
//
// Rounding follows the reference term by term: coefficients arrive already rounded to fp32 as dt*beta
// (misc.py:25), products and sums are rounded separately and added left to right.  FP contraction is
// therefore switched OFF for this translation unit - an fma here would change the last bit relative to
// the reference's separate mul / add kernels.
#include <atomic>
#include <stdlib.h>

#include "common.h"

#pragma clang fp contract(off)

namespace ndcn {

constexpr int kMaxTerms = 8;
constexpr int kRedBlocks = 2048;       // partial sums per reduction (deterministic two-pass)

struct Terms {
    const float *k[kMaxTerms];
    float c[kMaxTerms];
    int n;
    const float *dt_dev = nullptr;   // the step size lives in device memory (hipGraph replay) and c[] holds the bare
                              // tableau entries; the coefficient is then formed here as fl(dt * c) - the same single fp32
                              // rounding the host applies when it passes dt * beta by value
};

__device__ __forceinline__ void apply_dt(Terms &t) {
    if (t.dt_dev) {
        const float dt = *t.dt_dev;
#pragma unroll
        for (int j = 0; j < kMaxTerms; ++j) t.c[j] = dt * t.c[j];
    }
}

__device__ __forceinline__ float4 ld4(const float *p, int64_t i) { return reinterpret_cast<const float4 *>(p)[i]; }
__device__ __forceinline__ void st4(float *p, int64_t i, float4 v) { reinterpret_cast<float4 *>(p)[i] = v; }

// sum_j c_j * k_j[i], left to right, separate roundings  (misc.py:22-25)
__device__ __forceinline__ float wsum1(const Terms &t, int64_t i) {
    float acc = t.c[0] * t.k[0][i];
#pragma unroll
    for (int j = 1; j < kMaxTerms; ++j)
        if (j < t.n) acc = acc + t.c[j] * t.k[j][i];
    return acc;
}
__device__ __forceinline__ float4 wsum4(const Terms &t, int64_t i) {
    float4 k = ld4(t.k[0], i);
    float4 acc = make_float4(t.c[0] * k.x, t.c[0] * k.y, t.c[0] * k.z, t.c[0] * k.w);
#pragma unroll
    for (int j = 1; j < kMaxTerms; ++j)
        if (j < t.n) {
            k = ld4(t.k[j], i);
            const float c = t.c[j];
            acc.x = acc.x + c * k.x; acc.y = acc.y + c * k.y; acc.z = acc.z + c * k.z; acc.w = acc.w + c * k.w;
        }
    return acc;
}

// -------------------------------------------------------------------------------- combine
// out = y0 + sum_j c_j k_j
template <bool VEC>
__global__ __launch_bounds__(256) void combine_kernel(float *__restrict__ out, const float *__restrict__ y0, Terms t,
                                                      int64_t n_items) {
    apply_dt(t);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 s = wsum4(t, i);
            if (!y0) { st4(out, i, s); continue; }              // plain linear combination (no y0 term)
            const float4 y = ld4(y0, i);
            st4(out, i, make_float4(y.x + s.x, y.y + s.y, y.z + s.z, y.w + s.w));
        } else {
            out[i] = y0 ? y0[i] + wsum1(t, i) : wsum1(t, i);
        }
    }
}

// -------------------------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-level sum of two doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b) {
    __shared__ double sa[4], sb[4];
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sa[w] = a; sb[w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = ((sa[0] + sa[1]) + sa[2]) + sa[3];
        b = ((sb[0] + sb[1]) + sb[2]) + sb[3];
    }
}

// Panels up to this many elements are reduced in ATen's float32 order - the order of ONE torch CPU build: torch 2.10.0, x86-64
// AVX2 kernels (8-lane vectors, 4 interleaved accumulators, the cascade sum of SumKernel.cpp), which is what the reference's
// Python executed when the fixtures under tests/golden were captured; an AVX-512 dispatch or another torch version sums in
// another order and these kernels would then match IT no better than any parallel reduction does.  The one-workgroup serial
// form costs ~4 ns per element and lane group (2^18 elements: ~0.13 ms), so it is confined to the sizes where a last-bit
// difference of the mean can flip an accept / reject decision and a reference run exists to compare with: 2^18 elements
// covers every reference-sized solve (the README commands: 400 x 20; Cora 2708 x 64).  NDCN_ATEN_NORM_MAX=<elements> moves
// the bound (0 or NDCN_ATEN_NORM=0: parallel fp64 reductions everywhere).
constexpr int64_t kAtenNormMaxDefault = 1ll << 18;
constexpr int kAtenBlock = 2048;

__device__ __forceinline__ bool nonfinite(float v) { return !(fabsf(v) <= 3.402823466e38f); }

__device__ __forceinline__ float err_ratio_sq(float e, float a, float b, float rtol, float atol) {
    // misc.py:151-156: tol = atol + rtol * max(|y0|, |y1|); r = err / tol; r * r
    const float tol = atol + rtol * max_nan(fabsf(a), fabsf(b));
    const float r = e / tol;
    return r * r;
}

// The mean squared error ratio in the order ATen's float32 `sum` forms it (torch.mean = sum / n on the CPU; SumKernel.cpp's
// cascade sum, matched bit for bit against torch on random vectors like the norm above - tools/micro/aten_norm_order.py):
//   the row is read as vectors of 8 lanes, 4 vectors interleaved ("ilp"): 32 independent running sums, sum (k, w) owning
//   elements 32 i + 8 k + w; they are kept in 4 cascade LEVELS: level 0 takes the elements, after every `step` = 2^p of them
//   (p = max(4, ceil_log2(n / 32) / 4)) it is added into level 1 and cleared, level 1 into level 2 after step^2 ... ; at the
//   end the levels are added up (1, 2, 3 into 0), then the left-over vectors into ilp slot 0, the ilp slots 1..3 into slot
//   0, and finally: the n % 8 tail elements, then the 8 lanes, left to right, starting from 0.
// dopri5's accept / reject decision compares that mean with 1; at rtol 1e-7 the ratios sit close enough to 1 for the
// last bit to matter, so panels up to aten_order_max_elems() elements are reduced in exactly this order (one workgroup: all threads
// form r^2 of a 2048-element chunk in LDS, 32 lanes of the first wave run the cascade); larger panels keep the parallel
// fp64 reduction, as do the error records formed inside the fused right-hand sides.
// Round 5: 1024 threads.  Waves 1 .. 15 form r^2 of chunk c + 1 into the other half of a double buffer while 32 lanes of wave 0 run the
// cascade over chunk c, whose LDS reads are requested 16 steps at a time ahead of the (serial, order-defining) additions: the
// 8000-element record of the README-sized solves 66 -> ~15 us under rocprofv3 (it was longer than the step's six evaluations).
constexpr int kAtenThreads = 1024;
__global__ __launch_bounds__(kAtenThreads) void rk_error_aten_kernel(const float *__restrict__ y0, const float *__restrict__ y1, Terms t,
                                                                     float rtol, float atol, int64_t n, double *__restrict__ out, int accum) {
    apply_dt(t);
    __shared__ float v2[2][kAtenBlock];
    __shared__ float lane32[32];
    __shared__ int bad_cnt;
    const int tid = threadIdx.x;
    if (tid == 0) bad_cnt = 0;
    int bad = 0;
    const int64_t nv = n / 8, size_ilp = nv / 4;
    int lg = 1;                                                   // CeilLog2 of ATen: 1 for x <= 2
    if (size_ilp > 2) { lg = 0; while ((1ll << lg) < size_ilp) ++lg; }
    const int level_power = (lg / 4) > 4 ? (lg / 4) : 4;
    const int64_t level_step = 1ll << level_power, level_mask = level_step - 1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t i = 0;                                                // cascade position in steps of 32 elements
    int64_t in_level = 0;                                         // steps taken inside the current level-0 run
    const int64_t n_main = size_ilp * 32;
    auto ratio_sq = [&](int64_t e) {
        const float b = y1[e];
        bad += (int)nonfinite(b);
        return err_ratio_sq(wsum1(t, e), y0[e], b, rtol, atol);
    };
    auto produce = [&](int64_t base, float *dst, int first, int stride) {
        const int cnt = (int)((n_main - base) < kAtenBlock ? (n_main - base) : kAtenBlock);
        for (int q = first; q < cnt; q += stride) dst[q] = ratio_sq(base + q);
    };
    if (n_main > 0) produce(0, v2[0], tid, kAtenThreads);
    __syncthreads();
    int buf = 0;
    for (int64_t base = 0; base < n_main; base += kAtenBlock, buf ^= 1) {
        const int cnt = (int)((n_main - base) < kAtenBlock ? (n_main - base) : kAtenBlock);
        if (tid >= 64) {
            if (base + kAtenBlock < n_main) produce(base + kAtenBlock, v2[buf ^ 1], tid - 64, kAtenThreads - 64);
        } else if (tid < 32) {
            const float *v = v2[buf];
            for (int q0 = tid; q0 < cnt; q0 += 32 * 16) {
                float x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) x[u] = (q0 + 32 * u < cnt) ? v[q0 + 32 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (q0 + 32 * u >= cnt) break;
                    // whole runs of level_step steps cascade upwards; a trailing partial run stays in level 0
                    a0 = a0 + x[u];
                    ++i; ++in_level;
                    if (in_level == level_step && i <= (size_ilp / level_step) * level_step) {
                        in_level = 0;
                        a1 = a1 + a0; a0 = 0.f;
                        if ((i & (level_mask << level_power)) == 0) {
                            a2 = a2 + a1; a1 = 0.f;
                            if ((i & (level_mask << (2 * level_power))) == 0) { a3 = a3 + a2; a2 = 0.f; }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    float *v = v2[0];
    __syncthreads();
    if (tid < 32) {
        a0 = a0 + a1; a0 = a0 + a2; a0 = a0 + a3;
        lane32[tid] = a0;
    }
    // the left-over elements (fewer than 32 + 8): r^2 into LDS, then the serial finish by one thread
    const int n_left = (int)(n - n_main);
    __syncthreads();
    for (int q = tid; q < n_left; q += kAtenThreads) v[q] = ratio_sq(n_main + q);
    if (bad) atomicAdd(&bad_cnt, bad);
    __syncthreads();
    if (tid == 0) {
        float p[8];
        const int left_vecs = (int)(nv - size_ilp * 4);           // whole 8-lane vectors after the interleaved part: into ilp slot 0
        for (int w = 0; w < 8; ++w) {
            float s0 = lane32[w];
            for (int u = 0; u < left_vecs; ++u) s0 = s0 + v[8 * u + w];
            p[w] = ((s0 + lane32[8 + w]) + lane32[16 + w]) + lane32[24 + w];
        }
        float s = 0.f;
        for (int q = 8 * left_vecs; q < n_left; ++q) s = s + v[q];  // the n % 8 tail first
        for (int w = 0; w < 8; ++w) s = s + p[w];
        out[0] = accum ? out[0] + (double)s : (double)s;
        out[1] = accum ? out[1] + (double)bad_cnt : (double)bad_cnt;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void rk_error_kernel(const float *__restrict__ y0, const float *__restrict__ y1,
                                                       Terms t, float rtol, float atol, int64_t n_items,
                                                       double *__restrict__ partial) {
    apply_dt(t);
    double s = 0.0, bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 e = wsum4(t, i);
            const float4 a = ld4(y0, i), b = ld4(y1, i);
            s += (double)err_ratio_sq(e.x, a.x, b.x, rtol, atol);
            s += (double)err_ratio_sq(e.y, a.y, b.y, rtol, atol);
            s += (double)err_ratio_sq(e.z, a.z, b.z, rtol, atol);
            s += (double)err_ratio_sq(e.w, a.w, b.w, rtol, atol);
            bad += (double)((int)nonfinite(b.x) + (int)nonfinite(b.y) + (int)nonfinite(b.z) + (int)nonfinite(b.w));
        } else {
            const float b = y1[i];
            s += (double)err_ratio_sq(wsum1(t, i), y0[i], b, rtol, atol);
            bad += (double)(int)nonfinite(b);
        }
    }
    block_sum2(s, bad);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = bad; }
}

__device__ __forceinline__ float scaled_sq(float a, float b, float y, float rtol, float atol) {
    // misc.py:121-138: scale = atol + |y0| * rtol ; ((a - b) / scale)^2
    const float scale = atol + fabsf(y) * rtol;
    const float q = (a - b) / scale;
    return q * q;
}

// The same sum in the order ATen's float32 `norm` forms it (torch 2.x CPU, the build the fixtures were captured with;
// found by matching torch's result bit for bit on 60 random vectors, tools/micro/aten_norm_order.py): EIGHT running sums -
// lane j owns elements j, j + 8, j + 16, ... and accumulates acc_j = fma(q, q, acc_j) in index order - added up left to
// right, then the n % 8 tail elements with fma.  The reference's initial step (misc.py:121-138) takes three such norms; at
// rtol 1e-7 a 1-ulp difference there reshuffles later accept / reject decisions (the error estimate is then a cancellation
// of O(1e-9) terms), so for panels up to aten_order_max_elems() elements the sum is formed in exactly that order: one workgroup,
// q = (a - b) / scale computed by all threads into LDS, then 8 lanes walk their chains.  Larger panels keep the parallel
// fp64 reduction (8 sequential chains over 10^8 elements would take tens of milliseconds per norm).
template <bool HASB>
__global__ __launch_bounds__(kAtenThreads) void scaled_sumsq_aten_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                         const float *__restrict__ y, float rtol, float atol, int64_t n,
                                                                         double *__restrict__ out) {
    __shared__ float q2[2][kAtenBlock];
    __shared__ float lane_sum[8];
    __shared__ int bad_cnt;
    const int tid = threadIdx.x;
    if (tid == 0) bad_cnt = 0;
    float acc = 0.f;
    int bad = 0;
    const int64_t n8 = n - (n % 8);
    // (as rk_error_aten_kernel: waves 1 .. 15 form the quotients of the next chunk while 8 lanes of wave 0 walk their chains over this one)
    auto produce = [&](int64_t base, float *dst, int first, int stride) {
        const int cnt = (int)((n8 - base) < kAtenBlock ? (n8 - base) : kAtenBlock);
        for (int i = first; i < cnt; i += stride) {
            const float av = a[base + i];
            const float scale = atol + fabsf(y[base + i]) * rtol;
            dst[i] = HASB ? (av - b[base + i]) / scale : av / scale;
            bad += (int)nonfinite(av);
        }
    };
    if (n8 > 0) produce(0, q2[0], tid, kAtenThreads);
    __syncthreads();
    int buf = 0;
    for (int64_t base = 0; base < n8; base += kAtenBlock, buf ^= 1) {
        const int cnt = (int)((n8 - base) < kAtenBlock ? (n8 - base) : kAtenBlock);
        if (tid >= 64) {
            if (base + kAtenBlock < n8) produce(base + kAtenBlock, q2[buf ^ 1], tid - 64, kAtenThreads - 64);
        } else if (tid < 8) {
            const float *q = q2[buf];
            for (int i0 = tid; i0 < cnt; i0 += 8 * 16) {
                float x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) x[u] = (i0 + 8 * u < cnt) ? q[i0 + 8 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (i0 + 8 * u >= cnt) break;
                    acc = fmaf(x[u], x[u], acc);
                }
            }
        }
        __syncthreads();
    }
    if (tid < 8) lane_sum[tid] = acc;
    if (bad) atomicAdd(&bad_cnt, bad);
    __syncthreads();
    if (tid == 0) {
        float s = lane_sum[0];
        for (int j = 1; j < 8; ++j) s = s + lane_sum[j];
        int tb = bad_cnt;
        for (int64_t i = n8; i < n; ++i) {
            const float av = a[i];
            const float scale = atol + fabsf(y[i]) * rtol;
            const float qq = HASB ? (av - b[i]) / scale : av / scale;
            s = fmaf(qq, qq, s);
            tb += (int)nonfinite(av);
        }
        out[0] = (double)s;
        out[1] = (double)tb;
    }
}

template <bool VEC, bool HASB>
__global__ __launch_bounds__(256) void scaled_sumsq_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           const float *__restrict__ y, float rtol, float atol,
                                                           int64_t n_items, double *__restrict__ partial) {
    double s = 0.0, bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 av = ld4(a, i), yv = ld4(y, i);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HASB) bv = ld4(b, i);
            if (HASB) {
                s += (double)scaled_sq(av.x, bv.x, yv.x, rtol, atol);
                s += (double)scaled_sq(av.y, bv.y, yv.y, rtol, atol);
                s += (double)scaled_sq(av.z, bv.z, yv.z, rtol, atol);
                s += (double)scaled_sq(av.w, bv.w, yv.w, rtol, atol);
            } else {
                // no subtraction at all when b is absent: (a / scale)^2 as misc.py:123-124
                float q;
                q = av.x / (atol + fabsf(yv.x) * rtol); s += (double)(q * q);
                q = av.y / (atol + fabsf(yv.y) * rtol); s += (double)(q * q);
                q = av.z / (atol + fabsf(yv.z) * rtol); s += (double)(q * q);
                q = av.w / (atol + fabsf(yv.w) * rtol); s += (double)(q * q);
            }
            bad += (double)((int)nonfinite(av.x) + (int)nonfinite(av.y) + (int)nonfinite(av.z) + (int)nonfinite(av.w));
        } else {
            const float av = a[i];
            if (HASB) {
                s += (double)scaled_sq(av, b[i], y[i], rtol, atol);
            } else {
                const float q = av / (atol + fabsf(y[i]) * rtol);
                s += (double)(q * q);
            }
            bad += (double)(int)nonfinite(av);
        }
    }
    block_sum2(s, bad);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = bad; }
}

// d0 and d1 of the initial step (misc.py:121-127) in ONE pass over {y0, f0}: sum (y0 / scale)^2 and sum (f0 / scale)^2 with
// scale = atol + |y0| rtol, element for element the arithmetic of scaled_sumsq_kernel<., false>; two partial arrays
template <bool VEC>
__global__ __launch_bounds__(256) void scaled_sumsq_pair_kernel(const float *__restrict__ f, const float *__restrict__ y,
                                                                float rtol, float atol, int64_t n_items,
                                                                double *__restrict__ partial_y, double *__restrict__ partial_f) {
    double sy = 0.0, by = 0.0, sf = 0.0, bf = 0.0;
    auto one = [&](float fv, float yv) {
        const float scale = atol + fabsf(yv) * rtol;
        const float qy = yv / scale, qf = fv / scale;
        sy += (double)(qy * qy);
        sf += (double)(qf * qf);
        by += (double)(int)nonfinite(yv);
        bf += (double)(int)nonfinite(fv);
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 fv = ld4(f, i), yv = ld4(y, i);
            one(fv.x, yv.x); one(fv.y, yv.y); one(fv.z, yv.z); one(fv.w, yv.w);
        } else {
            one(f[i], y[i]);
        }
    }
    block_sum2(sy, by);
    if (threadIdx.x == 0) { partial_y[2 * blockIdx.x] = sy; partial_y[2 * blockIdx.x + 1] = by; }
    __syncthreads();
    block_sum2(sf, bf);
    if (threadIdx.x == 0) { partial_f[2 * blockIdx.x] = sf; partial_f[2 * blockIdx.x + 1] = bf; }
}

// fixed-order final sum of the per-block partials
__global__ __launch_bounds__(256) void reduce_finish_kernel(const double *__restrict__ partial, int n_partial,
                                                            double *__restrict__ out, int accum) {
    double s = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) { s += partial[2 * i]; bad += partial[2 * i + 1]; }
    block_sum2(s, bad);
    if (threadIdx.x == 0) { out[0] = accum ? out[0] + s : s; out[1] = accum ? out[1] + bad : bad; }
}

// -------------------------------------------------------------------------------- dense output
struct FitArgs {
    const float *y0, *y1;
    Terms mid;              // dt * DPS_C_MID terms (zero coefficients already dropped by the caller)
    const float *f0, *f1;
    float dt;
    float *a, *b, *c, *d;
};

__device__ __forceinline__ void fit1(float y0, float y1, float ms, float f0, float f1, float dt, float &a, float &b,
                                     float &c, float &d) {
    const float ym = y0 + ms;                                                  // dopri5.py:42
    // interp.py:21-35 -- `_dot_product` sums c*x products left to right; -2*dt etc. are fp32 scalars
    a = ((((-2.f * dt) * f0 + (2.f * dt) * f1) + -8.f * y0) + -8.f * y1) + 16.f * ym;
    b = ((((5.f * dt) * f0 + (-3.f * dt) * f1) + 18.f * y0) + 14.f * y1) + -32.f * ym;
    c = ((((-4.f * dt) * f0 + dt * f1) + -11.f * y0) + -5.f * y1) + 16.f * ym;
    d = dt * f0;
}

template <bool VEC>
__global__ __launch_bounds__(256) void interp_fit_kernel(FitArgs p, int64_t n_items) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 ms = wsum4(p.mid, i);
            const float4 y0 = ld4(p.y0, i), y1 = ld4(p.y1, i), f0 = ld4(p.f0, i), f1 = ld4(p.f1, i);
            float4 a, b, c, d;
            fit1(y0.x, y1.x, ms.x, f0.x, f1.x, p.dt, a.x, b.x, c.x, d.x);
            fit1(y0.y, y1.y, ms.y, f0.y, f1.y, p.dt, a.y, b.y, c.y, d.y);
            fit1(y0.z, y1.z, ms.z, f0.z, f1.z, p.dt, a.z, b.z, c.z, d.z);
            fit1(y0.w, y1.w, ms.w, f0.w, f1.w, p.dt, a.w, b.w, c.w, d.w);
            st4(p.a, i, a); st4(p.b, i, b); st4(p.c, i, c); st4(p.d, i, d);
        } else {
            float a, b, c, d;
            fit1(p.y0[i], p.y1[i], wsum1(p.mid, i), p.f0[i], p.f1[i], p.dt, a, b, c, d);
            p.a[i] = a; p.b[i] = b; p.c[i] = c; p.d[i] = d;
        }
    }
}

struct EvalArgs {
    const float *a, *b, *c, *d, *e;
    float x4, x3, x2, x1, x0;
    float *out;
};

__device__ __forceinline__ float eval1(float a, float b, float c, float d, float e, const EvalArgs &p) {
    // interp.py:65: a*x^4 + b*x^3 + c*x^2 + d*x + e*1, left to right
    return (((a * p.x4 + b * p.x3) + c * p.x2) + d * p.x1) + e * p.x0;
}

template <bool VEC>
__global__ __launch_bounds__(256) void interp_eval_kernel(EvalArgs p, int64_t n_items) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 a = ld4(p.a, i), b = ld4(p.b, i), c = ld4(p.c, i), d = ld4(p.d, i), e = ld4(p.e, i);
            st4(p.out, i, make_float4(eval1(a.x, b.x, c.x, d.x, e.x, p), eval1(a.y, b.y, c.y, d.y, e.y, p),
                                      eval1(a.z, b.z, c.z, d.z, e.z, p), eval1(a.w, b.w, c.w, d.w, e.w, p)));
        } else {
            p.out[i] = eval1(p.a[i], p.b[i], p.c[i], p.d[i], p.e[i], p);
        }
    }
}

// fit + evaluate in one pass, nothing stored: for the (common) step that is sampled at a single tick
struct DirectArgs {
    FitArgs f;                      // a, b, c, d unused
    float x4, x3, x2, x1, x0;
    float *out;
};

__device__ __forceinline__ float direct1(float y0, float y1, float ms, float f0, float f1, const DirectArgs &p) {
    float a, b, c, d;
    fit1(y0, y1, ms, f0, f1, p.f.dt, a, b, c, d);
    return (((a * p.x4 + b * p.x3) + c * p.x2) + d * p.x1) + y0 * p.x0;     // interp.py:65, e = y0
}

template <bool VEC>
__global__ __launch_bounds__(256) void interp_direct_kernel(DirectArgs p, int64_t n_items) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 ms = wsum4(p.f.mid, i);
            const float4 y0 = ld4(p.f.y0, i), y1 = ld4(p.f.y1, i), f0 = ld4(p.f.f0, i), f1 = ld4(p.f.f1, i);
            st4(p.out, i, make_float4(direct1(y0.x, y1.x, ms.x, f0.x, f1.x, p), direct1(y0.y, y1.y, ms.y, f0.y, f1.y, p),
                                      direct1(y0.z, y1.z, ms.z, f0.z, f1.z, p), direct1(y0.w, y1.w, ms.w, f0.w, f1.w, p)));
        } else {
            p.out[i] = direct1(p.f.y0[i], p.f.y1[i], wsum1(p.f.mid, i), p.f.f0[i], p.f.f1[i], p);
        }
    }
}

// the same for up to kMaxTicks ticks that fall into ONE accepted step: the step's panels are read once, one output
// panel is written per tick ((2 + m) P + nt P of traffic instead of nt (3 + m) P)
constexpr int kMaxTicks = 8;
struct DirectMultiArgs {
    FitArgs f;                      // a, b, c, d unused
    float xp[kMaxTicks][5];         // {x^4, x^3, x^2, x, 1} per tick
    float *out[kMaxTicks];
    int nt;
};

__device__ __forceinline__ float poly1(float a, float b, float c, float d, float y0, const float *xp) {
    return (((a * xp[0] + b * xp[1]) + c * xp[2]) + d * xp[3]) + y0 * xp[4];      // interp.py:65, e = y0
}

template <bool VEC>
__global__ __launch_bounds__(256) void interp_direct_multi_kernel(DirectMultiArgs p, int64_t n_items) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 ms = wsum4(p.f.mid, i);
            const float4 y0 = ld4(p.f.y0, i), y1 = ld4(p.f.y1, i), f0 = ld4(p.f.f0, i), f1 = ld4(p.f.f1, i);
            float4 a, b, c, d;
            fit1(y0.x, y1.x, ms.x, f0.x, f1.x, p.f.dt, a.x, b.x, c.x, d.x);
            fit1(y0.y, y1.y, ms.y, f0.y, f1.y, p.f.dt, a.y, b.y, c.y, d.y);
            fit1(y0.z, y1.z, ms.z, f0.z, f1.z, p.f.dt, a.z, b.z, c.z, d.z);
            fit1(y0.w, y1.w, ms.w, f0.w, f1.w, p.f.dt, a.w, b.w, c.w, d.w);
#pragma unroll
            for (int t = 0; t < kMaxTicks; ++t)
                if (t < p.nt)
                    st4(p.out[t], i, make_float4(poly1(a.x, b.x, c.x, d.x, y0.x, p.xp[t]), poly1(a.y, b.y, c.y, d.y, y0.y, p.xp[t]),
                                                 poly1(a.z, b.z, c.z, d.z, y0.z, p.xp[t]), poly1(a.w, b.w, c.w, d.w, y0.w, p.xp[t])));
        } else {
            const float y0 = p.f.y0[i];
            float a, b, c, d;
            fit1(y0, p.f.y1[i], wsum1(p.f.mid, i), p.f.f0[i], p.f.f1[i], p.f.dt, a, b, c, d);
#pragma unroll
            for (int t = 0; t < kMaxTicks; ++t)
                if (t < p.nt) p.out[t][i] = poly1(a, b, c, d, y0, p.xp[t]);
        }
    }
}

// -------------------------------------------------------------------------------- fixed-grid stages
template <int OP>
__device__ __forceinline__ float stage1(float y, float k1, float k2, float k3, float k4, float dt) {
    if (OP == 0) return y + dt * k1;                                   // fixed_grid.py:8 + solvers.py:92
    if (OP == 1) return y + k1 * dt / 2.f;                             // fixed_grid.py:18
    if (OP == 2) return y + dt * k1 / 3.f;                             // rk_common.py:75
    if (OP == 3) return y + dt * (k1 / -3.f + k2);                     // rk_common.py:76
    if (OP == 4) return y + dt * (k1 - k2 + k3);                       // rk_common.py:77
    return y + (k1 + 3.f * k2 + 3.f * k3 + k4) * (dt / 8.f);           // rk_common.py:78 + solvers.py:92
}

template <int OP, bool VEC>
__global__ __launch_bounds__(256) void fixed_stage_kernel(float *out, const float *y, const float *k1, const float *k2,
                                                          const float *k3, const float *k4, float dt_val,
                                                          const float *__restrict__ dt_dev, int64_t n_items) {
    // dt_dev != NULL: the step size lives in device memory, so ONE captured hipGraph serves every step of an
    // irregular time grid (solvers.py:51: the grid is the caller's t, each interval has its own dt)
    const float dt = dt_dev ? *dt_dev : dt_val;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (VEC) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 yv = ld4(y, i), a = ld4(k1, i);
            const float4 b = (OP >= 3) ? ld4(k2, i) : z;
            const float4 c = (OP >= 4) ? ld4(k3, i) : z;
            const float4 d = (OP >= 5) ? ld4(k4, i) : z;
            st4(out, i, make_float4(stage1<OP>(yv.x, a.x, b.x, c.x, d.x, dt), stage1<OP>(yv.y, a.y, b.y, c.y, d.y, dt),
                                    stage1<OP>(yv.z, a.z, b.z, c.z, d.z, dt), stage1<OP>(yv.w, a.w, b.w, c.w, d.w, dt)));
        } else {
            out[i] = stage1<OP>(y[i], k1[i], OP >= 3 ? k2[i] : 0.f, OP >= 4 ? k3[i] : 0.f, OP >= 5 ? k4[i] : 0.f, dt);
        }
    }
}

__global__ void scale_coef_kernel(float *__restrict__ out, const float *__restrict__ beta, const float *__restrict__ dt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dt[0] * beta[i];
}

// out = w * x  (the VJP of one term of a Runge-Kutta linear combination)
__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ out, const float *__restrict__ x, float w, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ld4(x, i);
        st4(out, i, make_float4(w * v.x, w * v.y, w * v.z, w * v.w));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = w * x[i];
}

// dst = src, 16 bytes per lane and 4 independent accesses in flight per lane: the streaming rate the box sustains for a
// plain panel pass (bench.py reports every kernel's HBM traffic against it next to the 8 TB/s spec peak)
typedef float cp_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n4, int64_t n) {
    const cp_f32x4 *s = reinterpret_cast<const cp_f32x4 *>(src);
    cp_f32x4 *d = reinterpret_cast<cp_f32x4 *>(dst);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) dst[j] = src[j];
}

int copy_f32(float *dst, const float *src, int64_t n, hipStream_t st) {
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_STAGE, st, 8.0 * n, 0.0);
    const int64_t n4 = (aligned16(dst) && aligned16(src)) ? n / 4 : 0;
    hipLaunchKernelGGL(copy_kernel, dim3(stream_grid_full(n4 ? n4 : n, 256)), dim3(256), 0, st, dst, src, n4, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// out = g where y > 0, else 0  (VJP of relu given its output; neural_dynamics.py:36)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float *__restrict__ out, const float *__restrict__ g, const float *__restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] > 0.f ? g[i] : 0.f;
}

int relu_bwd_f32(float *out, const float *g, const float *y, int64_t n, hipStream_t st) {
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_RELU_BWD, st, 12.0 * n, 0.0);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid_full(n, 256)), dim3(256), 0, st, out, g, y, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int scale_f32(float *out, const float *x, float w, int64_t n, hipStream_t st) {
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_STAGE, st, 8.0 * n, 1.0 * n);
    // views at odd element offsets (torch.stack's backward hands out slices of one buffer): scalar path, like the
    // other panel kernels of this file
    const int64_t n4 = (aligned16(out) && aligned16(x)) ? n / 4 : 0;
    hipLaunchKernelGGL(scale_kernel, dim3(stream_grid_full((n4 ? n4 : n) + 1, 256)), dim3(256), 0, st, out, x, w, n4, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// -------------------------------------------------------------------------------- host wrappers
int scale_coef_f32(float *out, const float *beta, const float *dt, int n, hipStream_t st) {
    if (n <= 0) return NDCN_OK;
    hipLaunchKernelGGL(scale_coef_kernel, dim3((n + 63) / 64), dim3(64), 0, st, out, beta, dt, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

static bool fill_terms(Terms &t, const float *const *h_k, const float *h_c, int n_k, bool &vec) {
    if (n_k < 1 || n_k > kMaxTerms) return false;
    t.n = n_k;
    for (int j = 0; j < kMaxTerms; ++j) {
        t.k[j] = j < n_k ? h_k[j] : h_k[0];
        t.c[j] = j < n_k ? h_c[j] : 0.f;
        if (j < n_k) {
            if (!h_k[j]) return false;
            vec = vec && aligned16(h_k[j]);
        }
    }
    return true;
}

int rk_combine_f32(float *out, const float *y0, const float *const *h_k, const float *h_c, int n_k, int64_t n,
                   hipStream_t st, const float *dt_dev) {
    Terms t;
    t.dt_dev = dt_dev;
    bool vec = (n % 4 == 0) && aligned16(out) && (!y0 || aligned16(y0));
    if (!fill_terms(t, h_k, h_c, n_k, vec)) { set_error("rk_combine: need 1..%d non-null terms", kMaxTerms); return NDCN_EINVAL; }
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_COMBINE, st, 4.0 * n * (n_k + 2), 2.0 * n * n_k);
    if (vec) hipLaunchKernelGGL((combine_kernel<true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, out, y0, t, n / 4);
    else hipLaunchKernelGGL((combine_kernel<false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, out, y0, t, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// run-time override (ndcn_set_aten_norm_max): < 0 = none, the environment's bound applies
static std::atomic<int64_t> g_aten_override{-1};
int64_t set_aten_order_max_elems(int64_t v) { return g_aten_override.exchange(v < 0 ? (int64_t)-1 : (v > ((int64_t)1 << 24) ? ((int64_t)1 << 24) : v)); }

int64_t aten_order_max_elems() {
    const int64_t ov = g_aten_override.load(std::memory_order_relaxed);
    if (ov >= 0) return ov;
    static const int64_t bound = []() -> int64_t {
        const char *e = getenv("NDCN_ATEN_NORM");
        if (e && e[0] == '0') return (int64_t)0;
        const char *m = getenv("NDCN_ATEN_NORM_MAX");
        const int64_t v = m ? atoll(m) : kAtenNormMaxDefault;
        const int64_t cap = (int64_t)1 << 24;
        return v < 0 ? (int64_t)0 : (v > cap ? cap : v);
    }();
    return bound;
}

int64_t rhs_fused2_partials_bytes();
int64_t reduce_ws_bytes() {
    const int64_t a = (int64_t)kRedBlocks * 2 * sizeof(double), b = rhs_fused2_partials_bytes();
    const int64_t c = (int64_t)kCus * 4 * 4 * 2 * sizeof(double);       // row SpMM with the ERROR epilogue: one slot per wave
    const int64_t m = a > b ? a : b;
    return m > c ? m : c;      // one scratch serves the reductions of rk.hip and of the RHS epilogues
}

static int red_grid(int64_t items) {
    int g = stream_grid(items, 256);
    return g > kRedBlocks ? kRedBlocks : g;
}

int rk_error_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol,
                 float atol, int64_t n, double *d_out, void *d_ws, hipStream_t st, const float *dt_dev, int accum) {
    Terms t;
    t.dt_dev = dt_dev;
    bool vec = (n % 4 == 0) && aligned16(y0) && aligned16(y1);
    if (!fill_terms(t, h_k, h_c, n_k, vec)) { set_error("rk_error: need 1..%d non-null terms", kMaxTerms); return NDCN_EINVAL; }
    if (n >= 8 && n <= aten_order_max_elems()) {
        ProfScope prof(PROF_ERROR, st, 4.0 * n * (n_k + 2), 2.0 * n * (n_k + 4));
        hipLaunchKernelGGL(rk_error_aten_kernel, dim3(1), dim3(kAtenThreads), 0, st, y0, y1, t, rtol, atol, n, d_out, accum);
        NDCN_LAUNCH_CHECK();
        return NDCN_OK;
    }
    double *partial = static_cast<double *>(d_ws);
    const int64_t items = vec ? n / 4 : n;
    const int g = red_grid(items);
    ProfScope prof(PROF_ERROR, st, 4.0 * n * (n_k + 2), 2.0 * n * (n_k + 4));
    if (vec) hipLaunchKernelGGL((rk_error_kernel<true>), dim3(g), dim3(256), 0, st, y0, y1, t, rtol, atol, items, partial);
    else hipLaunchKernelGGL((rk_error_kernel<false>), dim3(g), dim3(256), 0, st, y0, y1, t, rtol, atol, items, partial);
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, st, partial, g, d_out, accum);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int scaled_sumsq_f32(const float *a, const float *b, const float *y, float rtol, float atol, int64_t n, double *d_out,
                     void *d_ws, hipStream_t st) {
    if (n > 0 && n <= aten_order_max_elems()) {
        ProfScope prof(PROF_SUMSQ, st, 4.0 * n * (b ? 3 : 2), 6.0 * n);
        if (b) hipLaunchKernelGGL(scaled_sumsq_aten_kernel<true>, dim3(1), dim3(kAtenThreads), 0, st, a, b, y, rtol, atol, n, d_out);
        else hipLaunchKernelGGL(scaled_sumsq_aten_kernel<false>, dim3(1), dim3(kAtenThreads), 0, st, a, b, y, rtol, atol, n, d_out);
        NDCN_LAUNCH_CHECK();
        return NDCN_OK;
    }
    const bool vec = (n % 4 == 0) && aligned16(a) && aligned16(y) && (!b || aligned16(b));
    double *partial = static_cast<double *>(d_ws);
    const int64_t items = vec ? n / 4 : n;
    const int g = red_grid(items);
    ProfScope prof(PROF_SUMSQ, st, 4.0 * n * (b ? 3 : 2), 6.0 * n);
#define NDCN_SS(V, B) hipLaunchKernelGGL((scaled_sumsq_kernel<V, B>), dim3(g), dim3(256), 0, st, a, b, y, rtol, atol, items, partial)
    if (vec) { if (b) NDCN_SS(true, true); else NDCN_SS(true, false); }
    else { if (b) NDCN_SS(false, true); else NDCN_SS(false, false); }
#undef NDCN_SS
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, st, partial, g, d_out, 0);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// {sum (y / scale)^2, non-finite y, sum (f / scale)^2, non-finite f} -> d_out4; d_ws, d_ws2: two reduction workspaces
int scaled_sumsq_pair_f32(const float *f, const float *y, float rtol, float atol, int64_t n, double *d_out4, void *d_ws,
                          void *d_ws2, hipStream_t st) {
    const bool vec = (n % 4 == 0) && aligned16(f) && aligned16(y);
    const int64_t items = vec ? n / 4 : n;
    const int g = red_grid(items);
    double *py = static_cast<double *>(d_ws), *pf = static_cast<double *>(d_ws2);
    ProfScope prof(PROF_SUMSQ, st, 4.0 * n * 2, 12.0 * n);
    if (vec) hipLaunchKernelGGL((scaled_sumsq_pair_kernel<true>), dim3(g), dim3(256), 0, st, f, y, rtol, atol, items, py, pf);
    else hipLaunchKernelGGL((scaled_sumsq_pair_kernel<false>), dim3(g), dim3(256), 0, st, f, y, rtol, atol, items, py, pf);
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, st, py, g, d_out4, 0);
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, st, pf, g, d_out4 + 2, 0);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int interp_fit_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt, float *a,
                   float *b, float *c, float *d, int64_t n, hipStream_t st) {
    FitArgs p;
    bool vec = (n % 4 == 0) && aligned16(y0) && aligned16(y1) && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d);
    // drop zero coefficients (c_mid[1] == 0): the product would be an exact zero
    const float *kk[kMaxTerms];
    float cc[kMaxTerms];
    int m = 0;
    for (int j = 0; j < 7; ++j) {
        if (!h_k[j]) { set_error("interp_fit: null stage pointer"); return NDCN_EINVAL; }
        vec = vec && aligned16(h_k[j]);
        if (h_cmid[j] != 0.f) { kk[m] = h_k[j]; cc[m] = h_cmid[j]; ++m; }
    }
    if (m == 0) { kk[0] = h_k[0]; cc[0] = 0.f; m = 1; }
    bool dummy = true;
    fill_terms(p.mid, kk, cc, m, dummy);
    p.y0 = y0; p.y1 = y1; p.f0 = h_k[0]; p.f1 = h_k[6]; p.dt = dt; p.a = a; p.b = b; p.c = c; p.d = d;
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_FIT, st, 4.0 * n * (2 + m + 4), 2.0 * n * (m + 16));
    if (vec) hipLaunchKernelGGL((interp_fit_kernel<true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, p, n / 4);
    else hipLaunchKernelGGL((interp_fit_kernel<false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, p, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int interp_direct_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                      const float xp[5], float *out, int64_t n, hipStream_t st) {
    DirectArgs p;
    bool vec = (n % 4 == 0) && aligned16(y0) && aligned16(y1) && aligned16(out);
    const float *kk[kMaxTerms];
    float cc[kMaxTerms];
    int m = 0;
    for (int j = 0; j < 7; ++j) {
        if (!h_k[j]) { set_error("interp_direct: null stage pointer"); return NDCN_EINVAL; }
        vec = vec && aligned16(h_k[j]);
        if (h_cmid[j] != 0.f) { kk[m] = h_k[j]; cc[m] = h_cmid[j]; ++m; }
    }
    if (m == 0) { kk[0] = h_k[0]; cc[0] = 0.f; m = 1; }
    bool dummy = true;
    fill_terms(p.f.mid, kk, cc, m, dummy);
    p.f.y0 = y0; p.f.y1 = y1; p.f.f0 = h_k[0]; p.f.f1 = h_k[6]; p.f.dt = dt;
    p.f.a = p.f.b = p.f.c = p.f.d = nullptr;
    p.x4 = xp[0]; p.x3 = xp[1]; p.x2 = xp[2]; p.x1 = xp[3]; p.x0 = xp[4];
    p.out = out;
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_EVAL, st, 4.0 * n * (2 + m + 1), 2.0 * n * (m + 24));
    if (vec) hipLaunchKernelGGL((interp_direct_kernel<true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, p, n / 4);
    else hipLaunchKernelGGL((interp_direct_kernel<false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, p, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int interp_direct_multi_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                            const float *h_xp /*nt x 5*/, float *const *h_out, int nt, int64_t n, hipStream_t st) {
    if (nt < 1 || nt > kMaxTicks) { set_error("interp_direct_multi: 1..%d ticks per launch", kMaxTicks); return NDCN_EINVAL; }
    DirectMultiArgs p;
    bool vec = (n % 4 == 0) && aligned16(y0) && aligned16(y1);
    const float *kk[kMaxTerms];
    float cc[kMaxTerms];
    int m = 0;
    for (int j = 0; j < 7; ++j) {
        if (!h_k[j]) { set_error("interp_direct_multi: null stage pointer"); return NDCN_EINVAL; }
        vec = vec && aligned16(h_k[j]);
        if (h_cmid[j] != 0.f) { kk[m] = h_k[j]; cc[m] = h_cmid[j]; ++m; }
    }
    if (m == 0) { kk[0] = h_k[0]; cc[0] = 0.f; m = 1; }
    bool dummy = true;
    fill_terms(p.f.mid, kk, cc, m, dummy);
    p.f.y0 = y0; p.f.y1 = y1; p.f.f0 = h_k[0]; p.f.f1 = h_k[6]; p.f.dt = dt;
    p.f.a = p.f.b = p.f.c = p.f.d = nullptr;
    p.nt = nt;
    for (int t = 0; t < kMaxTicks; ++t) {
        p.out[t] = h_out[t < nt ? t : 0];
        vec = vec && aligned16(p.out[t]);
        for (int q = 0; q < 5; ++q) p.xp[t][q] = h_xp[(t < nt ? t : 0) * 5 + q];
    }
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_EVAL, st, 4.0 * n * (2 + m + nt), 2.0 * n * (m + 16 + 9 * nt));
    if (vec) hipLaunchKernelGGL((interp_direct_multi_kernel<true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, p, n / 4);
    else hipLaunchKernelGGL((interp_direct_multi_kernel<false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, p, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int interp_eval_f32(const float *a, const float *b, const float *c, const float *d, const float *e, const float xp[5],
                    float *out, int64_t n, hipStream_t st) {
    EvalArgs p{a, b, c, d, e, xp[0], xp[1], xp[2], xp[3], xp[4], out};
    const bool vec = (n % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d) && aligned16(e) && aligned16(out);
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_EVAL, st, 4.0 * n * 6, 9.0 * n);
    if (vec) hipLaunchKernelGGL((interp_eval_kernel<true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, p, n / 4);
    else hipLaunchKernelGGL((interp_eval_kernel<false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, p, n);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

template <int OP>
static void launch_stage(bool vec, float *out, const float *y, const float *k1, const float *k2, const float *k3,
                         const float *k4, float dt, const float *dt_dev, int64_t n, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((fixed_stage_kernel<OP, true>), dim3(stream_grid_full(n / 4, 256)), dim3(256), 0, st, out, y, k1, k2, k3, k4, dt, dt_dev, n / 4);
    else hipLaunchKernelGGL((fixed_stage_kernel<OP, false>), dim3(stream_grid_full(n, 256)), dim3(256), 0, st, out, y, k1, k2, k3, k4, dt, dt_dev, n);
}

int fixed_stage_f32(int op, float *out, const float *y, const float *k1, const float *k2, const float *k3,
                    const float *k4, float dt, int64_t n, hipStream_t st, const float *dt_dev) {
    const int need = op <= 2 ? 1 : op == 3 ? 2 : op == 4 ? 3 : 4;
    const float *ks[4] = {k1, k2, k3, k4};
    bool vec = (n % 4 == 0) && aligned16(out) && aligned16(y);
    for (int j = 0; j < need; ++j) {
        if (!ks[j]) { set_error("fixed_stage: op %d needs %d stage pointers", op, need); return NDCN_EINVAL; }
        vec = vec && aligned16(ks[j]);
    }
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_STAGE, st, 4.0 * n * (need + 2), 2.0 * n * need);
    switch (op) {
        case 0: launch_stage<0>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        case 1: launch_stage<1>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        case 2: launch_stage<2>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        case 3: launch_stage<3>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        case 4: launch_stage<4>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        case 5: launch_stage<5>(vec, out, y, k1, k2, k3, k4, dt, dt_dev, n, st); break;
        default: set_error("fixed_stage: unknown op %d", op); return NDCN_EINVAL;
    }
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
