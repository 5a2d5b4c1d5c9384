// Fused ODEFunc right-hand side, second generation (H = 256):
//
//     K = relu((A X) W^T + b)          [neural_dynamics.py:27-36]
//   + optionally, in the same pass, the Runge-Kutta algebra that consumes K:
//       COMBINE :  y_next = y0 + sum_m c_m k_m   (k_new last)      [rk_common.py:51 / misc.py:22-25]
//       ERROR   :  sum ((sum_m c_m k_m) / (atol + rtol max(|y0|, |y1|)))^2 and the non-finite count of y1
//                                                                  [rk_common.py:60, misc.py:146-157, dopri5.py:101]
//
// What changed against rhs_fused.hip (measured on the 1M-node grid, MI355X):
//   * the gather was bound by the CU's vector-memory pipe (9 x 1 KiB loads per row, ~18 cycles each even on a
//     hit) and by latency (few producer waves).  Producers now stage, per group of 8 consecutive rows, the
//     DISTINCT neighbour rows once in LDS (ndcn_csr::ug_* plan: 30 rows for 72 non-zeros on the lattice) with
//     the next group's fetches in flight while the current group's rows are summed out of LDS;
//   * consumers no longer store their accumulators with 4-byte scattered stores: they drop the K tile into the
//     LDS tile they just consumed, and the producers stream it out as whole 1 KiB rows - together with the RK
//     stage algebra, whose extra panels are row-local and ride on the same coalesced pass.
//
// Workgroup = 4 consumer waves (one per SIMD, fp32 MFMA) + NPROD producer waves, persistent, one per CU.
// Tile = 32 rows.  LDS: S[2][32][260] fp32 (66.5 KB) + stage[2][30][256] fp32 (60 KB) + sync word.
//
//   phase A(t): consumers  MFMA on S[t&1]                         -> K_t in registers
//               producers  epilogue of K_{t-1} (in S[(t-1)&1]) -> HBM ; then gather S_{t+1} into S[(t-1)&1]
//   barrier
//   phase B(t): consumers  K_t (+bias, relu) -> S[t&1] row-major
//   barrier
// A producer wave always touches the same rows of a tile (row 8 g + p of every group g), first reading K
// out of them, then overwriting them with the gathered S - no cross-wave hazard inside phase A.  Producer waves
// synchronise among themselves (stage complete / stage free) with an LDS counter, so the MFMA waves never wait
// on a memory phase boundary.
#include <stdlib.h>

#include "kernels.h"

#pragma clang fp contract(off)   // the RK algebra below must round like the reference's separate mul / add ops

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH2 = 256;
constexpr int kTile2 = 32;
constexpr int kLd2 = kH2 + 4;
constexpr int kTileFloats2 = kTile2 * kLd2;
constexpr int kGroupRows = 8;              // must equal the union plan's ug_rows
constexpr int kGroupsPerTile = kTile2 / kGroupRows;
constexpr int kStageCap = 30;              // distinct neighbour rows staged per group
constexpr int kStageFloats = kStageCap * kH2;
constexpr int kMaxPrev = 7;

struct Fused2Args {
    const int *rowptr, *colidx;
    const float *val;
    const int *ug_ptr, *ug_cols;
    const unsigned short *ug_lidx;
    const float *X, *Xh;
    int n_own;
    const float *Wp, *bias;
    float *K;                               // relu(...) output panel
    int n_rows, n_tiles, relu;
    // epilogue
    const float *y0;
    const float *kprev[kMaxPrev];
    float c[kMaxPrev + 1];                  // c[0 .. n_prev-1] for kprev, c[n_prev] for the new K
    int n_prev;
    float *y_next;
    float rtol, atol;
    double *partials;                       // ERROR: [gridDim.x * NPROD][2]
};

enum { MODE_PLAIN = 0, MODE_COMBINE = 1, MODE_ERROR = 2 };

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 a) {
    return (f32x4){fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w)};
}

template <int U>
__device__ __forceinline__ void stage_batch(int li, float v, int i, const f32x4 *st, int lane, f32x4 &acc) {
    int ll[U];
    float vv[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        ll[q] = __builtin_amdgcn_readlane(li, i + q);
        vv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i + q));
    }
    f32x4 x[U];
#pragma unroll
    for (int q = 0; q < U; ++q) x[q] = st[ll[q] * 64 + lane];
#pragma unroll
    for (int q = 0; q < U; ++q) acc = fma4(vv[q], x[q], acc);
}

template <int U, bool HALO>
__device__ __forceinline__ void direct_batch(int c, float v, int i, const f32x4 *__restrict__ X,
                                             const f32x4 *__restrict__ Xh, int n_own, int lane, f32x4 &acc) {
    int cc[U];
    float vv[U];
    const f32x4 *pp[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        cc[q] = __builtin_amdgcn_readlane(c, i + q);
        vv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i + q));
        pp[q] = X;
        if (HALO && cc[q] >= n_own) { pp[q] = Xh; cc[q] -= n_own; }
    }
    f32x4 x[U];
#pragma unroll
    for (int q = 0; q < U; ++q) x[q] = pp[q][(size_t)cc[q] * 64 + lane];
#pragma unroll
    for (int q = 0; q < U; ++q) acc = fma4(vv[q], x[q], acc);
}

// LDS counter barrier among the producer waves only
__device__ __forceinline__ void producer_barrier(int *cnt, int nprod, int &target, int lane) {
    target += nprod;
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

template <int NPROD, bool HALO, int MODE>
__global__ __launch_bounds__(256 + 64 * NPROD) void rhs_fused2_kernel(Fused2Args a) {
    static_assert(NPROD == kGroupRows, "one producer wave per row of a group");
    __shared__ __attribute__((aligned(16))) float s_mem[2 * kTileFloats2 + 2 * kStageFloats + 4];
    float *s_tile = s_mem;
    float *s_stage = s_mem + 2 * kTileFloats2;
    int *s_cnt = reinterpret_cast<int *>(s_mem + 2 * kTileFloats2 + 2 * kStageFloats);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int p = wave - 4;                                   // producer index 0..NPROD-1
    const f32x4 *X = reinterpret_cast<const f32x4 *>(a.X);
    const f32x4 *Xh = reinterpret_cast<const f32x4 *>(a.Xh);
    const f32x4 *Wp = reinterpret_cast<const f32x4 *>(a.Wp);

    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();

    // tiles of this workgroup: XCD x owns a contiguous chunk; its workgroups take tiles round-robin
    const int xcd = blockIdx.x % kXcds;
    const int wg = blockIdx.x / kXcds;
    const int wgs_per_xcd = gridDim.x / kXcds;
    const int chunk = (a.n_tiles + kXcds - 1) / kXcds;
    const int t_lo = xcd * chunk, t_hi = min(a.n_tiles, t_lo + chunk);
    const int t_first = t_lo + wg;
    const int my_tiles = t_first < t_hi ? (t_hi - t_first + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;                                // uniform per workgroup

    int pb_target = 0;                                        // producer-barrier epoch
    double err_sum = 0.0, err_bad = 0.0;                      // MODE_ERROR, producers

    // ---- producer helpers -------------------------------------------------------------------------
    // Fetch (issue only) everything this wave needs for group `g`: its share of the group's union rows of X
    // (into registers) and the index data of ITS row of the group (row 8 g + p).
    f32x4 pend[4];
    int pend_nu = 0, pend_j0 = 0, pend_j1 = 0, pend_idx = 0;
    float pend_v = 0.f;
    auto issue_stage = [&](int g) {
        const int u0 = a.ug_ptr[g];
        const int nu = a.ug_ptr[g + 1] - u0;
        pend_nu = nu;
        const int r = g * kGroupRows + p;
        pend_j0 = pend_j1 = 0;
        if (r < a.n_rows) { pend_j0 = a.rowptr[r]; pend_j1 = a.rowptr[r + 1]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = p + NPROD * q;
            if (u < nu) {
                int c = a.ug_cols[u0 + u];
                const f32x4 *src = X;
                if (HALO && c >= a.n_own) { src = Xh; c -= a.n_own; }
                pend[q] = src[(size_t)c * 64 + lane];
            }
        }
        pend_idx = 0; pend_v = 0.f;
        if (lane < pend_j1 - pend_j0) {
            pend_idx = nu > 0 ? (int)a.ug_lidx[pend_j0 + lane] : a.colidx[pend_j0 + lane];
            pend_v = a.val[pend_j0 + lane];
        }
    };
    auto commit_stage = [&](int sb) {
        f32x4 *st = reinterpret_cast<f32x4 *>(s_stage + sb * kStageFloats);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = p + NPROD * q;
            if (u < pend_nu) st[u * 64 + lane] = pend[q];
        }
    };
    // Sum one row out of stage buffer `sb` (or directly when its group has no union) into LDS row dst.
    // (j0, j1, idx, v) = the row's extent and its first <= 64 (index, value) pairs, prefetched by issue_stage.
    auto gather_row = [&](bool staged, int j0, int j1, int idx, float v, int sb, float *dst) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 *st = reinterpret_cast<const f32x4 *>(s_stage + sb * kStageFloats);
        for (int jb = j0; jb < j1; jb += 64) {
            const int cnt = min(64, j1 - jb);
            if (jb != j0) {
                idx = 0; v = 0.f;
                if (lane < cnt) {
                    idx = staged ? (int)a.ug_lidx[jb + lane] : a.colidx[jb + lane];
                    v = a.val[jb + lane];
                }
            }
            int i = 0;
            if (staged) {
                for (; i + 8 <= cnt; i += 8) stage_batch<8>(idx, v, i, st, lane, acc);
                if (i + 4 <= cnt) { stage_batch<4>(idx, v, i, st, lane, acc); i += 4; }
                if (i + 2 <= cnt) { stage_batch<2>(idx, v, i, st, lane, acc); i += 2; }
                if (i < cnt) stage_batch<1>(idx, v, i, st, lane, acc);
            } else {
                for (; i + 8 <= cnt; i += 8) direct_batch<8, HALO>(idx, v, i, X, Xh, a.n_own, lane, acc);
                if (i + 4 <= cnt) { direct_batch<4, HALO>(idx, v, i, X, Xh, a.n_own, lane, acc); i += 4; }
                if (i + 2 <= cnt) { direct_batch<2, HALO>(idx, v, i, X, Xh, a.n_own, lane, acc); i += 2; }
                if (i < cnt) direct_batch<1, HALO>(idx, v, i, X, Xh, a.n_own, lane, acc);
            }
        }
        *reinterpret_cast<f32x4 *>(dst + 4 * lane) = acc;
    };
    // Gather tile `t` into LDS tile `dst_tile`: 4 groups, stage double-buffered across groups.
    // Precondition: group 4t's fetches were issued (issue_stage) by the caller.
    auto gather_tile = [&](int t, float *dst_tile, int t_next_or_neg) {
        for (int k = 0; k < kGroupsPerTile; ++k) {
            const int g = t * kGroupsPerTile + k;
            commit_stage(k & 1);                               // this wave's part of group g is in LDS
            const bool staged = pend_nu > 0;
            const int j0 = pend_j0, j1 = pend_j1, idx = pend_idx;
            const float v = pend_v;
            producer_barrier(s_cnt, NPROD, pb_target, lane);  // all parts landed; stage[(k+1)&1] is free
            // next group's fetches fly while this group is summed
            int gn = -1;
            if (k + 1 < kGroupsPerTile) gn = g + 1;
            else if (t_next_or_neg >= 0) gn = t_next_or_neg * kGroupsPerTile;
            if (gn >= 0 && gn * kGroupRows < a.n_rows) issue_stage(gn);
            else { pend_nu = 0; pend_j0 = pend_j1 = 0; }
            gather_row(staged, j0, j1, idx, v, k & 1, dst_tile + (k * kGroupRows + p) * kLd2);
        }
    };
    // Stream K rows of tile `t` out of LDS tile `src_tile` (+ RK algebra).
    auto epilogue_tile = [&](int t, const float *src_tile) {
#pragma unroll 1
        for (int k = 0; k < kGroupsPerTile; ++k) {
            const int lr = k * kGroupRows + p;
            const int r = t * kTile2 + lr;
            if (r >= a.n_rows) continue;
            const f32x4 kn = *reinterpret_cast<const f32x4 *>(src_tile + lr * kLd2 + 4 * lane);
            const size_t off = (size_t)r * 64 + lane;
            __builtin_nontemporal_store(kn, reinterpret_cast<f32x4 *>(a.K) + off);
            if (MODE != MODE_PLAIN) {
                f32x4 s;
                bool first = true;
#pragma unroll
                for (int m = 0; m < kMaxPrev; ++m)
                    if (m < a.n_prev) {
                        const f32x4 km = reinterpret_cast<const f32x4 *>(a.kprev[m])[off];
                        const f32x4 term = km * a.c[m];
                        s = first ? term : s + term;
                        first = false;
                    }
                const f32x4 tn = kn * a.c[a.n_prev];
                s = first ? tn : s + tn;
                const f32x4 y0v = reinterpret_cast<const f32x4 *>(a.y0)[off];
                if (MODE == MODE_COMBINE) {
                    __builtin_nontemporal_store(y0v + s, reinterpret_cast<f32x4 *>(a.y_next) + off);
                } else {
                    const f32x4 y1v = X[off];                  // the input of this evaluation is y1
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tol = a.atol + a.rtol * fmaxf(fabsf(y0v[e]), fabsf(y1v[e]));
                        const float q = s[e] / tol;
                        err_sum += (double)(q * q);
                        err_bad += (double)(int)(!(fabsf(y1v[e]) <= 3.402823466e38f));
                    }
                }
            }
        }
    };

    // ---- consumer -----------------------------------------------------------------------------------
    f32x16 acc0, acc1;
    auto mfma_tile = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        const float *ap = src + (lane & 31) * kLd2 + 128 * (lane >> 5);
        const f32x4 *b0p = Wp + (size_t)(2 * wave) * 32 * 64 + lane;
        const f32x4 *b1p = b0p + 32 * 64;
        f32x4 b0 = b0p[0], b1 = b1p[0];
        f32x4 b0n = b0p[64], b1n = b1p[64];
#pragma unroll 2
        for (int q = 0; q < 32; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 4 * q);
            const f32x4 c0 = b0, c1 = b1;
            b0 = b0n; b1 = b1n;
            if (q + 2 < 32) { b0n = b0p[(q + 2) * 64]; b1n = b1p[(q + 2) * 64]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c1[e], acc1, 0, 0, 0);
            }
        }
    };
    auto dump_tile = [&](float *dst) {
        // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]; wave owns columns [64 wave, 64 wave + 64)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = 64 * wave + 32 * n + (lane & 31);
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float o = (n == 0 ? acc0[r] : acc1[r]) + bv;
                if (a.relu) o = fmaxf(o, 0.f);
                dst[m * kLd2 + col] = o;
            }
        }
    };

    // ---- prologue: producers gather the first tile into S[0] ----------------------------------------
    if (producer) {
        issue_stage(t_first * kGroupsPerTile);
        gather_tile(t_first, s_tile, my_tiles > 1 ? t_first + wgs_per_xcd : -1);
    }
    __syncthreads();

    for (int it = 0; it < my_tiles; ++it) {
        const int t = t_first + it * wgs_per_xcd;
        const int b = it & 1;
        float *cur = s_tile + b * kTileFloats2;
        float *oth = s_tile + (b ^ 1) * kTileFloats2;
        // ---- phase A
        if (producer) {
            if (it > 0) epilogue_tile(t - wgs_per_xcd, oth);
            if (it + 1 < my_tiles) gather_tile(t + wgs_per_xcd, oth, it + 2 < my_tiles ? t + 2 * wgs_per_xcd : -1);
        } else {
            mfma_tile(cur);
        }
        __syncthreads();
        // ---- phase B
        if (!producer) dump_tile(cur);
        __syncthreads();
    }
    // ---- drain: epilogue of the last tile
    if (producer) {
        const int t_last = t_first + (my_tiles - 1) * wgs_per_xcd;
        epilogue_tile(t_last, s_tile + ((my_tiles - 1) & 1) * kTileFloats2);
        if (MODE == MODE_ERROR) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                err_sum += __shfl_down(err_sum, off, 64);
                err_bad += __shfl_down(err_bad, off, 64);
            }
            if (lane == 0) {
                a.partials[2 * (blockIdx.x * NPROD + p)] = err_sum;
                a.partials[2 * (blockIdx.x * NPROD + p) + 1] = err_bad;
            }
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
    if (!enabled || H != kH2) return 0;
    if (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) return 0;
    return (A && A->ug_ptr && A->ug_rows == kGroupRows && A->ug_cap <= kStageCap) ? 1 : 0;
}

int64_t rhs_fused2_partials_bytes() { return (int64_t)kCus * kGroupRows * 2 * sizeof(double); }

// mode: 0 plain; 1 combine (y_next = y0 + sum c_m k_m, new K last); 2 error (d_out[0..1], d_ws scratch)
int rhs_fused2_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b,
                   float *K, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c,
                   int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    if (n_prev < 0 || n_prev > kMaxPrev) { set_error("rhs_fused2: at most %d previous stages", kMaxPrev); return NDCN_EINVAL; }
    Fused2Args a;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
    a.ug_ptr = A->ug_ptr; a.ug_cols = A->ug_cols; a.ug_lidx = A->ug_lidx;
    a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.Wp = Wp; a.bias = b; a.K = K;
    a.n_rows = n_rows; a.n_tiles = (n_rows + kTile2 - 1) / kTile2; a.relu = (flags & NDCN_F_RELU) ? 1 : 0;
    a.y0 = y0; a.n_prev = n_prev; a.y_next = y_next; a.rtol = rtol; a.atol = atol;
    a.partials = static_cast<double *>(d_ws);
    for (int m = 0; m < kMaxPrev; ++m) a.kprev[m] = (m < n_prev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kMaxPrev; ++m) a.c[m] = (mode != MODE_PLAIN && m <= n_prev) ? h_c[m] : 0.f;
    int per_xcd = kCus / kXcds;
    const int need = (a.n_tiles + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds), block(256 + 64 * kGroupRows);
    const double P = 4.0 * kH2 * (double)A->n_rows;
    double bytes = 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * kH2 * (double)(A->n_rows + A->n_cols) + 4.0 * kH2 * kH2;
    if (mode == MODE_COMBINE) bytes += P * (n_prev + 2);
    if (mode == MODE_ERROR) bytes += P * (n_prev + 2);
    ProfScope prof(PROF_RHS_FUSED, st, bytes, 2.0 * A->nnz * kH2 + 2.0 * (double)A->n_rows * kH2 * kH2);
#define NDCN_F2(HALO_, MODE_) \
    hipLaunchKernelGGL((rhs_fused2_kernel<kGroupRows, HALO_, MODE_>), grid, block, 0, st, a)
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
        hipLaunchKernelGGL(fused2_finish_kernel, dim3(1), dim3(256), 0, st, a.partials, (int)grid.x * kGroupRows, d_out);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
