// CSR SpMM  Y = alpha * (A X) [relu]  for a dense row-major fp32 feature panel X (n x H).
//
// Replaces torch.sparse.mm(A, x) at neural_dynamics.py:29 (and its siblings, see include/ndcn_hip.h).
//
// Shape of the work on MI355X: HBM-bound gather.  Algorithmic bytes 8 nnz + 4 (N+1) + 8 N H
// (SURVEY.md 8d).  Design:
//   * one 256-thread workgroup owns a block of consecutive rows; the block's slice of rowptr and its
//     (contiguous) slice of colidx/val are staged into LDS with coalesced loads, once;
//   * a group of LPR lanes walks one row: every lane owns one 16-byte column slot of the H-wide panel row,
//     so a neighbour row is fetched by ONE fully coalesced access (H=256: 64 lanes x 16 B = the 1 KiB row);
//     the neighbour loop is unrolled 4x so four independent row fetches are in flight per group;
//   * logical row blocks are remapped so each XCD walks a contiguous range of rows: neighbouring rows
//     share neighbours, which then hit in that XCD's private 4 MiB L2;
//   * accumulation order = stored (column-ascending) order, one fma per non-zero.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ndcn {

constexpr int kSpmmThreads = 256;
constexpr int kStageCap = 2048;     // staged non-zeros per workgroup (16 KiB of LDS -> 8 workgroups per CU)
constexpr int kMaxRowsPerBlock = 256;

template <int VW> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ __forceinline__ void vfma(float4 &a, float v, const float4 &x) {
    a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
}
__device__ __forceinline__ void vfma(float &a, float v, const float &x) { a = fmaf(v, x, a); }
__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int VW> __device__ __forceinline__ typename VecT<VW>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return vzero4(); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
__device__ __forceinline__ float4 vfinish(float4 a, float alpha, bool relu) {
    a.x *= alpha; a.y *= alpha; a.z *= alpha; a.w *= alpha;
    if (relu) { a.x = relu_nan(a.x); a.y = relu_nan(a.y); a.z = relu_nan(a.z); a.w = relu_nan(a.w); }
    return a;
}
__device__ __forceinline__ float vfinish(float a, float alpha, bool relu) {
    a *= alpha;
    return relu ? relu_nan(a) : a;
}

// One row: acc = sum_j val[j] * X[col[j], slot]; `cols`/`vals` index from `j0` (LDS or global view).
template <int VW, bool HALO>
__device__ __forceinline__ typename VecT<VW>::type
row_gather(const int *__restrict__ cols, const float *__restrict__ vals, int j0, int j1,
           const typename VecT<VW>::type *__restrict__ X, const typename VecT<VW>::type *__restrict__ Xh,
           int n_own, size_t row_stride /* in vector units */, int slot) {
    using V = typename VecT<VW>::type;
    V acc = vzero<VW>();
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        int c0 = cols[j], c1 = cols[j + 1], c2 = cols[j + 2], c3 = cols[j + 3];
        float v0 = vals[j], v1 = vals[j + 1], v2 = vals[j + 2], v3 = vals[j + 3];
        const V *p0 = X, *p1 = X, *p2 = X, *p3 = X;
        if (HALO) {
            if (c0 >= n_own) { p0 = Xh; c0 -= n_own; }
            if (c1 >= n_own) { p1 = Xh; c1 -= n_own; }
            if (c2 >= n_own) { p2 = Xh; c2 -= n_own; }
            if (c3 >= n_own) { p3 = Xh; c3 -= n_own; }
        }
        V x0 = p0[(size_t)c0 * row_stride + slot];
        V x1 = p1[(size_t)c1 * row_stride + slot];
        V x2 = p2[(size_t)c2 * row_stride + slot];
        V x3 = p3[(size_t)c3 * row_stride + slot];
        vfma(acc, v0, x0); vfma(acc, v1, x1); vfma(acc, v2, x2); vfma(acc, v3, x3);
    }
    for (; j < j1; ++j) {
        int c = cols[j];
        float v = vals[j];
        const V *p = X;
        if (HALO && c >= n_own) { p = Xh; c -= n_own; }
        vfma(acc, v, p[(size_t)c * row_stride + slot]);
    }
    return acc;
}

// VW: floats per lane slot (4 -> float4, needs H % 4 == 0 ; 1 -> scalar).
// LPR: lanes that share one row (power of two <= 64).
template <int VW, int LPR, bool HALO>
__global__ __launch_bounds__(kSpmmThreads) void spmm_csr_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const float *__restrict__ val,
    const float *__restrict__ Xf, const float *__restrict__ Xhf, int n_own, float *__restrict__ Yf,
    int n_rows, int H, float alpha, int relu, int rows_per_block) {
    using V = typename VecT<VW>::type;
    __shared__ int s_rp[kMaxRowsPerBlock + 1];
    __shared__ int s_col[kStageCap];
    __shared__ float s_val[kStageCap];

    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = blk * rows_per_block;
    if (r0 >= n_rows) return;
    const int nr = min(rows_per_block, n_rows - r0);
    const int tid = threadIdx.x;

    for (int i = tid; i <= nr; i += kSpmmThreads) s_rp[i] = rowptr[r0 + i];
    __syncthreads();
    const int e_lo = s_rp[0];
    const int e_hi = s_rp[nr];
    const int n_stage = min(e_hi - e_lo, kStageCap);
    for (int i = tid; i < n_stage; i += kSpmmThreads) {
        s_col[i] = colidx[e_lo + i];
        s_val[i] = val[e_lo + i];
    }
    __syncthreads();

    const V *X = reinterpret_cast<const V *>(Xf);
    const V *Xh = reinterpret_cast<const V *>(Xhf);
    V *Y = reinterpret_cast<V *>(Yf);
    const int slots = H / VW;                       // vector slots per panel row
    const size_t stride = (size_t)slots;
    constexpr int kGroups = kSpmmThreads / LPR;     // rows in flight per workgroup
    const int grp = tid / LPR;
    const int li = tid % LPR;

    for (int r = grp; r < nr; r += kGroups) {
        const int j0 = s_rp[r], j1 = s_rp[r + 1];
        const bool staged = (j1 - e_lo) <= kStageCap;
        for (int slot = li; slot < slots; slot += LPR) {
            V acc;
            if (staged)
                acc = row_gather<VW, HALO>(s_col, s_val, j0 - e_lo, j1 - e_lo, X, Xh, n_own, stride, slot);
            else
                acc = row_gather<VW, HALO>(colidx, val, j0, j1, X, Xh, n_own, stride, slot);
            Y[(size_t)(r0 + r) * stride + slot] = vfinish(acc, alpha, relu != 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wide panels (H a multiple of 256): persistent waves, ONE row per wave at a time, no LDS, no barriers.
//
// Why a second kernel: with 64 rows per workgroup and ~256 workgroups resident per XCD the rows in flight on
// one XCD span ~16 lattice rows of the 1000 x 1000 grid; their neighbour rows (18 MB of X) do not fit the
// XCD's 4 MiB L2, so every X row was re-fetched ~3x from the Infinity Cache / HBM (measured 2.7 TB/s
// algorithmic).  Here every XCD owns a contiguous chunk of rows and its waves walk that chunk ROW-INTERLEAVED
// (wave w takes rows w, w + W, w + 2W, ...), so at any moment the rows in flight on an XCD are ~W consecutive
// rows and the X window they touch (W + 2 * bandwidth rows) stays L2-resident.
// The row's (col, val) pairs are fetched by one coalesced load (lane j holds entry j) and broadcast with
// v_readlane; each neighbour row is one 1 KiB coalesced access per 256 columns; 4 neighbour fetches in flight.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// U neighbour rows of one output row: broadcast (col, val) of entries i .. i+U-1 from the lanes that hold
// them, issue the U (x NV) coalesced fetches back to back, then accumulate in stored order.
template <int U, int NV, bool HALO>
__device__ __forceinline__ void wide_batch(int c, float v, int i, const f32x4 *__restrict__ X,
                                           const f32x4 *__restrict__ Xh, int n_own, int lane, f32x4 (&acc)[NV]) {
    constexpr size_t stride = 64 * NV;
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
    f32x4 x[U][NV];
#pragma unroll
    for (int q = 0; q < U; ++q)
#pragma unroll
        for (int u = 0; u < NV; ++u) x[q][u] = pp[q][(size_t)cc[q] * stride + lane + 64 * u];
#pragma unroll
    for (int q = 0; q < U; ++q)
#pragma unroll
        for (int u = 0; u < NV; ++u) acc[u] = vv[q] * x[q][u] + acc[u];
}

template <int NV, bool HALO>
__device__ __forceinline__ void wide_chunk(int c, float v, int cnt, const f32x4 *__restrict__ X,
                                           const f32x4 *__restrict__ Xh, int n_own, int lane, f32x4 (&acc)[NV]) {
    int i = 0;
    for (; i + 8 <= cnt; i += 8) wide_batch<8, NV, HALO>(c, v, i, X, Xh, n_own, lane, acc);
    if (i + 4 <= cnt) { wide_batch<4, NV, HALO>(c, v, i, X, Xh, n_own, lane, acc); i += 4; }
    if (i + 2 <= cnt) { wide_batch<2, NV, HALO>(c, v, i, X, Xh, n_own, lane, acc); i += 2; }
    if (i < cnt) wide_batch<1, NV, HALO>(c, v, i, X, Xh, n_own, lane, acc);
}

// Runge-Kutta epilogue of the no_control right-hand side relu(A X) on operators without a group-record plan (any graph;
// H = 256): the row-local panels are requested at the start of the row, ahead of its neighbour fetches, and consumed
// after the sum - same term order and roundings as rk.hip / rhs_fused2.hip / spmm_rec.hip.
constexpr int kWideMaxPrev = 5;
struct WideEpi {
    const float *y0;
    const float *kprev[kWideMaxPrev];
    float *y_next;
    double *partials;                 // ERROR: [gridDim.x * 4][2]
    float c[kWideMaxPrev + 1];
    int n_prev;
    float rtol, atol;
    const float *c_dev;               // nullable: coefficients in device memory (hipGraph replay)
    const float *y1;                  // ERROR: the state of the error record, by row of this launch
    float *y_aux;                     // COMBINE, nullable: second linear combination (no y0), coefficients c2[] (by value only)
    float c2[kWideMaxPrev + 1];
};
enum { WIDE_PLAIN = 0, WIDE_COMBINE = 1, WIDE_ERROR = 2, WIDE_RK4 = 3 };

template <int NV, bool HALO, int MODE>
__global__ __launch_bounds__(256) void spmm_wide_kernel(const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                        const float *__restrict__ val, const int *__restrict__ order,
                                                        const float *__restrict__ Xf,
                                                        const float *__restrict__ Xhf, int n_own,
                                                        float *__restrict__ Yf, int n_rows, float alpha, int relu, WideEpi e) {
    static_assert(MODE == WIDE_PLAIN || NV == 1, "the RK epilogue is built for H = 256");
    double err_sum = 0.0, err_bad = 0.0;
    const int np = MODE == WIDE_PLAIN ? 0 : e.n_prev;
    auto coef = [&](int m) { return (MODE != WIDE_PLAIN && e.c_dev) ? e.c_dev[m] : e.c[m]; };
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x % kXcds;
    const int waves_per_xcd = (gridDim.x / kXcds) * 4;
    const int w = (blockIdx.x / kXcds) * 4 + (threadIdx.x >> 6);
    const int chunk = (n_rows + kXcds - 1) / kXcds;
    const int row_lo = xcd * chunk;
    const int row_hi = min(n_rows, row_lo + chunk);
    const f32x4 *X = reinterpret_cast<const f32x4 *>(Xf);
    const f32x4 *Xh = reinterpret_cast<const f32x4 *>(Xhf);
    f32x4 *Y = reinterpret_cast<f32x4 *>(Yf);
    constexpr size_t stride = 64 * NV;                    // float4 slots per panel row

    int pos = __builtin_amdgcn_readfirstlane(row_lo + w);          // position in the walk order
    if (pos >= row_hi) {
        if (MODE == WIDE_ERROR && lane == 0) {                     // the finish kernel sums EVERY slot
            e.partials[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = 0.0;
            e.partials[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = 0.0;
        }
        return;
    }
    int r = order ? order[pos] : pos;
    // software pipeline: the first <= 64 (col, val) pairs of the NEXT row are fetched while this row's
    // neighbour rows are in flight
    int j0 = rowptr[r], j1 = rowptr[r + 1];
    int c = 0;
    float v = 0.f;
    if (lane < j1 - j0) { c = colidx[j0 + lane]; v = val[j0 + lane]; }
    while (true) {
        const int pn = pos + waves_per_xcd;
        const bool more = pn < row_hi;
        int rn = 0, nj0 = 0, nj1 = 0;
        if (more) { rn = order ? order[pn] : pn; nj0 = rowptr[rn]; nj1 = rowptr[rn + 1]; }
        f32x4 acc[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 pk[kWideMaxPrev], py0, py1;
        if (MODE != WIDE_PLAIN) {                                  // requested ahead of the neighbour rows
            const size_t o = (size_t)r * 64 + lane;
#pragma unroll
            for (int m = 0; m < kWideMaxPrev; ++m)
                if (m < np) pk[m] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(e.kprev[m]) + o);
            py0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(e.y0) + o);
            if (MODE == WIDE_ERROR) py1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(e.y1) + o);
        }
        int nc = 0;
        float nv = 0.f;
        if (j1 - j0 <= 64) {
            // common case: the whole row sits in the lanes already
            if (more && lane < nj1 - nj0) { nc = colidx[nj0 + lane]; nv = val[nj0 + lane]; }
            wide_chunk<NV, HALO>(c, v, j1 - j0, X, Xh, n_own, lane, acc);
        } else {
            wide_chunk<NV, HALO>(c, v, 64, X, Xh, n_own, lane, acc);
            for (int jb = j0 + 64; jb < j1; jb += 64) {
                const int cnt = min(64, j1 - jb);
                int c2 = 0;
                float v2 = 0.f;
                if (lane < cnt) { c2 = colidx[jb + lane]; v2 = val[jb + lane]; }
                wide_chunk<NV, HALO>(c2, v2, cnt, X, Xh, n_own, lane, acc);
            }
            if (more && lane < nj1 - nj0) { nc = colidx[nj0 + lane]; nv = val[nj0 + lane]; }
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            f32x4 o = acc[u] * alpha;
            if (relu) { o.x = relu_nan(o.x); o.y = relu_nan(o.y); o.z = relu_nan(o.z); o.w = relu_nan(o.w); }
            __builtin_nontemporal_store(o, &Y[(size_t)r * stride + lane + 64 * u]);
            if (MODE != WIDE_PLAIN && u == 0) {
#pragma clang fp contract(off)       // the RK algebra rounds like the reference's separate mul / add ops
                const size_t oo = (size_t)r * 64 + lane;
                f32x4 *yn = reinterpret_cast<f32x4 *>(e.y_next);
                if (MODE == WIDE_RK4) {
                    // rk4_alt_step_func (rk_common.py:72-78), same operator order as fixed_stage_kernel ops 2-5
                    const float dt = coef(0);
                    f32x4 sdt;
                    if (np == 0) sdt = (o * dt) / 3.f;
                    else if (np == 1) sdt = (pk[0] / -3.f + o) * dt;
                    else if (np == 2) sdt = ((pk[0] - pk[1]) + o) * dt;
                    else sdt = (((pk[0] + pk[1] * 3.f) + pk[2] * 3.f) + o) * (dt / 8.f);
                    __builtin_nontemporal_store(py0 + sdt, yn + oo);
                } else {
                    // sum of the stages left to right, the new one last (misc.py:22-25), each product rounded on its own
                    f32x4 sm = o * coef(np);
                    if (np > 0) {
                        f32x4 uu = pk[0] * coef(0);
#pragma unroll
                        for (int m = 1; m < kWideMaxPrev; ++m)
                            if (m < np) uu = uu + pk[m] * coef(m);
                        sm = uu + sm;
                    }
                    if (MODE == WIDE_COMBINE) {
                        __builtin_nontemporal_store(py0 + sm, yn + oo);
                        if (e.y_aux) {
                            f32x4 w2 = o * e.c2[np];
                            if (np > 0) {
                                f32x4 u2 = pk[0] * e.c2[0];
#pragma unroll
                                for (int m = 1; m < kWideMaxPrev; ++m)
                                    if (m < np) u2 = u2 + pk[m] * e.c2[m];
                                w2 = u2 + w2;
                            }
                            __builtin_nontemporal_store(w2, reinterpret_cast<f32x4 *>(e.y_aux) + oo);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float tol = e.atol + e.rtol * max_nan(fabsf(py0[q]), fabsf(py1[q]));
                            const float z = sm[q] / tol;
                            err_sum += (double)(z * z);
                            err_bad += (double)(int)(!(fabsf(py1[q]) <= 3.402823466e38f));
                        }
                    }
                }
            }
        }
        if (!more) break;
        pos = pn; r = rn; j0 = nj0; j1 = nj1; c = nc; v = nv;
    }
    if (MODE == WIDE_ERROR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            err_sum += __shfl_down(err_sum, off, 64);
            err_bad += __shfl_down(err_bad, off, 64);
        }
        if (lane == 0) {
            e.partials[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = err_sum;
            e.partials[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = err_bad;
        }
    }
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

template <int NV, int MODE = WIDE_PLAIN>
static int launch_wide(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, float alpha,
                       uint32_t flags, hipStream_t st, const WideEpi *epi = nullptr, int *n_partials = nullptr) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    static const int bpc = env_int("NDCN_SPMM_BLOCKS_PER_CU", 4);          // 4 waves each
    int per_xcd = (kCus / kXcds) * bpc;
    const int need = ((n_rows + kXcds - 1) / kXcds + 3) / 4;               // no more waves than rows
    if (per_xcd > need) per_xcd = need < 1 ? 1 : need;
    const int relu = (flags & NDCN_F_RELU) ? 1 : 0;
    const dim3 grid(per_xcd * kXcds), block(256);
    WideEpi e = {};
    if (epi) e = *epi;
    if (n_partials) *n_partials = (int)grid.x * 4;
    if (Xh)
        hipLaunchKernelGGL((spmm_wide_kernel<NV, true, MODE>), grid, block, 0, st, A->rowptr, A->colidx, A->val,
                           A->row_order, X, Xh, (int)n_own, Y, n_rows, alpha, relu, e);
    else
        hipLaunchKernelGGL((spmm_wide_kernel<NV, false, MODE>), grid, block, 0, st, A->rowptr, A->colidx, A->val,
                           A->row_order, X, Xh, (int)n_own, Y, n_rows, alpha, relu, e);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int spmm_wide_rk_supported(const ndcn_csr *A, int H) {
    static const int enabled = env_int("NDCN_SPMM_WIDE_RK", 1);
    return enabled && A && H == 256 && A->n_rows > 0;
}

// K = relu(A X) plus the RK algebra in the row SpMM's epilogue (modes and arguments as spmm_rec_f32 / rhs_fused2_f32)
int spmm_wide_rk_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *K, uint32_t flags, int mode,
                     const float *y0, const float *const *h_kprev, const float *h_c, int n_prev, float *y_next, float rtol,
                     float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev, const RkOpt *opt) {
    if (n_prev < 0 || n_prev > kWideMaxPrev || (mode == WIDE_RK4 && n_prev > 3)) { set_error("spmm_wide_rk: bad stage count"); return NDCN_EINVAL; }
    WideEpi e = {};
    e.y0 = y0; e.y_next = y_next; e.n_prev = n_prev; e.rtol = rtol; e.atol = atol; e.partials = static_cast<double *>(d_ws);
    e.c_dev = c_dev;
    e.y1 = (opt && opt->y1) ? opt->y1 : X;
    e.y_aux = (mode == WIDE_COMBINE && !c_dev && opt && opt->y_aux && opt->c_aux) ? opt->y_aux : nullptr;
    for (int m = 0; m <= kWideMaxPrev; ++m) e.c2[m] = (e.y_aux && m <= n_prev) ? opt->c_aux[m] : 0.f;
    for (int m = 0; m < kWideMaxPrev; ++m) e.kprev[m] = (m < n_prev && h_kprev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kWideMaxPrev; ++m) e.c[m] = (mode != WIDE_RK4 && m <= n_prev) ? h_c[m] : 0.f;
    if (mode == WIDE_RK4) e.c[0] = h_c[0];
    const double P = 4.0 * 256 * (double)A->n_rows;
    ProfScope prof(PROF_RHS_FUSED, st, 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * 256 * (double)(A->n_rows + A->n_cols) + P * (n_prev + 2),
                   2.0 * A->nnz * 256);
    int np = 0, rc;
    if (mode == WIDE_COMBINE) rc = launch_wide<1, WIDE_COMBINE>(A, X, Xh, n_own, K, 1.f, flags, st, &e, &np);
    else if (mode == WIDE_ERROR) rc = launch_wide<1, WIDE_ERROR>(A, X, Xh, n_own, K, 1.f, flags, st, &e, &np);
    else rc = launch_wide<1, WIDE_RK4>(A, X, Xh, n_own, K, 1.f, flags, st, &e, &np);
    if (rc) return rc;
    if (mode == WIDE_ERROR) return partials_finish(e.partials, np, d_out, st, (opt && opt->accum) ? 1 : 0);
    return NDCN_OK;
}

template <int VW, int LPR>
static int launch_spmm(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, int H,
                       float alpha, uint32_t flags, hipStream_t st) {
    const int n_rows = (int)A->n_rows;
    constexpr int groups = kSpmmThreads / LPR;
    // rows per workgroup: enough rows that every lane group gets >= 8 of them when rows are short,
    // bounded so the block's non-zeros usually fit the LDS stage
    int rpb = groups * 8;
    if (rpb < 64) rpb = 64;
    if (rpb > kMaxRowsPerBlock) rpb = kMaxRowsPerBlock;
    const double avg = n_rows > 0 ? (double)A->nnz / (double)n_rows : 0.0;
    while (rpb > groups && rpb > 16 && avg * rpb > kStageCap) rpb /= 2;
    const int nblk = (n_rows + rpb - 1) / rpb;
    if (nblk == 0) return NDCN_OK;
    const int relu = (flags & NDCN_F_RELU) ? 1 : 0;
    if (Xh)
        hipLaunchKernelGGL((spmm_csr_kernel<VW, LPR, true>), dim3(nblk), dim3(kSpmmThreads), 0, st, A->rowptr,
                           A->colidx, A->val, X, Xh, (int)n_own, Y, n_rows, H, alpha, relu, rpb);
    else
        hipLaunchKernelGGL((spmm_csr_kernel<VW, LPR, false>), dim3(nblk), dim3(kSpmmThreads), 0, st, A->rowptr,
                           A->colidx, A->val, X, Xh, (int)n_own, Y, n_rows, H, alpha, relu, rpb);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

template <int VW>
static int dispatch_lpr(int slots, const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y,
                        int H, float alpha, uint32_t flags, hipStream_t st) {
    if (slots > 32) return launch_spmm<VW, 64>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    if (slots > 16) return launch_spmm<VW, 32>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    if (slots > 8) return launch_spmm<VW, 16>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    if (slots > 4) return launch_spmm<VW, 8>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    if (slots > 2) return launch_spmm<VW, 4>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    if (slots > 1) return launch_spmm<VW, 2>(A, X, Xh, n_own, Y, H, alpha, flags, st);
    return launch_spmm<VW, 1>(A, X, Xh, n_own, Y, H, alpha, flags, st);
}

int spmm_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, int H, float alpha,
             uint32_t flags, hipStream_t st) {
    const bool vec = (H % 4 == 0) && aligned16(X) && aligned16(Y) && (Xh == nullptr || aligned16(Xh));
    // Long-row plan (struct ndcn_csr): a 3900-entry hub row of a power-law graph is one wave's sequential work in the row
    // kernels - the tail of the launch.  As in the fused right-hand side: the hubs' rows are formed ahead by two small SpMMs
    // (<= 256-entry segments, then their sums in order) and the launch proper runs on the light operator, which reads each
    // hub's finished row as ONE entry of a second panel.
    // Column-sweep plan (struct ndcn_csr; spmm_sweep.hip): operators without locality whose partial sums fit the register files
    if (vec && !Xh && alpha == 1.f && !(flags & NDCN_F_RELU) && spmm_sweep_supported(A, H)) return spmm_sweep_f32(A, X, Y, st);
    static const int use_hub = env_int("NDCN_SPMM_HUB", 1);
    if (use_hub && vec && !Xh && A->hub_n > 0 && A->hub_H == H && A->hub_S && A->hub_Sseg && A->lt_rowptr) {
        ndcn_csr seg = {};
        seg.n_rows = A->hub_nseg; seg.n_cols = A->n_cols; seg.nnz = A->hub_nnz;
        seg.rowptr = A->hub_seg_rowptr; seg.colidx = A->hub_colidx; seg.val = A->hub_val;
        int rc = spmm_f32(&seg, X, nullptr, A->n_cols, A->hub_Sseg, H, 1.f, 0, st);
        if (rc) return rc;
        ndcn_csr cmb = {};
        cmb.n_rows = A->hub_n; cmb.n_cols = A->hub_nseg; cmb.nnz = A->hub_nseg;
        cmb.rowptr = A->hub_cmb_rowptr; cmb.colidx = A->hub_cmb_colidx; cmb.val = A->hub_cmb_val;
        rc = spmm_f32(&cmb, A->hub_Sseg, nullptr, A->hub_nseg, A->hub_S, H, 1.f, 0, st);
        if (rc) return rc;
        ndcn_csr light = {};
        light.n_rows = A->n_rows; light.n_cols = A->n_cols + A->hub_n; light.nnz = A->lt_nnz;
        light.rowptr = A->lt_rowptr; light.colidx = A->lt_colidx; light.val = A->lt_val;
        return spmm_f32(&light, X, A->hub_S, A->n_cols, Y, H, alpha, flags, st);
    }
    if (vec && spmm_rec_supported(A, H) && A->n_rows * (int64_t)1024 < (1ll << 32))     // operator carries a group-record plan
        return spmm_rec_f32(A, X, Xh, n_own, Y, alpha, flags, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f, 0.f, nullptr,
                            nullptr, st);
    ProfScope prof(PROF_SPMM, st, 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * H * (double)(A->n_rows + A->n_cols),
                   2.0 * A->nnz * H);
    if (vec && H % 256 == 0 && H <= 1024) {
        static const int use_wide = env_int("NDCN_SPMM_WIDE", 1);
        if (use_wide) {
            if (H == 256) return launch_wide<1>(A, X, Xh, n_own, Y, alpha, flags, st);
            if (H == 512) return launch_wide<2>(A, X, Xh, n_own, Y, alpha, flags, st);
            if (H == 768) return launch_wide<3>(A, X, Xh, n_own, Y, alpha, flags, st);
            return launch_wide<4>(A, X, Xh, n_own, Y, alpha, flags, st);
        }
    }
    if (vec) return dispatch_lpr<4>(H / 4, A, X, Xh, n_own, Y, H, alpha, flags, st);
    return dispatch_lpr<1>(H, A, X, Xh, n_own, Y, H, alpha, flags, st);
}

// ---------------------------------------------------------------------------------------------------
// out[i, :] = X[idx[i], :]   (halo send-buffer packing)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ X, const int *__restrict__ idx,
                                                          int64_t n_idx, int H, float *__restrict__ out) {
    const int64_t total = n_idx * (int64_t)H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / H;
        const int c = (int)(i - r * H);
        out[i] = X[(size_t)idx[r] * H + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows4_kernel(const float4 *__restrict__ X, const int *__restrict__ idx,
                                                           int64_t n_idx, int H4, float4 *__restrict__ out) {
    const int64_t total = n_idx * (int64_t)H4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / H4;
        const int c = (int)(i - r * H4);
        out[i] = X[(size_t)idx[r] * H4 + c];
    }
}

int gather_rows_f32(const float *X, const int32_t *idx, int64_t n_idx, int H, float *out, hipStream_t st) {
    if (n_idx == 0) return NDCN_OK;
    ProfScope prof(PROF_GATHER, st, 8.0 * n_idx * H + 4.0 * n_idx, 0.0);
    if (H % 4 == 0 && aligned16(X) && aligned16(out)) {
        const int64_t total = n_idx * (H / 4);
        hipLaunchKernelGGL(gather_rows4_kernel, dim3(stream_grid_full(total, 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(X), idx, n_idx, H / 4, reinterpret_cast<float4 *>(out));
    } else {
        const int64_t total = n_idx * (int64_t)H;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, X, idx, n_idx, H, out);
    }
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

}  // namespace ndcn
