// A WHOLE fixed-grid solve in one launch for states that fit one CU: FixedGridODESolver.integrate (solvers.py:79-99) with
// Euler / midpoint / RK4-3/8 steps (fixed_grid.py:7-29, rk_common.py:72-78) over the ODEFunc right-hand side
// (neural_dynamics.py:27-36), all ticks of the caller's time grid, and its reverse sweep for training.
//
// The reference's own commands run N = 400 nodes at H = 20 (heat_dynamics.py:33: a 32 KB state) for 80-120 Euler steps:
// one launch per step (rhs_small.hip) is 10 us of launch latency around 2 us of work.  Here ONE workgroup of 16 waves owns
// the solve: the stage input lives in LDS (every row's neighbours read it), W^T and the CSR arrays too, and everything
// row-local - the state y, the stages k1..k3 - stays in the registers of the thread that owns the element, across all
// steps.  Two workgroup barriers per right-hand side; the only global traffic is the tick output.
//
//   element (row r, column o) -> wave w = (r / RPW) % 16, pass it = (r / RPW) / 16, lane = (r % RPW) * H + o,
//   RPW = 64 / H rows share a wave pass (H = 20: 3 rows = 60 lanes)
//   gather   s = sum_j val_j T[col_j][o]: one fma per entry in stored order       (the chain of spmm_csr / rhs_small)
//   linear   K[o] = relu(b[o] + sum_h S[h] W[o][h]): fma chain over h ascending    (the chain of linear_f32 / rhs_small)
//   stages   the reference's operator order, products and sums rounded separately (rk.hip: stage1<OP>)
// => every tick is bit-identical to the per-step path (rhs_small + fixed_stage kernels): tests compare with torch.equal.
//
// Backward (Euler: the drivers' default method, heat_dynamics.py:20-22,313-334): the same workgroup walks the ticks in
// reverse, re-forming S_i = A y_i and the ReLU mask from the stored trajectory (the forward's own output) and
// accumulating  g_W += gZ^T S,  g_b += sum gZ,  a <- a + g_out[i] + A^T (gZ W)  with gZ = dt_i a (.) [K_i > 0].
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "kernels.h"

#pragma clang fp contract(off)

namespace ndcn {

namespace {

constexpr int kWaves = 16;
constexpr int kChunk = 128;                 // ticks per launch (the step sizes ride in the kernel arguments)
constexpr size_t kLdsMax = 160 * 1024;

// Workgroup barrier for LDS hand-overs: wait for this wave's LDS operations only.  __syncthreads() also drains the vector-memory
// counter, i.e. every barrier after a tick's output store would wait a global-memory round trip (~2 us) for nothing.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s += sum_j val_j T[col_j * H + o] over the cnt entries from j0, one fma per entry in stored order; the index / value / row
// fetches of B entries are issued together (two LDS round trips per B entries instead of two per entry)
template <int B>
__device__ __forceinline__ float gather_row(const int *ci, const float *va, const float *T, int j0, int cnt, int H, int o, float s) {
    for (int jb = 0; __any(jb < cnt); jb += B) {
        int c[B];
        float v[B], x[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const bool ok = jb + u < cnt;
            const int idx = ok ? j0 + jb + u : 0;
            c[u] = ci[idx];
            v[u] = va[idx];
        }
#pragma unroll
        for (int u = 0; u < B; ++u) x[u] = T[(jb + u < cnt ? c[u] : 0) * H + o];
#pragma unroll
        for (int u = 0; u < B; ++u)
            if (jb + u < cnt) s = fmaf(v[u], x[u], s);
    }
    return s;
}

struct SolveArgs {
    const int *rowptr, *colidx;
    const float *val;
    const float *W, *bias;
    const float *y0;                 // [n_rows][H] state at the start of this launch
    float *out;                      // [n_ticks][n_rows][H]
    int n_rows, H, nnz, n_ticks, relu, no_graph, no_control, csr_in_lds;
    int width;                       // fast path: entries per ELL row (the operator's longest row)
    long long *dbg;                  // NDCN_SS_DEBUG=1: {shader cycles total, in evaluations, in stage updates, 100 MHz ticks total}
    float *keep;                     // nullable (Euler, fast path): [n_ticks][2][n] - S = A y_i and K_i = f(y_i) of every step, for the reverse sweep
    float dt[kChunk];
};

struct Lds {
    float *T, *wt, *srow, *val;
    int *rowptr, *colidx;
};

__device__ __forceinline__ Lds carve(float *base, int n_elem, int H, int n_rows, int nnz, bool csr) {
    Lds l;
    l.T = base;
    l.wt = l.T + n_elem;
    l.srow = l.wt + H * (H + 1);
    float *p = l.srow + kWaves * 64;
    l.rowptr = reinterpret_cast<int *>(p);
    l.colidx = l.rowptr + (csr ? n_rows + 1 : 0);
    l.val = reinterpret_cast<float *>(l.colidx + (csr ? nnz : 0));
    return l;
}

inline size_t lds_bytes(int64_t n_elem, int H, int64_t n_rows, int64_t nnz, bool csr, int64_t extra_floats = 0) {
    return sizeof(float) * (size_t)(n_elem + H * (H + 1) + kWaves * 64 + extra_floats + (csr ? (n_rows + 1) + 2 * nnz : 0));
}

// K for the element this lane owns in pass `it` (valid lanes only), from the stage input in l.T
template <int GB>
__device__ __forceinline__ float eval_rhs(const SolveArgs &a, const Lds &l, const int *rp, const int *ci, const float *va, int r, int q,
                                          int o, int lane, int wave, bool valid, float bias_o) {
    const int H = a.H;
    float s = 0.f;
    if (a.no_graph) {
        if (valid) s = l.T[r * H + o];
    } else {
        int j0 = 0, cnt = 0;
        if (valid) { j0 = rp[r]; cnt = rp[r + 1] - j0; }
        s = gather_row<GB>(ci, va, l.T, j0, cnt, H, o, s);
    }
    if (a.no_control) return a.relu ? relu_nan(s) : s;
    float *srow = l.srow + wave * 64;
    __builtin_amdgcn_wave_barrier();
    if (valid) srow[lane] = s;
    __builtin_amdgcn_wave_barrier();
    float k = 0.f;
    if (valid) {
        const float *sr = srow + q * H;
        const int ldw = H + 1;
#pragma unroll 4
        for (int h = 0; h < H; ++h) k = fmaf(sr[h], l.wt[h * ldw + o], k);
        k = k + bias_o;
        if (a.relu) k = relu_nan(k);
    }
    return k;
}

// The README shape on the fast path (HT = H at compile time; the plain ODEFunc: graph and control term, CSR in LDS):
//   * the operator's entries are packed as {column * H, value bits} pairs - one 8-byte LDS read per entry, the row address is
//     one add away;
//   * lane o keeps row o of W (its H weights) in registers for the whole solve: the Linear is H/4 16-byte reads of the wave's S
//     row + H fmas, no weight traffic;
// the same fma chains in the same order as the generic path (bit-identical), ~1/3 of its instructions and LDS bytes.
// The fast path keeps the operator in ELL form in LDS: every row padded to `width` (the longest row, <= 16) entries
// {column * H, value bits}, the padding {offset of an all-zero row, 0.0f}.  A row's gather is then `width` times {one 8-byte LDS read
// at a constant offset, one add, one 4-byte LDS read, one fma} - no row extents, no predicates, no loop over rounds: a third of
// the CSR form's instructions, and the solve is bound by instruction issue (one CU: 64 lanes per cycle).  The padding adds
// 0.0f * 0.0f to a sum that started at +0.0f: the bits of every sum are those of the CSR chain.
// (NP passes of a wave could be evaluated together - independent chains of LDS round trips - but measured, that buys nothing: 16
// waves share 4 SIMDs and the ~1000 instructions a wave issues per Euler step already keep every SIMD busy: 19 k cycles per step
// with one pass at a time, 23 k with three in a rolled double-buffered form.  What is left is instruction count, not latency.)
// One ELL row's gather, entries THREE at a time (the host pads the ELL width to a multiple of three with {zero row, 0.0f} entries - adds
// of +0 x 0 - so there is no tail): the three {offset, value} pairs are requested together, then the three panel values, then the fmas
// in stored order.  The rolled form (one entry per trip, `#pragma unroll 3`) compiled to two dependent LDS round trips PER ENTRY with a
// full wait behind each (llvm does not hoist the second entry's reads over the first's fma): a wave's chain of 9 passes x 9 entries x 2
// round trips was the floor of every phase that gathers (round 5: the disassembly of the reverse sweep).
__device__ __forceinline__ float ell_gather3(const int2 *row, int width3, const float *T, int o) {
    float s = 0.f;
    for (int j = 0; j < width3; j += 3) {
        const int2 e0 = row[j], e1 = row[j + 1], e2 = row[j + 2];
        const float x0 = T[e0.x + o], x1 = T[e1.x + o], x2 = T[e2.x + o];
        s = fmaf(__int_as_float(e0.y), x0, s);
        s = fmaf(__int_as_float(e1.y), x1, s);
        s = fmaf(__int_as_float(e2.y), x2, s);
    }
    return s;
}

template <int HT, int NP>
__device__ __forceinline__ void eval_fast(const int2 *ell, int width, const float *T, float *srow, const float (&wreg)[HT ? HT : 1],
                                          const int (&r)[NP], const bool (&valid)[NP], int q, int o, int lane, float bias_o, int relu,
                                          float (&out)[NP], float (&s)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        s[p] = ell_gather3(ell + (valid[p] ? r[p] : 0) * width, width, T, o);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NP; ++p) srow[64 * p + lane] = s[p];
    __builtin_amdgcn_wave_barrier();
    float k[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) k[p] = 0.f;
#pragma unroll
    for (int h4 = 0; h4 < HT / 4; ++h4) {
        float4 sv[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) sv[p] = reinterpret_cast<const float4 *>(srow + 64 * p + (valid[p] ? q : 0) * HT)[h4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            k[p] = fmaf(sv[p].x, wreg[4 * h4], k[p]);
            k[p] = fmaf(sv[p].y, wreg[4 * h4 + 1], k[p]);
            k[p] = fmaf(sv[p].z, wreg[4 * h4 + 2], k[p]);
            k[p] = fmaf(sv[p].w, wreg[4 * h4 + 3], k[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const float kk = k[p] + bias_o;
        out[p] = relu ? relu_nan(kk) : kk;
    }
}

// ELL image of a CSR in LDS (one thread per row); zero_off: offset (in floats, relative to the gather's panel) of H zeros
__device__ __forceinline__ void build_ell(int2 *ell, int width, int n_rows, int H, const int *rowptr, const int *colidx, const float *val,
                                          int zero_off, int tid) {
    for (int r = tid; r < n_rows; r += 1024) {
        const int j0 = rowptr[r], cnt = rowptr[r + 1] - j0;
        for (int j = 0; j < width; ++j)
            ell[r * width + j] = j < cnt ? make_int2(colidx[j0 + j] * H, __float_as_int(val[j0 + j])) : make_int2(zero_off, 0);
    }
}

constexpr int kFastNp = 1;                  // passes evaluated together (srow: kFastNp x 64 floats per wave)

// [ELL entries | panels (n_elem + extra) | srow | zero row]
inline size_t lds_bytes_fast(int64_t n_elem, int64_t n_rows, int64_t width, int H, int64_t extra_floats = 0) {
    return 8 * (size_t)(n_rows * width) + sizeof(float) * (size_t)(n_elem + extra_floats + kWaves * 64 * kFastNp + H);
}

// METHOD: NDCN_M_EULER / MIDPOINT / RK4.  MAXIT: passes a wave makes over its rows (register arrays are indexed by pass)
template <int METHOD, int MAXIT, bool CSR_LDS, int HT, bool KEEP = false>
__global__ __launch_bounds__(1024) void solve_small_kernel(SolveArgs a) {
    extern __shared__ float lds_raw[];
    const int H = HT ? HT : a.H, n_elem = a.n_rows * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RPW = 64 / H;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    // ---- stage: the weights (W^T in LDS, or this lane's row of W in registers), the CSR arrays, the initial state
    Lds l;
    int2 *f_ent = nullptr;
    float *f_srow = nullptr;
    float wreg[HT ? HT : 1];
    if (HT) {                                               // fast path: [entries | T | srow | rowptr]
        f_ent = reinterpret_cast<int2 *>(lds_raw);
        l.T = lds_raw + 2 * a.n_rows * a.width;
        f_srow = l.T + n_elem + wave * 64 * kFastNp;
        float *zrow = l.T + n_elem + kWaves * 64 * kFastNp;
        l.wt = l.srow = l.val = nullptr;
        l.rowptr = l.colidx = nullptr;
        for (int i = tid; i < H; i += 1024) zrow[i] = 0.f;
        build_ell(f_ent, a.width, a.n_rows, H, a.rowptr, a.colidx, a.val, (int)(zrow - l.T), tid);
#pragma unroll
        for (int h = 0; h < (HT ? HT : 1); ++h) wreg[h] = lane_on ? a.W[o * H + h] : 0.f;
    } else {
        l = carve(lds_raw, n_elem, H, a.n_rows, a.nnz, CSR_LDS);
        if (!a.no_control)
            for (int i = tid; i < H * H; i += 1024) {
                const int oo = i / H, h = i - oo * H;
                l.wt[h * (H + 1) + oo] = a.W[i];
            }
        if (CSR_LDS) {
            for (int i = tid; i <= a.n_rows; i += 1024) l.rowptr[i] = a.rowptr[i];
            for (int i = tid; i < a.nnz; i += 1024) { l.colidx[i] = a.colidx[i]; l.val[i] = a.val[i]; }
        }
    }
    const int *rp = CSR_LDS ? l.rowptr : a.rowptr;
    const int *ci = CSR_LDS ? l.colidx : a.colidx;
    const float *va = CSR_LDS ? l.val : a.val;
    const float bias_o = (!a.no_control && a.bias && lane_on) ? a.bias[o] : 0.f;
    float y[MAXIT], k1[MAXIT], k2[METHOD == NDCN_M_RK4 ? MAXIT : 1], k3[METHOD == NDCN_M_RK4 ? MAXIT : 1];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int r = (it * kWaves + wave) * RPW + q;
        const bool valid = lane_on && r < a.n_rows;
        y[it] = valid ? a.y0[r * H + o] : 0.f;
        if (valid) l.T[r * H + o] = y[it];
        k1[it] = 0.f;
    }
    lds_barrier();
    // one right-hand side over the stage input in T -> dst[it]; then T <- the next stage input
#define NDCN_EVAL(dst)                                                                                         \
    if (HT) {                                                                                                  \
        constexpr int NPF = 1;                                                                                 \
        _Pragma("unroll") for (int it = 0; it < MAXIT; it += NPF) {                                            \
            if ((it * kWaves + wave) * RPW < a.n_rows) {                                                       \
                int rr_[NPF];                                                                                  \
                bool vv_[NPF];                                                                                 \
                float oo_[NPF], ss_[NPF];                                                                      \
                _Pragma("unroll") for (int p = 0; p < NPF; ++p) {                                              \
                    rr_[p] = ((it + p) * kWaves + wave) * RPW + q;                                             \
                    vv_[p] = lane_on && rr_[p] < a.n_rows;                                                     \
                }                                                                                              \
                eval_fast<HT, NPF>(f_ent, a.width, l.T, f_srow, wreg, rr_, vv_, q, o, lane, bias_o, a.relu, oo_, ss_);  \
                _Pragma("unroll") for (int p = 0; p < NPF; ++p) {                                              \
                    dst[it + p] = oo_[p];                                                                      \
                    if (KEEP && METHOD == NDCN_M_EULER && vv_[p]) {                                            \
                        float *kp_ = a.keep + (size_t)tick * 2 * n_elem + rr_[p] * H + o;                      \
                        kp_[0] = ss_[p];                                                                       \
                        kp_[n_elem] = oo_[p];                                                                  \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    } else {                                                                                                   \
        _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                                 \
            const int r = (it * kWaves + wave) * RPW + q;                                                      \
            if ((it * kWaves + wave) * RPW < a.n_rows)                                                         \
                dst[it] = eval_rhs<(MAXIT > 4 ? 4 : 8)>(a, l, rp, ci, va, r, q, o, lane, wave, lane_on && r < a.n_rows, bias_o); \
        }                                                                                                      \
    }                                                                                                          \
    lds_barrier();
#define NDCN_PUT(expr, also_out)                                                                               \
    _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                                     \
        const int r = (it * kWaves + wave) * RPW + q;                                                          \
        if (lane_on && r < a.n_rows) {                                                                         \
            const float v_ = (expr);                                                                           \
            l.T[r * H + o] = v_;                                                                               \
            if (also_out) { y[it] = v_; a.out[(size_t)tick * n_elem + r * H + o] = v_; }                       \
        }                                                                                                      \
    }                                                                                                          \
    lds_barrier();
    long long c_eval = 0, c_put = 0, c0 = 0, w0 = 0;
    if (a.dbg) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
#define NDCN_T0 long long t_ = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
#define NDCN_T1(acc) if (a.dbg) { const long long n_ = __builtin_readcyclecounter(); acc += n_ - t_; t_ = n_; }
    for (int tick = 0; tick < a.n_ticks; ++tick) {
        const float dt = a.dt[tick];
        NDCN_T0
        if (METHOD == NDCN_M_EULER) {
            NDCN_EVAL(k1)
            NDCN_T1(c_eval)
            NDCN_PUT(y[it] + dt * k1[it], true)                                              // fixed_grid.py:8 + solvers.py:92
            NDCN_T1(c_put)
        } else if (METHOD == NDCN_M_MIDPOINT) {
            NDCN_EVAL(k1)
            NDCN_PUT(y[it] + k1[it] * dt / 2.f, false)                                       // fixed_grid.py:18
            NDCN_EVAL(k1)
            NDCN_PUT(y[it] + dt * k1[it], true)                                              // fixed_grid.py:19 + solvers.py:92
        } else {
            NDCN_EVAL(k1)
            NDCN_PUT(y[it] + dt * k1[it] / 3.f, false)                                       // rk_common.py:75
            NDCN_EVAL(k2)
            NDCN_PUT(y[it] + dt * (k1[it] / -3.f + k2[it]), false)                           // rk_common.py:76
            NDCN_EVAL(k3)
            NDCN_PUT(y[it] + dt * (k1[it] - k2[it] + k3[it]), false)                         // rk_common.py:77
            // the sum's first three terms, left to right as the reference adds them; k4 joins last
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) k1[it] = k1[it] + 3.f * k2[it] + 3.f * k3[it];
            NDCN_EVAL(k2)
            NDCN_PUT(y[it] + (k1[it] + k2[it]) * (dt / 8.f), true)                           // rk_common.py:78 + solvers.py:92
        }
    }
    if (a.dbg && tid == 0) {
        a.dbg[0] = (long long)__builtin_readcyclecounter() - c0;
        a.dbg[1] = c_eval;
        a.dbg[2] = c_put;
        a.dbg[3] = (long long)wall_clock64() - w0;
    }
#undef NDCN_T0
#undef NDCN_T1
#undef NDCN_EVAL
#undef NDCN_PUT
}

// ------------------------------------------------------------------------------------------------ Euler reverse sweep

struct BwdArgs {
    const int *rowptr, *colidx;      // A
    const float *val;
    const int *t_rowptr, *t_colidx;  // A^T
    const float *t_val;
    const float *W, *bias;
    const float *traj;               // [n_ticks + 1][n]: y_0 .. y_T (the forward's output, y_0 first)
    const float *g_out;              // [n_ticks + 1][n]: dL/dy_i (zeros where a tick carries no loss)
    float *g_y0;                     // [n]
    float *g_W, *g_b;                // [H][H], [H]  (added to what the caller zeroed: a chunked solve accumulates)
    long long *dbg;                  // NDCN_SS_DEBUG=1: shader cycles {total, forward pieces, g_W + gS, transposed gather + update}
    const float *a_in;               // nullable [n]: the adjoint at the LAST tick of this launch, handed over by the launch that
                                     // covered the later ticks (it already holds that tick's g_out); NULL: g_out[n_ticks] itself
    int n_rows, H, nnz, n_ticks, relu, no_graph, no_control;
    int width;                       // fast path: entries per ELL row
    const float *keep;               // nullable (fast path): what the forward launch kept - [n_ticks][2][n], S_i and K_i
    float dt[kChunk];
};

// LDS: T [n] (y_i, then gS = gZ W),  S [n] (A y_i, then A^T gS),  Z [n] (dt a, then gZ),  wt (W^T, padded rows), srow, and - when
// they fit - the CSR arrays of A (shared by the transposed gather when the operator is symmetric: normalised Laplacians are).
// Registers: the adjoint a = dL/dy_{i+1} of the elements this thread owns (one per pass), the prefetched y_{i-1}.
template <int MAXIT, bool CSR_LDS>
__global__ __launch_bounds__(1024) void solve_small_bwd_kernel(BwdArgs b, int symmetric) {
    extern __shared__ float lds_raw[];
    const int H = b.H, n_elem = b.n_rows * H, ldw = H + 1;
    float *T = lds_raw, *S = T + n_elem, *Z = S + n_elem, *wt = Z + n_elem, *srow_all = wt + H * ldw;
    int *l_rp = reinterpret_cast<int *>(srow_all + kWaves * 64), *l_ci = l_rp + (CSR_LDS ? b.n_rows + 1 : 0);
    float *l_va = reinterpret_cast<float *>(l_ci + (CSR_LDS ? b.nnz : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RPW = 64 / H;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    // the element this thread owns in pass `it`: e0 + it * estride (valid while < n_elem on an active lane)
    const int e0 = lane_on ? (wave * RPW + q) * H + o : n_elem, estride = kWaves * RPW * H;
    if (!b.no_control)
        for (int i = tid; i < H * H; i += 1024) {
            const int oo = i / H, h = i - oo * H;
            wt[h * ldw + oo] = b.W[i];
        }
    if (CSR_LDS) {
        for (int i = tid; i <= b.n_rows; i += 1024) l_rp[i] = b.rowptr[i];
        for (int i = tid; i < b.nnz; i += 1024) { l_ci[i] = b.colidx[i]; l_va[i] = b.val[i]; }
    }
    const int *rp = CSR_LDS ? l_rp : b.rowptr, *ci = CSR_LDS ? l_ci : b.colidx;
    const float *va = CSR_LDS ? l_va : b.val;
    const int *trp = symmetric ? rp : b.t_rowptr, *tci = symmetric ? ci : b.t_colidx;
    const float *tva = symmetric ? va : b.t_val;
    const float bias_o = (!b.no_control && b.bias && lane_on) ? b.bias[o] : 0.f;
    float adj[MAXIT];
    {
        const float *a0 = b.a_in ? b.a_in : b.g_out + (size_t)b.n_ticks * n_elem;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            adj[it] = e < n_elem ? a0[e] : 0.f;
        }
    }
    // y_i arrives through registers, requested a step ahead (MAXIT x 1024 elements cover the state: <= MAXIT passes of 1024 lanes)
    float pre[MAXIT];
    {
        const float *yl = b.traj + (size_t)(b.n_ticks - 1) * n_elem;
#pragma unroll
        for (int u = 0; u < MAXIT; ++u) {
            const int e = tid + 1024 * u;
            pre[u] = e < n_elem ? yl[e] : 0.f;
        }
    }
    // g_W: thread t < H*H owns entry (t / H, t % H) over the even half of the rows, thread H*H + t the other half; g_b likewise
    const int HH = H * H;
    const bool gw_on = !b.no_control && tid < 2 * HH, gb_on = !b.no_control && tid >= 2 * HH && tid < 2 * HH + 2 * H;
    const int half = gw_on ? tid / HH : gb_on ? (tid - 2 * HH) / H : 0;
    const int gw_o = gw_on ? (tid % HH) / H : gb_on ? (tid - 2 * HH) % H : 0, gw_h = gw_on ? tid % H : 0;
    const int r_lo = half ? b.n_rows / 2 : 0, r_hi = half ? b.n_rows : b.n_rows / 2;
    float gacc = 0.f;
    long long c_a = 0, c_b = 0, c_c = 0, c0 = b.dbg ? (long long)__builtin_readcyclecounter() : 0;
    lds_barrier();
    for (int i = b.n_ticks - 1; i >= 0; --i) {
        const float dt = b.dt[i];
        long long t_ = b.dbg ? (long long)__builtin_readcyclecounter() : 0;
        // ---- T <- y_i (prefetched), request y_{i-1};  Z <- dt a
#pragma unroll
        for (int u = 0; u < MAXIT; ++u) {
            const int e = tid + 1024 * u;
            if (e < n_elem) T[e] = pre[u];
        }
        if (i > 0) {
            const float *yp = b.traj + (size_t)(i - 1) * n_elem;
#pragma unroll
            for (int u = 0; u < MAXIT; ++u) {
                const int e = tid + 1024 * u;
                if (e < n_elem) pre[u] = yp[e];
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            if (e < n_elem) Z[e] = dt * adj[it];
        }
        lds_barrier();
        // ---- forward pieces at y_i: S = A y_i, K = relu(W S + b); gZ = dt a (.) [K > 0]  (the mask of relu's output)
#pragma nounroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((it * kWaves + wave) * RPW >= b.n_rows) break;
            const int r = (it * kWaves + wave) * RPW + q;
            const bool valid = lane_on && r < b.n_rows;
            float s = 0.f;
            if (b.no_graph) {
                if (valid) s = T[r * H + o];
            } else {
                int j0 = 0, cnt = 0;
                if (valid) { j0 = rp[r]; cnt = rp[r + 1] - j0; }
                s = gather_row<4>(ci, va, T, j0, cnt, H, o, s);
            }
            float k = s;
            if (!b.no_control) {
                float *srow = srow_all + wave * 64;
                __builtin_amdgcn_wave_barrier();
                if (valid) srow[lane] = s;
                __builtin_amdgcn_wave_barrier();
                k = 0.f;
                if (valid) {
                    const float *sr = srow + q * H;
#pragma unroll 4
                    for (int h = 0; h < H; ++h) k = fmaf(sr[h], wt[h * ldw + o], k);
                    k = k + bias_o;
                }
            }
            if (valid) {
                S[r * H + o] = s;
                if (b.relu && !(k > 0.f)) Z[r * H + o] = 0.f;
            }
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_a += n_ - t_; t_ = n_; }
        // ---- g_W[oo][h] += sum_r gZ[r][oo] S[r][h];  g_b[oo] += sum_r gZ[r][oo];  gS = gZ W -> T (y_i is no longer needed)
        if (gw_on) {
            float acc = 0.f;
#pragma unroll 4
            for (int r = r_lo; r < r_hi; ++r) acc = fmaf(Z[r * H + gw_o], S[r * H + gw_h], acc);
            gacc += acc;
        } else if (gb_on) {
            float acc = 0.f;
#pragma unroll 4
            for (int r = r_lo; r < r_hi; ++r) acc += Z[r * H + gw_o];
            gacc += acc;
        }
#pragma nounroll
        for (int e = tid; e < n_elem; e += 1024) {
            float gs;
            if (b.no_control) gs = Z[e];
            else {
                const int r = e / H, h = e - r * H;
                gs = 0.f;
#pragma unroll 4
                for (int oo = 0; oo < H; ++oo) gs = fmaf(Z[r * H + oo], wt[h * ldw + oo], gs);
            }
            T[e] = gs;
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_b += n_ - t_; t_ = n_; }
        // ---- S <- A^T gS (each element by the thread that owns it); then a <- (a + S) + g_out[i]
#pragma nounroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((it * kWaves + wave) * RPW >= b.n_rows) break;
            const int r = (it * kWaves + wave) * RPW + q;
            const bool valid = lane_on && r < b.n_rows;
            float s = 0.f;
            if (b.no_graph) {
                if (valid) s = T[r * H + o];
            } else {
                int j0 = 0, cnt = 0;
                if (valid) { j0 = trp[r]; cnt = trp[r + 1] - j0; }
                s = gather_row<4>(tci, tva, T, j0, cnt, H, o, s);
            }
            if (valid) S[r * H + o] = s;
        }
        {
            const float *gi = b.g_out + (size_t)i * n_elem;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                if (e < n_elem) adj[it] = (adj[it] + S[e]) + gi[e];
            }
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_c += n_ - t_; t_ = n_; }
    }
    if (b.dbg && tid == 0) {
        b.dbg[0] = (long long)__builtin_readcyclecounter() - c0;
        b.dbg[1] = c_a; b.dbg[2] = c_b; b.dbg[3] = c_c;
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = e0 + it * estride;
        if (e < n_elem) b.g_y0[e] = adj[it];
    }
    // the two halves of every g_W / g_b entry meet in LDS
    if (!b.no_control) {
        if ((gw_on || gb_on) && half) T[gw_on ? tid - HH : HH + gw_o] = gacc;
        lds_barrier();
        if (gw_on && !half) b.g_W[tid] += gacc + T[tid];
        else if (gb_on && !half) b.g_b[gw_o] += gacc + T[HH + gw_o];
    }
}

// ------------------------------------------------------------------------------------------------ midpoint / RK4 reverse sweep
// The same workgroup, the same LDS layout [T | S | Z | W^T | srow | CSR] as the Euler sweep above; per step, in reverse:
//   the stage inputs are re-formed from y_i by the forward kernel's formulas (fixed_grid.py:18-19, rk_common.py:75-77; the stage
//   derivatives stay in the registers of the thread that owns the element), then the stages' vector-Jacobian products in reverse -
//   each the Euler sweep's block (S = A u, mask, g_W / g_b, gS = gZ W, A^T gS) - with the recurrences of `_FixedGridSolve.backward`
//   (_impl/odeint.py) kept as running prefixes in registers:  RK4 3/8:  g_k3 = 3c a + dt g_u4;  g_k2 = 3c a - dt g_u4 + dt g_u3;
//   g_k1 = c a + dt g_u4 - (dt/3) g_u3 + (dt/3) g_u2, c = dt / 8;  a <- a + g_u4 + g_u3 + g_u2 + g_u1 + g_out[i].
template <int METHOD, int MAXIT, bool CSR_LDS>
__global__ __launch_bounds__(1024) void solve_small_bwd_rk_kernel(BwdArgs b, int symmetric) {
    extern __shared__ float lds_raw[];
    const int H = b.H, n_elem = b.n_rows * H, ldw = H + 1;
    float *T = lds_raw, *S = T + n_elem, *Z = S + n_elem, *wt = Z + n_elem, *srow_all = wt + H * ldw;
    int *l_rp = reinterpret_cast<int *>(srow_all + kWaves * 64), *l_ci = l_rp + (CSR_LDS ? b.n_rows + 1 : 0);
    float *l_va = reinterpret_cast<float *>(l_ci + (CSR_LDS ? b.nnz : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RPW = 64 / H;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    const int e0 = lane_on ? (wave * RPW + q) * H + o : n_elem, estride = kWaves * RPW * H;
    if (!b.no_control)
        for (int i = tid; i < H * H; i += 1024) {
            const int oo = i / H, h = i - oo * H;
            wt[h * ldw + oo] = b.W[i];
        }
    if (CSR_LDS) {
        for (int i = tid; i <= b.n_rows; i += 1024) l_rp[i] = b.rowptr[i];
        for (int i = tid; i < b.nnz; i += 1024) { l_ci[i] = b.colidx[i]; l_va[i] = b.val[i]; }
    }
    const int *rp = CSR_LDS ? l_rp : b.rowptr, *ci = CSR_LDS ? l_ci : b.colidx;
    const float *va = CSR_LDS ? l_va : b.val;
    const int *trp = symmetric ? rp : b.t_rowptr, *tci = symmetric ? ci : b.t_colidx;
    const float *tva = symmetric ? va : b.t_val;
    const float bias_o = (!b.no_control && b.bias && lane_on) ? b.bias[o] : 0.f;
    float adj[MAXIT];
    {
        const float *a0 = b.a_in ? b.a_in : b.g_out + (size_t)b.n_ticks * n_elem;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            adj[it] = e < n_elem ? a0[e] : 0.f;
        }
    }
    const int HH = H * H;
    const bool gw_on = !b.no_control && tid < 2 * HH, gb_on = !b.no_control && tid >= 2 * HH && tid < 2 * HH + 2 * H;
    const int half = gw_on ? tid / HH : gb_on ? (tid - 2 * HH) / H : 0;
    const int gw_o = gw_on ? (tid % HH) / H : gb_on ? (tid - 2 * HH) % H : 0, gw_h = gw_on ? tid % H : 0;
    const int r_lo = half ? b.n_rows / 2 : 0, r_hi = half ? b.n_rows : b.n_rows / 2;
    float gacc = 0.f;
    lds_barrier();

    // S[r][o] = (A T)[r][o] (or T itself) for the rows of pass `it`; returns the value of this lane's element
    auto gather = [&](int it, const int *prp, const int *pci, const float *pva, bool &valid, int &r) -> float {
        r = (it * kWaves + wave) * RPW + q;
        valid = lane_on && r < b.n_rows;
        float s = 0.f;
        if (b.no_graph) {
            if (valid) s = T[r * H + o];
        } else {
            int j0 = 0, cnt = 0;
            if (valid) { j0 = prp[r]; cnt = prp[r + 1] - j0; }
            s = gather_row<4>(pci, pva, T, j0, cnt, H, o, s);
        }
        return s;
    };
    // pre-activation of the Linear for this lane's element from the wave's S row (through srow)
    auto linear = [&](float s, bool valid) -> float {
        if (b.no_control) return s;
        float *srow = srow_all + wave * 64;
        __builtin_amdgcn_wave_barrier();
        if (valid) srow[lane] = s;
        __builtin_amdgcn_wave_barrier();
        float k = 0.f;
        if (valid) {
            const float *sr = srow + q * H;
#pragma unroll 4
            for (int h = 0; h < H; ++h) k = fmaf(sr[h], wt[h * ldw + o], k);
            k = k + bias_o;
        }
        return k;
    };
#define NDCN_PUT_T(expr)                                                                       \
    _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                     \
        const int e = e0 + it * estride;                                                       \
        if (e < n_elem) T[e] = (expr);                                                         \
    }                                                                                          \
    lds_barrier();
    // K = f(T) for the owned elements
#define NDCN_EVAL_K(dst)                                                                       \
    _Pragma("nounroll") for (int it = 0; it < MAXIT; ++it) {                                   \
        if ((it * kWaves + wave) * RPW >= b.n_rows) break;                                     \
        bool valid; int r;                                                                     \
        const float s_ = gather(it, rp, ci, va, valid, r);                                     \
        const float k_ = linear(s_, valid);                                                    \
        dst[it] = valid ? (b.relu ? relu_nan(k_) : k_) : 0.f;                                  \
    }                                                                                          \
    lds_barrier();
    // g[it] = alpha (J(T)^T z)[owned element];  g_W / g_b += scale * (gZ^T S, sum gZ)   (T is consumed)
#define NDCN_VJP(z, alpha, scale, g)                                                           \
    _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                     \
        const int e = e0 + it * estride;                                                       \
        if (e < n_elem) Z[e] = z[it];                                                          \
    }                                                                                          \
    _Pragma("nounroll") for (int it = 0; it < MAXIT; ++it) {                                   \
        if ((it * kWaves + wave) * RPW >= b.n_rows) break;                                     \
        bool valid; int r;                                                                     \
        const float s_ = gather(it, rp, ci, va, valid, r);                                     \
        const float k_ = linear(s_, valid);                                                    \
        if (valid) {                                                                           \
            S[r * H + o] = s_;                                                                 \
            if (b.relu && !(k_ > 0.f)) Z[r * H + o] = 0.f;                                     \
        }                                                                                      \
    }                                                                                          \
    lds_barrier();                                                                             \
    if (gw_on) {                                                                               \
        float acc_ = 0.f;                                                                      \
        _Pragma("unroll 4") for (int r = r_lo; r < r_hi; ++r) acc_ = fmaf(Z[r * H + gw_o], S[r * H + gw_h], acc_); \
        gacc += (scale) * acc_;                                                                \
    } else if (gb_on) {                                                                        \
        float acc_ = 0.f;                                                                      \
        _Pragma("unroll 4") for (int r = r_lo; r < r_hi; ++r) acc_ += Z[r * H + gw_o];         \
        gacc += (scale) * acc_;                                                                \
    }                                                                                          \
    _Pragma("nounroll") for (int e = tid; e < n_elem; e += 1024) {                             \
        float gs_;                                                                             \
        if (b.no_control) gs_ = Z[e];                                                          \
        else {                                                                                 \
            const int r = e / H, h = e - r * H;                                                \
            gs_ = 0.f;                                                                         \
            _Pragma("unroll 4") for (int oo = 0; oo < H; ++oo) gs_ = fmaf(Z[r * H + oo], wt[h * ldw + oo], gs_); \
        }                                                                                      \
        T[e] = gs_;                                                                            \
    }                                                                                          \
    lds_barrier();                                                                             \
    _Pragma("nounroll") for (int it = 0; it < MAXIT; ++it) {                                   \
        if ((it * kWaves + wave) * RPW >= b.n_rows) { g[it] = 0.f; continue; }                 \
        bool valid; int r;                                                                     \
        const float s_ = gather(it, trp, tci, tva, valid, r);                                  \
        g[it] = valid ? (alpha) * s_ : 0.f;                                                    \
    }                                                                                          \
    lds_barrier();

    for (int i = b.n_ticks - 1; i >= 0; --i) {
        const float dt = b.dt[i];
        const float *yp = b.traj + (size_t)i * n_elem, *gi = b.g_out + (size_t)i * n_elem;
        float y[MAXIT], k1[MAXIT], g[MAXIT], A[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            y[it] = e < n_elem ? yp[e] : 0.f;
        }
        NDCN_PUT_T(y[it])
        NDCN_EVAL_K(k1)
        if (METHOD == NDCN_M_MIDPOINT) {
            const float h = dt / 2.f;
            NDCN_PUT_T(y[it] + k1[it] * dt / 2.f)                                     // ym
            NDCN_VJP(adj, dt, dt, g)                                                   // dL / d ym
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) { A[it] = adj[it] + g[it]; k1[it] = g[it]; }
            NDCN_PUT_T(y[it])
            NDCN_VJP(k1, h, h, g)
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                adj[it] = e < n_elem ? (A[it] + g[it]) + gi[e] : 0.f;
            }
        } else {
            float k2[MAXIT], B2[MAXIT], B1[MAXIT];
            const float c8 = dt / 8.f, c38 = 3.f * c8, d3 = dt / 3.f;
            NDCN_PUT_T(y[it] + dt * k1[it] / 3.f)                                      // u2
            NDCN_EVAL_K(k2)
            NDCN_PUT_T(y[it] + dt * (k1[it] / -3.f + k2[it]))                          // u3
            NDCN_EVAL_K(g)                                                             // k3 (only u4 needs it)
            NDCN_PUT_T(y[it] + dt * (k1[it] - k2[it] + g[it]))                         // u4
            NDCN_VJP(adj, c8, c8, g)                                                   // g_u4 = J4^T (c8 a)
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                A[it] = adj[it] + g[it];
                B2[it] = c38 * adj[it] + (-dt) * g[it];
                B1[it] = c8 * adj[it] + dt * g[it];
                adj[it] = c38 * adj[it] + dt * g[it];                                  // g_k3 (a itself lives on in A)
            }
            NDCN_PUT_T(y[it] + dt * (k1[it] / -3.f + k2[it]))                          // u3
            NDCN_VJP(adj, 1.f, 1.f, g)                                                 // g_u3
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                A[it] = A[it] + g[it];
                B2[it] = B2[it] + dt * g[it];                                          // g_k2
                B1[it] = B1[it] + (-d3) * g[it];
            }
            NDCN_PUT_T(y[it] + dt * k1[it] / 3.f)                                      // u2
            NDCN_VJP(B2, 1.f, 1.f, g)                                                  // g_u2
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                A[it] = A[it] + g[it];
                B1[it] = B1[it] + d3 * g[it];                                          // g_k1
            }
            NDCN_PUT_T(y[it])
            NDCN_VJP(B1, 1.f, 1.f, g)                                                  // g_u1
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                adj[it] = e < n_elem ? (A[it] + g[it]) + gi[e] : 0.f;
            }
        }
    }
#undef NDCN_PUT_T
#undef NDCN_EVAL_K
#undef NDCN_VJP
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = e0 + it * estride;
        if (e < n_elem) b.g_y0[e] = adj[it];
    }
    if (!b.no_control) {
        if ((gw_on || gb_on) && half) T[gw_on ? tid - HH : HH + gw_o] = gacc;
        lds_barrier();
        if (gw_on && !half) b.g_W[tid] += gacc + T[tid];
        else if (gb_on && !half) b.g_b[gw_o] += gacc + T[HH + gw_o];
    }
}

// The reverse sweep on the fast path (HT = H at compile time, the plain ODEFunc, a SYMMETRIC operator - the reference's
// normalised Laplacian / adjacency - so that one packed copy of the entries serves A and A^T):
//   forward pieces   the forward kernel's chains (packed entries, this lane's row of W in registers)
//   g_W              4 x 4 register blocks: thread (block, row group) keeps 16 partial sums for the WHOLE sweep and adds, per row of
//                    its group, the outer product of two 16-byte reads (gZ[r][4a..4a+3], S[r][4b..4b+3]); the groups meet once,
//                    at the end
//   gS = gZ W        lane (r, o) holds column o of W in registers: H/4 16-byte reads of the row's gZ + H fmas
//   LDS: [entries | T | S | Z | srow | rowptr]; registers: the adjoint of the owned elements, the prefetched y_{i-1}
// KEEP (round 5): the forward launch kept S_i = A y_i and K_i of every step (BwdArgs::keep): the sweep re-forms nothing - no gather, no
// Linear, no y_i - a third of a tick's cycles; the next tick's S and K are requested during the transposed gather and put into the S and Z
// panels once their owner is done with them.
template <int MAXIT, int HT, bool KEEP>
__global__ __launch_bounds__(1024) void solve_small_bwd_fast_kernel(BwdArgs b, int n_groups, int rows_per_group) {
    extern __shared__ float lds_raw[];
    constexpr int H = HT, RPW = 64 / HT, NB = HT / 4;        // NB x NB blocks of 4 x 4 outputs
    const int n_elem = b.n_rows * H;
    int2 *ell = reinterpret_cast<int2 *>(lds_raw);
    const int width = b.width;
    float *T = lds_raw + 2 * b.n_rows * width, *S = T + n_elem, *Z = S + n_elem, *Adj = Z + n_elem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *srow = Adj + n_elem + wave * 64;
    float *zrow = Adj + n_elem + kWaves * 64;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    const int e0 = lane_on ? (wave * RPW + q) * H + o : n_elem, estride = kWaves * RPW * H;
    for (int i = tid; i < H; i += 1024) zrow[i] = 0.f;
    build_ell(ell, width, b.n_rows, H, b.rowptr, b.colidx, b.val, (int)(zrow - T), tid);
    float wreg[KEEP ? 1 : HT], wcol[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        if (!KEEP) wreg[h] = lane_on ? b.W[o * H + h] : 0.f; // row o: K[o] = sum_h S[h] W[o][h]
        wcol[h] = lane_on ? b.W[h * H + o] : 0.f;            // column o: gS[o] = sum_oo gZ[oo] W[oo][o]
    }
    const float bias_o = (b.bias && lane_on) ? b.bias[o] : 0.f;
    float pre[MAXIT];                                        // (the adjoint itself lives in LDS: with it in registers the 12-pass build spilled 40)
    {
        const float *a0 = b.a_in ? b.a_in : b.g_out + (size_t)b.n_ticks * n_elem;
        const float *yl = KEEP ? b.keep + (size_t)(b.n_ticks - 1) * 2 * n_elem : b.traj + (size_t)(b.n_ticks - 1) * n_elem;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            if (e < n_elem) Adj[e] = a0[e];
            pre[it] = e < n_elem ? yl[e] : 0.f;
            if (KEEP && e < n_elem) { S[e] = pre[it]; Z[e] = yl[n_elem + e]; }      // S_{T-1}, K_{T-1}
        }
    }
    // g_W blocks: thread t < NB * NB * n_groups owns block (ba, bb) = ((t % (NB NB)) / NB, t % NB) over the rows of group t / (NB NB);
    // g_b: the NB n_groups threads behind them own 4 bias entries each over one group
    const int n_gw = NB * NB * n_groups, n_gb = NB * n_groups;
    const bool gw_on = tid < n_gw, gb_on = !gw_on && tid < n_gw + n_gb;
    const int grp = gw_on ? tid / (NB * NB) : gb_on ? (tid - n_gw) / NB : 0;
    const int ba = gw_on ? (tid % (NB * NB)) / NB : gb_on ? (tid - n_gw) % NB : 0, bb = gw_on ? tid % NB : 0;
    const int r_lo = grp * rows_per_group, r_hi = min(b.n_rows, r_lo + rows_per_group);
    float acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
    long long c_a = 0, c_b = 0, c_c = 0, c0 = b.dbg ? (long long)__builtin_readcyclecounter() : 0;
    lds_barrier();
    for (int i = b.n_ticks - 1; i >= 0; --i) {
        const float dt = b.dt[i];
        long long t_ = b.dbg ? (long long)__builtin_readcyclecounter() : 0;
        if (KEEP) {
            // ---- S holds S_i, Z holds K_i (put there by their owners at the end of the previous tick): gZ = dt a (.) [K_i > 0] in place
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                if (e < n_elem) Z[e] = (b.relu && !(Z[e] > 0.f)) ? 0.f : dt * Adj[e];
            }
        } else {
        // ---- T <- y_i (requested during the previous step's last phase, by the owners);  Z <- dt a
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            if (e < n_elem) { T[e] = pre[it]; Z[e] = dt * Adj[e]; }
        }
        lds_barrier();
        // ---- forward pieces at y_i: S = A y_i, K = relu(W S + b); gZ = dt a (.) [K > 0]
#pragma nounroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((it * kWaves + wave) * RPW >= b.n_rows) break;
            const int r = (it * kWaves + wave) * RPW + q;
            const bool valid = lane_on && r < b.n_rows;
            const float s = ell_gather3(ell + (valid ? r : 0) * width, width, T, o);
            __builtin_amdgcn_wave_barrier();
            srow[lane] = s;
            __builtin_amdgcn_wave_barrier();
            const float4 *sr = reinterpret_cast<const float4 *>(srow + (valid ? q : 0) * H);
            float k = 0.f;
#pragma unroll
            for (int h4 = 0; h4 < NB; ++h4) {
                const float4 sv = sr[h4];
                k = fmaf(sv.x, wreg[4 * h4], k);
                k = fmaf(sv.y, wreg[4 * h4 + 1], k);
                k = fmaf(sv.z, wreg[4 * h4 + 2], k);
                k = fmaf(sv.w, wreg[4 * h4 + 3], k);
            }
            k = k + bias_o;
            if (valid) {
                S[r * H + o] = s;
                if (b.relu && !(k > 0.f)) Z[r * H + o] = 0.f;
            }
        }
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_a += n_ - t_; t_ = n_; }
        // ---- g_W / g_b partial sums over this thread's row group;  gS = gZ W -> T
        if (gw_on) {
#pragma nounroll
            for (int r = r_lo; r < r_hi; ++r) {
                const float4 z = reinterpret_cast<const float4 *>(Z + r * H)[ba];
                const float4 sv = reinterpret_cast<const float4 *>(S + r * H)[bb];
                const float zz[4] = {z.x, z.y, z.z, z.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(zz[u], ss[v], acc[u][v]);
            }
        } else if (gb_on) {
#pragma nounroll
            for (int r = r_lo; r < r_hi; ++r) {
                const float4 z = reinterpret_cast<const float4 *>(Z + r * H)[ba];
                acc[0][0] += z.x; acc[0][1] += z.y; acc[0][2] += z.z; acc[0][3] += z.w;
            }
        }
#pragma nounroll
        for (int it = 0; it < MAXIT; ++it) {
            const int e = e0 + it * estride;
            if ((it * kWaves + wave) * RPW >= b.n_rows) break;
            if (e < n_elem) {
                const float4 *zr = reinterpret_cast<const float4 *>(Z + (e - o));
                float gs = 0.f;
#pragma unroll
                for (int h4 = 0; h4 < NB; ++h4) {
                    const float4 z = zr[h4];
                    gs = fmaf(z.x, wcol[4 * h4], gs);
                    gs = fmaf(z.y, wcol[4 * h4 + 1], gs);
                    gs = fmaf(z.z, wcol[4 * h4 + 2], gs);
                    gs = fmaf(z.w, wcol[4 * h4 + 3], gs);
                }
                T[e] = gs;
            }
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_b += n_ - t_; t_ = n_; }
        // ---- S <- A^T gS = A gS (symmetric), each element by its owner; then a <- (a + S) + g_out[i];  y_{i-1} is requested here so
        // that its registers are live across this phase only
        float kpre[KEEP ? MAXIT : 1];
        if (i > 0) {
            const float *yp = KEEP ? b.keep + (size_t)(i - 1) * 2 * n_elem : b.traj + (size_t)(i - 1) * n_elem;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                if (e < n_elem) pre[it] = yp[e];
                if (KEEP) kpre[it] = e < n_elem ? yp[n_elem + e] : 0.f;
            }
        }
        // (this tick's loss gradient is requested BEFORE the gather that hides its latency: read where it is added, the round trip to
        // memory sat on every tick's critical path - ~5 k of the phase's 16.8 k cycles)
        float gpre[MAXIT];
        {
            const float *gi = b.g_out + (size_t)i * n_elem;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                gpre[it] = e < n_elem ? gi[e] : 0.f;
            }
        }
#pragma nounroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((it * kWaves + wave) * RPW >= b.n_rows) break;
            const int r = (it * kWaves + wave) * RPW + q;
            const bool valid = lane_on && r < b.n_rows;
            const float s = ell_gather3(ell + (valid ? r : 0) * width, width, T, o);
            if (valid) S[r * H + o] = s;
        }
        {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int e = e0 + it * estride;
                if (e < n_elem) Adj[e] = (Adj[e] + S[e]) + gpre[it];
                if (KEEP && i > 0 && e < n_elem) { S[e] = pre[it]; Z[e] = kpre[it]; }     // the next tick's S and K, by their owner
            }
        }
        lds_barrier();
        if (b.dbg) { const long long n_ = __builtin_readcyclecounter(); c_c += n_ - t_; t_ = n_; }
    }
    if (b.dbg && tid == 0) {
        b.dbg[0] = (long long)__builtin_readcyclecounter() - c0;
        b.dbg[1] = c_a; b.dbg[2] = c_b; b.dbg[3] = c_c;
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = e0 + it * estride;
        if (e < n_elem) b.g_y0[e] = Adj[e];
    }
    // the row groups of every g_W / g_b entry meet in LDS: T[group][H*H + H], summed in group order (deterministic)
    const int HH = H * H;
    float *red = T;                                          // n_groups * (HH + H) floats <= 3 panels (checked by the host)
    if (gw_on) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[grp * (HH + H) + (4 * ba + u) * H + 4 * bb + v] = acc[u][v];
    } else if (gb_on) {
#pragma unroll
        for (int v = 0; v < 4; ++v) red[grp * (HH + H) + HH + 4 * ba + v] = acc[0][v];
    }
    lds_barrier();
    if (tid < HH + H) {
        float sum = 0.f;
        for (int g = 0; g < n_groups; ++g) sum += red[g * (HH + H) + tid];
        if (tid < HH) b.g_W[tid] += sum; else b.g_b[tid - HH] += sum;
    }
}

int passes(int64_t n_rows, int H) {
    const int RPW = 64 / H;
    const int64_t wave_passes = (n_rows + RPW - 1) / RPW;
    return (int)((wave_passes + kWaves - 1) / kWaves);
}

template <typename K>
int set_lds_cap(K kern) {
    NDCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
    return NDCN_OK;
}

}  // namespace

// Which fixed-grid solves run as one launch: a plain (un-sharded) operator, H <= 64, the state + W^T in one CU's LDS, and few
// enough rows per thread for the row-local panels to stay in registers (<= 12 passes of 16 waves: 576 rows at H = 20; the
// 32-pass build spilled 80-120 registers and is not shipped).
int solve_small_supported(const ndcn_csr *A, int H, uint32_t flags, int method) {
    static const bool enabled = [] { const char *e = getenv("NDCN_SOLVE_SMALL"); return !(e && e[0] == '0'); }();
    if (!enabled || !A || H < 1 || H > 64 || A->n_rows < 1) return 0;
    if (method != NDCN_M_EULER && method != NDCN_M_MIDPOINT && method != NDCN_M_RK4) return 0;
    if (!(flags & NDCN_F_NO_GRAPH) && A->n_cols != A->n_rows) return 0;
    const int64_t n_elem = A->n_rows * (int64_t)H;
    if (lds_bytes(n_elem, H, A->n_rows, 0, false) > kLdsMax) return 0;
    return passes(A->n_rows, H) <= 12 ? 1 : 0;
}

// The longest row: what the operator's view says (ndcn_csr::max_row_len, filled by ndcn_csr_create / the Python binding once per
// operator), else one small device-to-host copy of rowptr per call.  (Nothing is remembered per pointer: an address says
// nothing about what the arrays behind it hold now.)
static int ell_width(const ndcn_csr *A, hipStream_t st, int *out) {
    if (A->max_row_len > 0) { *out = A->max_row_len; return NDCN_OK; }
    std::vector<int32_t> rp((size_t)A->n_rows + 1);
    NDCN_HIP(hipMemcpyAsync(rp.data(), A->rowptr, rp.size() * 4, hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    int w = 0;
    for (int64_t r = 0; r < A->n_rows; ++r) w = std::max(w, rp[(size_t)r + 1] - rp[(size_t)r]);
    *out = w;
    return NDCN_OK;
}

// Do the forward AND the reverse launch of an Euler solve take their fast forms, so that the forward may keep S_i / K_i for the sweep?
// (Decided from the view alone - ndcn_csr::max_row_len and ::symmetric filled - so that asking costs no synchronisation.)
int solve_small_keep_supported(const ndcn_csr *A, int H, uint32_t flags) {
    const char *env = getenv("NDCN_SOLVE_SMALL_KEEP");               // (read per call: once per solve)
    const bool on = !(env && env[0] == '0');
    static const bool fast_on = [] { const char *e = getenv("NDCN_SOLVE_SMALL_FAST"); return !(e && e[0] == '0'); }();
    if (!on || !fast_on || !solve_small_bwd_supported(A, H, flags, NDCN_M_EULER)) return 0;
    if (!(H == 16 || H == 20) || (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) || A->symmetric != 1) return 0;
    const int width = (A->max_row_len + 2) / 3 * 3;
    const int64_t n_elem = A->n_rows * (int64_t)H;
    if (A->max_row_len < 1 || width > 18) return 0;
    if (lds_bytes_fast(n_elem, A->n_rows, width, H) > kLdsMax || lds_bytes_fast(4 * n_elem, A->n_rows, width, H) > kLdsMax) return 0;
    const int NBh = H / 4;
    const int n_groups = std::max(1, std::min<int>(1024 / (NBh * NBh + NBh), (int)A->n_rows));
    return (int64_t)n_groups * (H * H + H) <= 3 * n_elem ? 1 : 0;
}

int solve_small_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method, const float *y0,
                    const float *h_dt, int64_t n_ticks, float *out, hipStream_t st, float *keep) {
    if (!solve_small_supported(A, H, flags, method)) { set_error("solve_small: unsupported shape"); return NDCN_EINVAL; }
    if (keep && (method != NDCN_M_EULER || !solve_small_keep_supported(A, H, flags))) { set_error("solve_small: nothing keeps S / K for this shape (ndcn_solve_small_keep_supported)"); return NDCN_EINVAL; }
    const int64_t n_elem = A->n_rows * (int64_t)H;
    const bool no_graph = flags & NDCN_F_NO_GRAPH;
    const int64_t nnz = no_graph ? 0 : A->nnz;
    const bool csr = !no_graph && lds_bytes(n_elem, H, A->n_rows, nnz, true) <= kLdsMax;
    static const bool fast_on = [] { const char *e = getenv("NDCN_SOLVE_SMALL_FAST"); return !(e && e[0] == '0'); }();
    int width = 0;
    bool fast = fast_on && (H == 16 || H == 20) && !no_graph && !(flags & NDCN_F_NO_CONTROL);
    if (fast) {
        int rc = ell_width(A, st, &width);
        if (rc) return rc;
        width = (width + 2) / 3 * 3;                           // ell_gather3: entries three at a time
        fast = width >= 1 && width <= 18 && lds_bytes_fast(n_elem, A->n_rows, width, H) <= kLdsMax;
    }
    const size_t lds = fast ? lds_bytes_fast(n_elem, A->n_rows, width, H) : lds_bytes(n_elem, H, A->n_rows, nnz, csr);
    const int np = passes(A->n_rows, H);
    const float *start = y0;
    for (int64_t done = 0; done < n_ticks; done += kChunk) {
        SolveArgs a;
        a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val; a.W = W; a.bias = b; a.y0 = start;
        a.out = out + done * n_elem;
        a.n_rows = (int)A->n_rows; a.H = H; a.nnz = (int)nnz;
        a.n_ticks = (int)std::min<int64_t>(kChunk, n_ticks - done);
        a.relu = (flags & NDCN_F_RELU) ? 1 : 0; a.no_graph = no_graph ? 1 : 0; a.no_control = (flags & NDCN_F_NO_CONTROL) ? 1 : 0;
        a.csr_in_lds = csr ? 1 : 0;
        a.width = width;
        a.keep = keep ? keep + (size_t)done * 2 * n_elem : nullptr;
        for (int i = 0; i < a.n_ticks; ++i) a.dt[i] = h_dt[done + i];
        static const bool dbg_on = [] { const char *e = getenv("NDCN_SS_DEBUG"); return e && e[0] == '1'; }();
        static long long *dbg_buf = nullptr;
        if (dbg_on && !dbg_buf) NDCN_HIP(hipMalloc(&dbg_buf, 4 * sizeof(long long)));
        a.dbg = dbg_on ? dbg_buf : nullptr;
        const double evals = (method == NDCN_M_EULER ? 1 : method == NDCN_M_MIDPOINT ? 2 : 4) * (double)a.n_ticks;
        ProfScope prof(PROF_RHS_FUSED, st, 4.0 * n_elem * (a.n_ticks + 1) + 8.0 * nnz + 4.0 * H * H,
                       evals * (2.0 * nnz * H + 2.0 * (double)A->n_rows * H * H));
#define NDCN_GO(M_, IT_, C_, HT_)                                                              \
        do {                                                                                   \
            auto kern = solve_small_kernel<M_, IT_, C_, HT_>;                                  \
            static std::atomic<unsigned long long> cap_seen{0};                                                       \
            if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; }       \
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a);                         \
        } while (0)
#define NDCN_GO_IT(M_, C_, HT_)                                                                \
        do {                                                                                   \
            if (np <= 4) NDCN_GO(M_, 4, C_, HT_);                                              \
            else NDCN_GO(M_, 12, C_, HT_);                                                     \
        } while (0)
#define NDCN_GO_M(C_, HT_)                                                                     \
        do {                                                                                   \
            if (method == NDCN_M_EULER) NDCN_GO_IT(NDCN_M_EULER, C_, HT_);                     \
            else if (method == NDCN_M_MIDPOINT) NDCN_GO_IT(NDCN_M_MIDPOINT, C_, HT_);          \
            else NDCN_GO_IT(NDCN_M_RK4, C_, HT_);                                              \
        } while (0)
        if (fast && a.keep) {                                 // Euler with S / K kept for the reverse sweep
#define NDCN_KGO(IT_, HT_)                                                                     \
            do {                                                                               \
                auto kern = solve_small_kernel<NDCN_M_EULER, IT_, true, HT_, true>;            \
                static std::atomic<unsigned long long> cap_seen{0};                            \
                if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; } \
                hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a);                     \
            } while (0)
            if (H == 20) { if (np <= 4) NDCN_KGO(4, 20); else NDCN_KGO(12, 20); }
            else { if (np <= 4) NDCN_KGO(4, 16); else NDCN_KGO(12, 16); }
#undef NDCN_KGO
        } else if (fast && H == 20) NDCN_GO_M(true, 20);
        else if (fast) NDCN_GO_M(true, 16);
        else if (csr) NDCN_GO_M(true, 0);
        else NDCN_GO_M(false, 0);
#undef NDCN_GO_M
#undef NDCN_GO_IT
#undef NDCN_GO
        NDCN_LAUNCH_CHECK();
        if (dbg_on) {
            long long h[4];
            NDCN_HIP(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[solve_small] %d ticks: %lld shader cycles (eval %lld, update %lld), %.1f us by the 100 MHz clock -> %.0f MHz\n",
                    a.n_ticks, h[0], h[1], h[2], h[3] / 100.0, h[3] ? 100.0 * h[0] / h[3] : 0.0);
        }
        start = a.out + (size_t)(a.n_ticks - 1) * n_elem;
    }
    return NDCN_OK;
}

int solve_small_bwd_supported(const ndcn_csr *A, int H, uint32_t flags, int method) {
    if (!solve_small_supported(A, H, flags, method)) return 0;
    const int64_t n_elem = A->n_rows * (int64_t)H;
    if (method != NDCN_M_EULER) {
        // midpoint / RK4 have the generic sweep only (no register-resident weights, no ELL image): one compute unit beats the library's
        // multi-launch loops (ndcn_fixed_grid_backward_f32: ~28 dependent launches per RK4 step whatever the size) up to ~4 600
        // state elements - tools/micro/small_rk_ab.py: 1.4 against 5.9 ms at 64 x 16, 5.9 / 8.3 at 196 x 20, 11.2 / 8.7 at 400 x 20
        const char *env = getenv("NDCN_SOLVE_SMALL_RK_MAX");              // (read per call: once per solve)
        const int64_t rk_max = env ? atoll(env) : (int64_t)4608;
        if (n_elem > rk_max) return 0;
    }
    return lds_bytes(n_elem, H, A->n_rows, 0, false, 2 * n_elem) <= kLdsMax && 2 * (H * H + H) <= 1024 && n_elem <= 12 * 1024 ? 1 : 0;
}

// Is the operator equal to its transpose as stored (sorted CSR: the arrays then agree element for element)?  Three small
// device arrays compared through a host copy (a README-sized operator is ~30 KB); callers that know say so in the view.
static int csr_symmetric(const ndcn_csr *A, const ndcn_csr *At, hipStream_t st, int *out) {
    *out = 0;
    if (!At || A == At) { *out = A == At; return NDCN_OK; }
    if (A->n_rows != At->n_rows || A->n_cols != At->n_cols || A->nnz != At->nnz || A->n_rows != A->n_cols) return NDCN_OK;
    const size_t nr = (size_t)A->n_rows + 1, nz = (size_t)A->nnz;
    std::vector<int32_t> a(nr + nz), b(nr + nz);
    std::vector<float> va(nz), vb(nz);
    NDCN_HIP(hipMemcpyAsync(a.data(), A->rowptr, nr * 4, hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipMemcpyAsync(b.data(), At->rowptr, nr * 4, hipMemcpyDeviceToHost, st));
    if (nz) {
        NDCN_HIP(hipMemcpyAsync(a.data() + nr, A->colidx, nz * 4, hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipMemcpyAsync(b.data() + nr, At->colidx, nz * 4, hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipMemcpyAsync(va.data(), A->val, nz * 4, hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipMemcpyAsync(vb.data(), At->val, nz * 4, hipMemcpyDeviceToHost, st));
    }
    NDCN_HIP(hipStreamSynchronize(st));
    *out = a == b && memcmp(va.data(), vb.data(), nz * 4) == 0;
    return NDCN_OK;
}

int solve_small_bwd_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, int method,
                        const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, float *g_y0, float *g_W,
                        float *g_b, hipStream_t st, const float *keep) {
    if (!solve_small_bwd_supported(A, H, flags, method)) { set_error("solve_small_bwd: unsupported shape / method"); return NDCN_EINVAL; }
    if (keep && (method != NDCN_M_EULER || !solve_small_keep_supported(A, H, flags))) { set_error("solve_small_bwd: no kept S / K for this shape (ndcn_solve_small_keep_supported)"); return NDCN_EINVAL; }
    const bool no_graph = flags & NDCN_F_NO_GRAPH, no_control = flags & NDCN_F_NO_CONTROL;
    if (!no_graph && (!At || At->n_rows != A->n_cols || At->nnz != A->nnz)) { set_error("solve_small_bwd: the transposed operator is missing"); return NDCN_EINVAL; }
    const int64_t n_elem = A->n_rows * (int64_t)H;
    const int np = passes(A->n_rows, H);
    const int64_t nnz = no_graph ? 0 : A->nnz;
    const bool csr = !no_graph && lds_bytes(n_elem, H, A->n_rows, nnz, true, 2 * n_elem) <= kLdsMax;
    const size_t lds = lds_bytes(n_elem, H, A->n_rows, nnz, csr, 2 * n_elem);
    // a symmetric operator (the reference's normalised Laplacian / adjacency) serves its own transposed gather from the LDS copy:
    // the view says so (ndcn_csr::symmetric), else the arrays are compared through a host copy, per call
    int symmetric = 0;
    if (!no_graph) {
        if (A->symmetric == 1 || A == At) symmetric = 1;
        else if (A->symmetric == 0) {
            int rc = csr_symmetric(A, At, st, &symmetric);
            if (rc) return rc;
        }
    }
    if (!no_control) {
        NDCN_HIP(hipMemsetAsync(g_W, 0, sizeof(float) * H * H, st));
        NDCN_HIP(hipMemsetAsync(g_b, 0, sizeof(float) * H, st));
    }
    // ticks in chunks of kChunk from the end: each launch hands its adjoint to the next through g_y0
    bool first = true;
    for (int64_t hi = n_ticks, lo; hi > 0; hi = lo) {
        lo = hi > kChunk ? hi - kChunk : 0;
        BwdArgs a;
        a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
        a.t_rowptr = At ? At->rowptr : nullptr; a.t_colidx = At ? At->colidx : nullptr; a.t_val = At ? At->val : nullptr;
        a.W = W; a.bias = b;
        a.traj = traj + lo * n_elem;
        a.g_out = g_out + lo * n_elem;
        a.g_y0 = g_y0; a.g_W = g_W; a.g_b = g_b;
        a.a_in = first ? nullptr : g_y0;
        a.keep = keep ? keep + (size_t)lo * 2 * n_elem : nullptr;
        a.n_rows = (int)A->n_rows; a.H = H; a.nnz = (int)A->nnz; a.n_ticks = (int)(hi - lo);
        a.relu = (flags & NDCN_F_RELU) ? 1 : 0; a.no_graph = no_graph ? 1 : 0; a.no_control = no_control ? 1 : 0;
        for (int i = 0; i < a.n_ticks; ++i) a.dt[i] = h_dt[lo + i];
        static const bool dbg_on = [] { const char *e = getenv("NDCN_SS_DEBUG"); return e && e[0] == '1'; }();
        static long long *dbg_buf = nullptr;
        if (dbg_on && !dbg_buf) NDCN_HIP(hipMalloc(&dbg_buf, 4 * sizeof(long long)));
        a.dbg = dbg_on ? dbg_buf : nullptr;
        ProfScope prof(PROF_RHS_FUSED, st, 4.0 * n_elem * (2.0 * a.n_ticks + 3), 3.0 * a.n_ticks * (2.0 * A->nnz * H + 2.0 * (double)A->n_rows * H * H));
        static const bool fast_on = [] { const char *e = getenv("NDCN_SOLVE_SMALL_FAST"); return !(e && e[0] == '0'); }();
        const bool fast_shape = fast_on && (H == 16 || H == 20) && !no_graph && !no_control && symmetric;
        const int NBh = fast_shape ? H / 4 : 1;
        const int n_groups = std::max(1, std::min<int>(1024 / (NBh * NBh + NBh), (int)A->n_rows));
        const int rows_per_group = (int)((A->n_rows + n_groups - 1) / n_groups);
        int width = 0;
        if (fast_shape) { int rcw = ell_width(A, st, &width); if (rcw) return rcw; width = (width + 2) / 3 * 3; }
        const size_t lds_fast = lds_bytes_fast(4 * n_elem, A->n_rows, width, H);
        const bool fast = fast_shape && width >= 1 && width <= 18 && lds_fast <= kLdsMax && (int64_t)n_groups * (H * H + H) <= 3 * n_elem;
        a.width = width;
        if (method != NDCN_M_EULER) {
#define NDCN_RGO(M_, IT_, C_)                                                                  \
            do {                                                                               \
                auto kern = solve_small_bwd_rk_kernel<M_, IT_, C_>;                            \
                static std::atomic<unsigned long long> cap_seen{0};                            \
                if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; } \
                hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a, symmetric);          \
            } while (0)
#define NDCN_RGO_IT(M_, C_)                                                                    \
            do {                                                                               \
                if (np <= 4) NDCN_RGO(M_, 4, C_);                                              \
                else if (np <= 9) NDCN_RGO(M_, 9, C_);                                         \
                else NDCN_RGO(M_, 12, C_);                                                     \
            } while (0)
            if (method == NDCN_M_MIDPOINT) { if (csr) NDCN_RGO_IT(NDCN_M_MIDPOINT, true); else NDCN_RGO_IT(NDCN_M_MIDPOINT, false); }
            else { if (csr) NDCN_RGO_IT(NDCN_M_RK4, true); else NDCN_RGO_IT(NDCN_M_RK4, false); }
#undef NDCN_RGO_IT
#undef NDCN_RGO
        } else if (fast) {
#define NDCN_FGO(IT_, HT_)                                                                     \
            do {                                                                               \
                if (a.keep) {                                                                  \
                    auto kern = solve_small_bwd_fast_kernel<IT_, HT_, true>;                   \
                    static std::atomic<unsigned long long> cap_seen{0};                        \
                    if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; } \
                    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds_fast, st, a, n_groups, rows_per_group); \
                } else {                                                                       \
                    auto kern = solve_small_bwd_fast_kernel<IT_, HT_, false>;                  \
                    static std::atomic<unsigned long long> cap_seen{0};                        \
                    if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; } \
                    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds_fast, st, a, n_groups, rows_per_group); \
                }                                                                              \
            } while (0)
            // (9 passes: the README's 400 x 20 - the 12-pass build of the same kernel spills 22 registers)
            if (H == 20) { if (np <= 4) NDCN_FGO(4, 20); else if (np <= 9) NDCN_FGO(9, 20); else NDCN_FGO(12, 20); }
            else { if (np <= 4) NDCN_FGO(4, 16); else if (np <= 9) NDCN_FGO(9, 16); else NDCN_FGO(12, 16); }
#undef NDCN_FGO
        } else {
#define NDCN_GO(IT_, C_)                                                                       \
        do {                                                                                   \
            auto kern = solve_small_bwd_kernel<IT_, C_>;                                       \
            static std::atomic<unsigned long long> cap_seen{0};                                                       \
            if (once_per_device(cap_seen)) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; }       \
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a, symmetric);              \
        } while (0)
        if (np <= 4) { if (csr) NDCN_GO(4, true); else NDCN_GO(4, false); }
        else { if (csr) NDCN_GO(12, true); else NDCN_GO(12, false); }
#undef NDCN_GO
        }
        NDCN_LAUNCH_CHECK();
        if (dbg_on) {
            long long h[4];
            NDCN_HIP(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[solve_small_bwd] %d ticks: %lld shader cycles: forward pieces %lld, g_W + gS %lld, transposed gather %lld\n",
                    a.n_ticks, h[0], h[1], h[2], h[3]);
        }
        first = false;
    }
    return NDCN_OK;
}

}  // namespace ndcn
