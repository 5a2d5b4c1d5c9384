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

#include "kernels.h"

#pragma clang fp contract(off)

namespace ndcn {

namespace {

constexpr int kWaves = 16;
constexpr int kChunk = 128;                 // ticks per launch (the step sizes ride in the kernel arguments)
constexpr size_t kLdsMax = 160 * 1024;

struct SolveArgs {
    const int *rowptr, *colidx;
    const float *val;
    const float *W, *bias;
    const float *y0;                 // [n_rows][H] state at the start of this launch
    float *out;                      // [n_ticks][n_rows][H]
    int n_rows, H, nnz, n_ticks, relu, no_graph, no_control, csr_in_lds;
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
__device__ __forceinline__ float eval_rhs(const SolveArgs &a, const Lds &l, const int *rp, const int *ci, const float *va, int r, int q,
                                          int o, int lane, int wave, bool valid, float bias_o) {
    const int H = a.H;
    float s = 0.f;
    if (a.no_graph) {
        if (valid) s = l.T[r * H + o];
    } else {
        int j0 = 0, cnt = 0;
        if (valid) { j0 = rp[r]; cnt = rp[r + 1] - j0; }
        for (int j = 0; __any(j < cnt); ++j)
            if (j < cnt) s = fmaf(va[j0 + j], l.T[ci[j0 + j] * H + o], s);
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
        for (int h = 0; h < H; ++h) k = fmaf(sr[h], l.wt[h * ldw + o], k);
        k = k + bias_o;
        if (a.relu) k = relu_nan(k);
    }
    return k;
}

// METHOD: NDCN_M_EULER / MIDPOINT / RK4.  MAXIT: passes a wave makes over its rows (register arrays are indexed by pass)
template <int METHOD, int MAXIT, bool CSR_LDS>
__global__ __launch_bounds__(1024) void solve_small_kernel(SolveArgs a) {
    extern __shared__ float lds_raw[];
    const int H = a.H, n_elem = a.n_rows * H;
    const Lds l = carve(lds_raw, n_elem, H, a.n_rows, a.nnz, CSR_LDS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RPW = 64 / H;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    // ---- stage: W^T, CSR, the initial state
    if (!a.no_control)
        for (int i = tid; i < H * H; i += 1024) {
            const int oo = i / H, h = i - oo * H;
            l.wt[h * (H + 1) + oo] = a.W[i];
        }
    if (CSR_LDS) {
        for (int i = tid; i <= a.n_rows; i += 1024) l.rowptr[i] = a.rowptr[i];
        for (int i = tid; i < a.nnz; i += 1024) { l.colidx[i] = a.colidx[i]; l.val[i] = a.val[i]; }
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
    __syncthreads();
    // one right-hand side over the stage input in T -> dst[it]; then T <- the next stage input
#define NDCN_EVAL(dst)                                                                                         \
    _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                                     \
        const int r = (it * kWaves + wave) * RPW + q;                                                          \
        if ((it * kWaves + wave) * RPW < a.n_rows)                                                             \
            dst[it] = eval_rhs(a, l, rp, ci, va, r, q, o, lane, wave, lane_on && r < a.n_rows, bias_o);        \
    }                                                                                                          \
    __syncthreads();
#define NDCN_PUT(expr, also_out)                                                                               \
    _Pragma("unroll") for (int it = 0; it < MAXIT; ++it) {                                                     \
        const int r = (it * kWaves + wave) * RPW + q;                                                          \
        if (lane_on && r < a.n_rows) {                                                                         \
            const float v_ = (expr);                                                                           \
            l.T[r * H + o] = v_;                                                                               \
            if (also_out) { y[it] = v_; a.out[(size_t)tick * n_elem + r * H + o] = v_; }                       \
        }                                                                                                      \
    }                                                                                                          \
    __syncthreads();
    for (int tick = 0; tick < a.n_ticks; ++tick) {
        const float dt = a.dt[tick];
        if (METHOD == NDCN_M_EULER) {
            NDCN_EVAL(k1)
            NDCN_PUT(y[it] + dt * k1[it], true)                                              // fixed_grid.py:8 + solvers.py:92
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
    const float *a_in;               // nullable [n]: the adjoint at the LAST tick of this launch, handed over by the launch that
                                     // covered the later ticks (it already holds that tick's g_out); NULL: g_out[n_ticks] itself
    int n_rows, H, nnz, n_ticks, relu, no_graph, no_control;
    float dt[kChunk];
};

// LDS: T (y_i, then gS), wt (W^T for the forward chain), srow, then  S [n] (A y_i),  Z [n] (gZ),  w (W as stored: gS = gZ W)
template <int MAXIT>
__global__ __launch_bounds__(1024) void solve_small_bwd_kernel(BwdArgs b) {
    extern __shared__ float lds_raw[];
    const int H = b.H, n_elem = b.n_rows * H;
    float *T = lds_raw, *wt = T + n_elem, *srow_all = wt + H * (H + 1);
    float *S = srow_all + kWaves * 64, *Z = S + n_elem, *w = Z + n_elem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RPW = 64 / H;
    const int q = lane / H, o = lane - q * H;
    const bool lane_on = q < RPW;
    if (!b.no_control)
        for (int i = tid; i < H * H; i += 1024) {
            const int oo = i / H, h = i - oo * H;
            wt[h * (H + 1) + oo] = b.W[i];
            w[i] = b.W[i];
        }
    const float bias_o = (!b.no_control && b.bias && lane_on) ? b.bias[o] : 0.f;
    float adj[MAXIT];                                        // a = dL/dy_{i+1}, carried across the sweep
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int r = (it * kWaves + wave) * RPW + q;
        const bool valid = lane_on && r < b.n_rows;
        adj[it] = 0.f;
        if (valid) adj[it] = b.a_in ? b.a_in[r * H + o] : b.g_out[(size_t)b.n_ticks * n_elem + r * H + o];
    }
    // gradient accumulators of W: thread t < H*H owns g_W[t / H][t % H]; g_b: thread t < H
    float gw = 0.f, gb = 0.f;
    __syncthreads();
    for (int i = b.n_ticks - 1; i >= 0; --i) {
        const float dt = b.dt[i];
        const float *yi = b.traj + (size_t)i * n_elem;
        // ---- T <- y_i
        for (int e = tid; e < n_elem; e += 1024) T[e] = yi[e];
        __syncthreads();
        // ---- forward pieces at y_i: S = A y_i, K = relu(W S + b); gZ = dt a (.) [K > 0]  (the mask of relu's output)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int r = (it * kWaves + wave) * RPW + q;
            if ((it * kWaves + wave) * RPW >= b.n_rows) continue;
            const bool valid = lane_on && r < b.n_rows;
            float s = 0.f;
            if (b.no_graph) {
                if (valid) s = T[r * H + o];
            } else {
                int j0 = 0, cnt = 0;
                if (valid) { j0 = b.rowptr[r]; cnt = b.rowptr[r + 1] - j0; }
                for (int j = 0; __any(j < cnt); ++j)
                    if (j < cnt) s = fmaf(b.val[j0 + j], T[b.colidx[j0 + j] * H + o], s);
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
                    for (int h = 0; h < H; ++h) k = fmaf(sr[h], wt[h * (H + 1) + o], k);
                    k = k + bias_o;
                }
            }
            if (valid) {
                const float gk = dt * adj[it];
                S[r * H + o] = s;
                Z[r * H + o] = (!b.relu || k > 0.f) ? gk : 0.f;
            }
        }
        __syncthreads();
        // ---- g_W[oo][h] += sum_r Z[r][oo] S[r][h];  g_b[oo] += sum_r Z[r][oo];  gS = Z W  -> T (y_i is no longer needed)
        if (!b.no_control) {
            if (tid < H * H) {
                const int oo = tid / H, h = tid - oo * H;
                float acc = 0.f;
                for (int r = 0; r < b.n_rows; ++r) acc = fmaf(Z[r * H + oo], S[r * H + h], acc);
                gw += acc;
            } else if (tid < H * H + H) {
                const int oo = tid - H * H;
                float acc = 0.f;
                for (int r = 0; r < b.n_rows; ++r) acc += Z[r * H + oo];
                gb += acc;
            }
        }
        for (int e = tid; e < n_elem; e += 1024) {
            float gs;
            if (b.no_control) gs = Z[e];
            else {
                const int r = e / H, h = e - r * H;
                gs = 0.f;
                for (int oo = 0; oo < H; ++oo) gs = fmaf(Z[r * H + oo], w[oo * H + h], gs);
            }
            T[e] = gs;
        }
        __syncthreads();
        // ---- a <- a + g_out[i] + A^T gS   (no_graph: + gS)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int r = (it * kWaves + wave) * RPW + q;
            if (!(lane_on && r < b.n_rows)) continue;
            float s = 0.f;
            if (b.no_graph) s = T[r * H + o];
            else
                for (int j = b.t_rowptr[r]; j < b.t_rowptr[r + 1]; ++j) s = fmaf(b.t_val[j], T[b.t_colidx[j] * H + o], s);
            adj[it] = (adj[it] + s) + b.g_out[(size_t)i * n_elem + r * H + o];
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int r = (it * kWaves + wave) * RPW + q;
        if (lane_on && r < b.n_rows) b.g_y0[r * H + o] = adj[it];
    }
    if (!b.no_control) {
        if (tid < H * H) b.g_W[tid] += gw;
        else if (tid < H * H + H) b.g_b[tid - H * H] += gb;
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

int solve_small_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method, const float *y0,
                    const float *h_dt, int64_t n_ticks, float *out, hipStream_t st) {
    if (!solve_small_supported(A, H, flags, method)) { set_error("solve_small: unsupported shape"); return NDCN_EINVAL; }
    const int64_t n_elem = A->n_rows * (int64_t)H;
    const bool no_graph = flags & NDCN_F_NO_GRAPH;
    const int64_t nnz = no_graph ? 0 : A->nnz;
    const bool csr = !no_graph && lds_bytes(n_elem, H, A->n_rows, nnz, true) <= kLdsMax;
    const size_t lds = lds_bytes(n_elem, H, A->n_rows, nnz, csr);
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
        for (int i = 0; i < a.n_ticks; ++i) a.dt[i] = h_dt[done + i];
        const double evals = (method == NDCN_M_EULER ? 1 : method == NDCN_M_MIDPOINT ? 2 : 4) * (double)a.n_ticks;
        ProfScope prof(PROF_RHS_FUSED, st, 4.0 * n_elem * (a.n_ticks + 1) + 8.0 * nnz + 4.0 * H * H,
                       evals * (2.0 * nnz * H + 2.0 * (double)A->n_rows * H * H));
#define NDCN_GO(M_, IT_, C_)                                                                   \
        do {                                                                                   \
            auto kern = solve_small_kernel<M_, IT_, C_>;                                       \
            static bool cap_set = false;                                                       \
            if (!cap_set) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; cap_set = true; } \
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a);                         \
        } while (0)
#define NDCN_GO_IT(M_, C_)                                                                     \
        do {                                                                                   \
            if (np <= 4) NDCN_GO(M_, 4, C_);                                                   \
            else NDCN_GO(M_, 12, C_);                                                          \
        } while (0)
        if (method == NDCN_M_EULER) { if (csr) NDCN_GO_IT(NDCN_M_EULER, true); else NDCN_GO_IT(NDCN_M_EULER, false); }
        else if (method == NDCN_M_MIDPOINT) { if (csr) NDCN_GO_IT(NDCN_M_MIDPOINT, true); else NDCN_GO_IT(NDCN_M_MIDPOINT, false); }
        else if (np <= 4) { if (csr) NDCN_GO(NDCN_M_RK4, 4, true); else NDCN_GO(NDCN_M_RK4, 4, false); }
        else { if (csr) NDCN_GO(NDCN_M_RK4, 12, true); else NDCN_GO(NDCN_M_RK4, 12, false); }
#undef NDCN_GO_IT
#undef NDCN_GO
        NDCN_LAUNCH_CHECK();
        start = a.out + (size_t)(a.n_ticks - 1) * n_elem;
    }
    return NDCN_OK;
}

int solve_small_bwd_supported(const ndcn_csr *A, int H, uint32_t flags, int method) {
    if (method != NDCN_M_EULER || !solve_small_supported(A, H, flags, method)) return 0;
    const int64_t n_elem = A->n_rows * (int64_t)H;
    return lds_bytes(n_elem, H, A->n_rows, 0, false, 2 * n_elem + H * H) <= kLdsMax && H * H + H <= 1024 ? 1 : 0;
}

int solve_small_bwd_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, int method,
                        const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, float *g_y0, float *g_W,
                        float *g_b, hipStream_t st) {
    if (!solve_small_bwd_supported(A, H, flags, method)) { set_error("solve_small_bwd: unsupported shape / method"); return NDCN_EINVAL; }
    const bool no_graph = flags & NDCN_F_NO_GRAPH, no_control = flags & NDCN_F_NO_CONTROL;
    if (!no_graph && (!At || At->n_rows != A->n_cols || At->nnz != A->nnz)) { set_error("solve_small_bwd: the transposed operator is missing"); return NDCN_EINVAL; }
    const int64_t n_elem = A->n_rows * (int64_t)H;
    const size_t lds = lds_bytes(n_elem, H, A->n_rows, 0, false, 2 * n_elem + H * H);
    const int np = passes(A->n_rows, H);
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
        a.n_rows = (int)A->n_rows; a.H = H; a.nnz = (int)A->nnz; a.n_ticks = (int)(hi - lo);
        a.relu = (flags & NDCN_F_RELU) ? 1 : 0; a.no_graph = no_graph ? 1 : 0; a.no_control = no_control ? 1 : 0;
        for (int i = 0; i < a.n_ticks; ++i) a.dt[i] = h_dt[lo + i];
        ProfScope prof(PROF_RHS_FUSED, st, 4.0 * n_elem * (2.0 * a.n_ticks + 3), 3.0 * a.n_ticks * (2.0 * A->nnz * H + 2.0 * (double)A->n_rows * H * H));
#define NDCN_GO(IT_)                                                                           \
        do {                                                                                   \
            auto kern = solve_small_bwd_kernel<IT_>;                                           \
            static bool cap_set = false;                                                       \
            if (!cap_set) { int rc_ = set_lds_cap(kern); if (rc_) return rc_; cap_set = true; } \
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, a);                         \
        } while (0)
        if (np <= 4) NDCN_GO(4); else NDCN_GO(12);
#undef NDCN_GO
        NDCN_LAUNCH_CHECK();
        first = false;
    }
    return NDCN_OK;
}

}  // namespace ndcn
