// Training through the adaptive solver as ONE native object: a dopri5 solve of y' = relu(W (A y) + b) that keeps what its
// reverse pass needs (the "tape"), and that reverse pass.
//
// The reference trains by autograd THROUGH its solver (heat_dynamics.py:313-334, dgnn.py:192-222): every panel operation of
// every attempted step and the whole scalar chain of the step-size controller (dt, t0 / t1, the initial step, the interpolation
// abscissa are tensors with history: dopri5.py:76-122, rk_common.py:41-61, misc.py:84-170, interp.py:38-65) are nodes of a graph
// the interpreter builds and walks - ~65 nodes forward and as many backward for a README-sized solve, 5 + 7 ms of host time
// around < 1 ms of kernels (tools/micro/c1_host_profile.py).  Here the same launches, in the same order with the same arguments
// as ndcn_amd/torchdiffeq/_impl/autograd_path.py issues them (forward values bit-identical: same kernels), are enqueued by one
// C call per direction; the scalar chain's adjoint is written out by hand below (tape_backward: "scalar chain") and follows
// torch's conventions where they matter (maximum / minimum halve the gradient on ties, python max() picks the first maximum,
// float32 tensors take float32 gradients).
//
// Backward of one attempted step, stage sums in PULL form (autograd_path._RhsStagePullFn): with g_u_m the gradient of the m-th
// stage input (u_7 = y1), evaluation e receives  g_k_e = (what the dense output, the error ratio and later steps sent)
// + sum_{m > e} dt beta_{m,e} g_u_m  in one ndcn_rk_combine_f32 pass, then the right-hand side's VJP (mask, gS = gZ W, A^T gS;
// g_W += gZ^T S, g_b += sum gZ) yields g_u_e; the step size receives <g_u_m, u_m - y0> / dt per stage (one ndcn_rk_dot_diff_f32
// pass), the error ratio's and the dense output's inner products from their VJP kernels.  All inner products of an attempt land
// in one device array and come back in ONE read: the only host synchronisation per attempted step, as in the forward pass.
//
// Memory: panels come from the caller's allocator (ndcn_alloc_fn: torch's caching allocator through the binding) and live until
// ndcn_tape_destroy: 12 panels per attempted step (stage inputs and derivatives) + ~24 of scratch in the reverse pass.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "common.h"
#include "dopri5_tableau.h"
#include "hostrec.h"
#include "kernels.h"

using namespace ndcn;

namespace {

struct Attempt {
    bool accept = false, fused_err = false;
    double t0 = 0, dt = 0, dt_next = 0, factor = 0;
    float dts = 0, ratio = 0;
    const float *y0 = nullptr;
    const float *k[7] = {};
    const float *u[8] = {};            // u[2..6]: stage inputs, u[7] = y1
    const float *S[8] = {};            // S[e] = A u[e] where the evaluation's launch wrote it on the side (keep_s), else null
    std::vector<int> dense;            // indices into ndcn_tape::groups, in forward order
};

struct DenseGroup {
    int attempt = -1, nt = 0;
    int tick[7] = {};
    float x[7] = {};
    float a0 = 0, a1 = 0;
};

struct HostScratch {                   // per host thread: pinned mirror of the device records + an event
    double *h = nullptr;               // kDotSlots x 8 doubles (inner products) ...
    double *rec = nullptr;             // ... + 8 (a reduction record), same allocation
    double *h_dev = nullptr, *rec_dev = nullptr;      // their device aliases (polling mode: hostrec.h), else null
    hipEvent_t ev = nullptr;
};

constexpr int kDotSlots = 40;          // 8 doubles each
constexpr int kDenseBatch = 24;        // dense-output groups (<= 7 ticks each) whose inner products one read-back carries; the other slots: error + 6 rows

int host_scratch(HostScratch **out) {
    static thread_local HostScratch hs;
    if (!hs.h) {
        NDCN_HIP(hipHostMalloc(reinterpret_cast<void **>(&hs.h), (size_t)(kDotSlots + 1) * 8 * sizeof(double), hipHostMallocDefault));
        hs.rec = hs.h + (size_t)kDotSlots * 8;
        NDCN_HIP(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
        void *alias = nullptr;
        if (poll_records_enabled() && hipHostGetDevicePointer(&alias, hs.h, 0) == hipSuccess && alias) {
            hs.h_dev = static_cast<double *>(alias);
            hs.rec_dev = hs.h_dev + (size_t)kDotSlots * 8;
        }
    }
    *out = &hs;
    return NDCN_OK;
}

}  // namespace

struct ndcn_tape {
    ndcn_csr A = {}, At = {};
    const float *W = nullptr, *b = nullptr;
    int H = 0;
    uint32_t flags = 0;
    int64_t n_rows = 0, n = 0;
    double rtol = 0, atol = 0, safety = 0, ifactor = 0, dfactor = 0;
    int64_t max_steps = 0;
    bool first_given = false;
    bool keep_s = false;               // opts[5]: evaluations on the lattice-plan kernel also store S = A u (ndcn_rhs_rk_adj_f32's s_out): one panel
                                       // more per evaluation on the tape instead of one SpMM per evaluation in the reverse pass
    ndcn_alloc_fn alloc = nullptr;
    void *alloc_ctx = nullptr;
    // arena
    char *chunk = nullptr;
    size_t chunk_left = 0, panel_bytes = 0;
    // forward record
    std::vector<double> ticks;
    std::vector<Attempt> attempts;
    std::vector<DenseGroup> groups;
    std::vector<double> log;
    int64_t nfe = 0;
    const float *y_in = nullptr, *f0 = nullptr, *yh = nullptr, *f1 = nullptr;      // initial step: y0, f(y0), y0 + h0 f0, f(yh)
    double s0 = 0, s1 = 0, s2 = 0;     // the three sums of squares
    float d0 = 0, d1 = 0, d2r = 0, d2 = 0, h0 = 0, h1 = 0;
    bool h0_const = false, h1_alt = false;
    double dt_init = 0;
    // device scratch
    double *d_red = nullptr, *d_dots = nullptr;
    void *d_ws = nullptr, *d_ws2 = nullptr, *d_bws = nullptr;
    float *work = nullptr;             // rhs scratch (packed weights for H = 256)
    bool packed = false;
    void *bwork = nullptr;             // linear_bwd scratch
    bool bpacked = false;
    bool bwd_marked = false;           // arena position behind the forward record: every reverse pass starts its scratch there
    char *mark_chunk = nullptr;
    size_t mark_left = 0;
};

namespace {

int arena(ndcn_tape *t, size_t bytes, void **out) {
    bytes = (bytes + 255u) & ~(size_t)255u;
    if (bytes > t->chunk_left) {
        // small panels share a chunk (one allocator call per 64), large ones get their own
        size_t want = bytes;
        if (bytes * 64 <= ((size_t)32 << 20)) want = bytes * 64;
        void *p = t->alloc(t->alloc_ctx, (int64_t)want + 256);
        if (!p) { set_error("tape: the allocator returned no memory for %zu bytes", want); return NDCN_EINVAL; }
        t->chunk = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 255u) & ~(uintptr_t)255u);
        t->chunk_left = want;
    }
    *out = t->chunk;
    t->chunk += bytes;
    t->chunk_left -= bytes;
    return NDCN_OK;
}

int panel(ndcn_tape *t, float **p) {
    void *q = nullptr;
    int rc = arena(t, t->panel_bytes, &q);
    *p = static_cast<float *>(q);
    return rc;
}

int rhs_plain(ndcn_tape *t, const float *x, float *out, hipStream_t st) {
    t->nfe++;
    uint32_t fl = t->flags | (t->packed ? NDCN_F_PACKED : 0u);
    int rc = rhs_rk_f32(&t->A, x, nullptr, t->A.n_cols, t->W, t->b, out, t->work, t->H, fl, 0, nullptr, nullptr, nullptr, 0, nullptr,
                        0.f, 0.f, nullptr, nullptr, st, nullptr);
    return rc;
}

// the n doubles the launches enqueued since the matching arm wrote at `d_src` -> hs->h[0..n) (polling mode: d_src IS hs->h's alias)
int fetch(ndcn_tape *t, const double *d_src, int n_doubles, hipStream_t st, HostScratch *hs) {
    (void)t;
    if (hs->h_dev && d_src == hs->h_dev) return rec_wait(hs->h, n_doubles, st);
    NDCN_HIP(hipMemcpyAsync(hs->h, d_src, (size_t)n_doubles * sizeof(double), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipEventRecord(hs->ev, st));
    NDCN_HIP(hipEventSynchronize(hs->ev));
    return NDCN_OK;
}

// the 2-double reduction record: armed before the launches, then read into hs->h[0..1]
void arm_record(ndcn_tape *t, HostScratch *hs) {
    if (hs->rec_dev && t->d_red == hs->rec_dev) rec_arm(hs->rec, 2);
}
int fetch_record(ndcn_tape *t, hipStream_t st, HostScratch *hs) {
    if (hs->rec_dev && t->d_red == hs->rec_dev) {
        int rc = rec_wait(hs->rec, 2, st);
        hs->h[0] = hs->rec[0];
        hs->h[1] = hs->rec[1];
        return rc;
    }
    return fetch(t, t->d_red, 2, st, hs);
}

// misc.py:71-76: x.norm() / numel ** 0.5 as float32 values (autograd_path._rms_value)
float rms_value(double s, double numel) {
    const float nrm = (s == s && s >= 0) ? (float)sqrt(s) : NAN;
    return nrm / (float)sqrt(numel);
}

int rms(ndcn_tape *t, const float *a, const float *b, const float *y, hipStream_t st, HostScratch *hs, double &sum, double &bad) {
    arm_record(t, hs);
    int rc = scaled_sumsq_f32(a, b, y, (float)t->rtol, (float)t->atol, t->n, t->d_red, t->d_ws, st);
    if (rc) return rc;
    rc = fetch_record(t, st, hs);
    if (rc) return rc;
    sum = hs->h[0];
    bad = hs->h[1];
    return NDCN_OK;
}

// misc.py:84-143 with order = 4, as autograd_path._initial_step evaluates it
int initial_step(ndcn_tape *t, hipStream_t st, HostScratch *hs, int64_t &pending_bad) {
    double bad;
    int rc = rms(t, t->y_in, nullptr, t->y_in, st, hs, t->s0, bad);
    if (rc) return rc;
    pending_bad = (int64_t)bad;
    rc = rms(t, t->f0, nullptr, t->y_in, st, hs, t->s1, bad);
    if (rc) return rc;
    t->d0 = rms_value(t->s0, (double)t->n);
    t->d1 = rms_value(t->s1, (double)t->n);
    t->h0_const = t->d0 < 1e-5 || t->d1 < 1e-5;
    t->h0 = t->h0_const ? 1e-6f : 0.01f * (t->d0 / t->d1);
    float *yh, *f1;
    if ((rc = panel(t, &yh)) || (rc = panel(t, &f1))) return rc;
    const float *kp[1] = {t->f0};
    const float cp[1] = {t->h0};
    rc = rk_combine_f32(yh, t->y_in, kp, cp, 1, t->n, st);
    if (rc) return rc;
    rc = rhs_plain(t, yh, f1, st);
    if (rc) return rc;
    t->yh = yh;
    t->f1 = f1;
    rc = rms(t, f1, t->f0, t->y_in, st, hs, t->s2, bad);
    if (rc) return rc;
    t->d2r = rms_value(t->s2, (double)t->n);
    t->d2 = t->d2r / t->h0;
    t->h1_alt = t->d1 <= 1e-15 && t->d2 <= 1e-15;
    if (t->h1_alt) {
        const float a = 1e-6f, b = t->h0 * 1e-3f;
        t->h1 = a > b ? a : b;
    } else {
        const float m = t->d1 >= t->d2 ? t->d1 : t->d2;       // python max([d1, d2]): the first maximum
        const float a = (1.0f / m) * 0.01f;                   // solver.hip initial_step: the same two float32 roundings
        t->h1 = (float)pow((double)a, 1. / 5.);
    }
    const float h100 = 100.f * t->h0;
    t->dt_init = (double)(h100 < t->h1 ? h100 : t->h1);
    if (isnan(h100) || isnan(t->h1)) t->dt_init = NAN;
    return NDCN_OK;
}

// the non-zero terms of row `beta` scaled by the step size (misc.py:25 in float32), k indices kept
int terms(float dts, const double *beta, int n, const float *const *kall, const float **kp, float *cp, int *idx) {
    int m = 0;
    for (int j = 0; j < n; ++j) {
        const float bj = (float)beta[j];
        if (bj == 0.f) continue;
        kp[m] = kall[j];
        cp[m] = dts * bj;
        if (idx) idx[m] = j;
        ++m;
    }
    return m;
}

int forward_attempt(ndcn_tape *t, Attempt &a, hipStream_t st, HostScratch *hs, double &bad_out) {
    int rc;
    arm_record(t, hs);
    float *u[8] = {}, *k[7] = {}, *S[8] = {};
    for (int e = 2; e <= 7; ++e)
        if ((rc = panel(t, &u[e]))) return rc;
    const bool both = !(t->flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL));
    // (the narrow-panel kernel - rhs_small.hip: what rhs_rk_f32 picks for H <= 128 at launch-bound sizes - writes S on the side too)
    const bool small_s = both && !rhs_fused2_supported(&t->A, t->H, t->flags) && rhs_small_supported(&t->A, t->H, t->flags);
    for (int j = 1; j < 7; ++j)
        if ((rc = panel(t, &k[j]))) return rc;
    const float *kall[7] = {a.k[0], k[1], k[2], k[3], k[4], k[5], k[6]};
    const float *kp[8];
    float cp[8];
    const float dts = a.dts;
    int m = terms(dts, kBeta[0], 1, kall, kp, cp, nullptr);
    rc = rk_combine_f32(u[2], a.y0, kp, cp, m, t->n, st);
    if (rc) return rc;
    const uint32_t fl = t->flags;
    for (int i = 0; i < 5; ++i) {
        // evaluation i + 2: k[i + 1] = f(u[i + 2]); its epilogue forms u[i + 3] from row i + 1 of the tableau
        int mp = 0;
        for (int j = 0; j <= i; ++j) {
            const float bj = (float)kBeta[i + 1][j];
            if (bj == 0.f) continue;
            kp[mp] = kall[j];
            cp[mp] = dts * bj;
            ++mp;
        }
        cp[mp] = dts * (float)kBeta[i + 1][i + 1];
        t->nfe++;
        RkOpt opt = {};
        if (t->keep_s && both && (rhs_adj_supported(&t->A, t->H, fl, NDCN_RK_COMBINE, mp) || small_s)) {
            if ((rc = panel(t, &S[i + 2]))) return rc;
            opt.s_out = S[i + 2];
        }
        rc = rhs_rk_f32(&t->A, u[i + 2], nullptr, t->A.n_cols, t->W, t->b, k[i + 1], t->work, t->H, fl | (t->packed ? NDCN_F_PACKED : 0u),
                        NDCN_RK_COMBINE, a.y0, kp, cp, mp, u[i + 3], 0.f, 0.f, nullptr, nullptr, st, opt.s_out ? &opt : nullptr);
        if (rc) return rc;
    }
    a.fused_err = t->n > aten_order_max_elems();
    if (a.fused_err) {
        int mp = 0;
        for (int j = 0; j < 6; ++j) {
            const float cj = (float)kCErr[j];
            if (cj == 0.f) continue;
            kp[mp] = kall[j];
            cp[mp] = dts * cj;
            ++mp;
        }
        cp[mp] = dts * (float)kCErr[6];
        t->nfe++;
        RkOpt opt = {};
        if (t->keep_s && both && rhs_adj_supported(&t->A, t->H, fl, NDCN_RK_ERROR, mp)) {
            if ((rc = panel(t, &S[7]))) return rc;
            opt.s_out = S[7];
        }
        rc = rhs_rk_f32(&t->A, u[7], nullptr, t->A.n_cols, t->W, t->b, k[6], t->work, t->H, fl | (t->packed ? NDCN_F_PACKED : 0u),
                        NDCN_RK_ERROR, a.y0, kp, cp, mp, nullptr, (float)t->rtol, (float)t->atol, t->d_red, t->d_ws2, st,
                        opt.s_out ? &opt : nullptr);
        if (rc) return rc;
    } else {
        rc = rhs_plain(t, u[7], k[6], st);
        if (rc) return rc;
        m = terms(dts, kCErr, 7, kall, kp, cp, nullptr);
        rc = rk_error_f32(a.y0, u[7], kp, cp, m, (float)t->rtol, (float)t->atol, t->n, t->d_red, t->d_ws, st);
        if (rc) return rc;
    }
    rc = fetch_record(t, st, hs);
    if (rc) return rc;
    a.ratio = (float)(hs->h[0] / (double)t->n);
    bad_out = hs->h[1];
    for (int e = 2; e <= 7; ++e) a.u[e] = u[e], a.S[e] = S[e];
    for (int j = 1; j < 7; ++j) a.k[j] = k[j];
    return NDCN_OK;
}

// ---------------------------------------------------------------------------------------------------- reverse pass helpers
struct Bwd {
    ndcn_tape *t;
    hipStream_t st;
    float *gW_acc = nullptr, *gb_acc = nullptr;
    bool have_w = false;
    float *tmpS = nullptr, *tmpG = nullptr;      // S = A x ; gS / gZ
};

int add_into(float *acc, const float *x, int64_t n, hipStream_t st) {
    const float *kp[1] = {x};
    const float one[1] = {1.f};
    return rk_combine_f32(acc, acc, kp, one, 1, n, st);
}

// (g_X into gx (nullable: not wanted); g_W, g_b accumulated) of K = f(X) for the upstream gradient g: autograd_ops.rhs_vjp
// premasked: g already IS gZ = g (.) [K > 0] (rk_pull_f32 applied the mask where it formed g)
int rhs_vjp(Bwd &B, const float *X, const float *K, const float *g, float *gx, const float *S_kept = nullptr, bool premasked = false) {
    ndcn_tape *t = B.t;
    const bool no_graph = t->flags & NDCN_F_NO_GRAPH, no_control = t->flags & NDCN_F_NO_CONTROL;
    const float *mask = ((t->flags & NDCN_F_RELU) && !premasked) ? K : nullptr;
    int rc;
    const float *gS = nullptr;
    if (!no_control) {
        const float *S = X;
        if (!no_graph && S_kept) {
            S = S_kept;
        } else if (!no_graph) {
            rc = spmm_f32(&t->A, X, nullptr, t->A.n_cols, B.tmpS, t->H, 1.f, 0, B.st);
            if (rc) return rc;
            S = B.tmpS;
        }
        float *gs_out = gx ? (no_graph ? gx : B.tmpG) : nullptr;
        // g_W / g_b: this evaluation's + what the later evaluations sent (autograd_path._add_carried), in the launch that sums the chunks
        rc = linear_bwd_f32(g, mask, S, t->W, gs_out, B.gW_acc, t->b ? B.gb_acc : nullptr, t->bwork, t->n_rows, t->H, t->H, B.st,
                            t->bpacked ? NDCN_F_PACKED : 0u, 1.f, B.have_w);
        if (rc) return rc;
        if (gs_out && t->H == 256) t->bpacked = true;
        B.have_w = true;
        gS = gs_out;
    } else if (gx) {
        float *gs_out = no_graph ? gx : B.tmpG;
        if (mask) {
            rc = relu_bwd_f32(gs_out, g, mask, t->n, B.st);
            if (rc) return rc;
            gS = gs_out;
        } else if (no_graph) {
            rc = copy_f32(gx, g, t->n, B.st);
            if (rc) return rc;
            gS = gx;
        } else {
            gS = g;
        }
    }
    if (gx && !no_graph) {
        rc = spmm_f32(&t->At, gS, nullptr, t->At.n_cols, gx, t->H, 1.f, 0, B.st);
        if (rc) return rc;
    }
    return NDCN_OK;
}

// out = base (nullable) + sum_i c_i p_i over the non-null p_i; returns the result pointer through `res` (nullptr: nothing to add;
// base itself when there are no terms)
// must_own: the result has to live in `out` (it outlives the scratch panels `base` may point into)
int gather_sum(ndcn_tape *t, float *out, const float *base, const float *const *p, const float *c, int n_p, hipStream_t st, const float **res,
               bool must_own = false) {
    const float *kp[8];
    float cp[8];
    int m = 0;
    for (int i = 0; i < n_p; ++i)
        if (p[i] && c[i] != 0.f) {
            kp[m] = p[i];
            cp[m] = c[i];
            ++m;
        }
    if (m == 0) {
        if (must_own && base && base != out) {
            NDCN_HIP(hipMemcpyAsync(out, base, (size_t)t->n * sizeof(float), hipMemcpyDeviceToDevice, st));
            base = out;
        }
        *res = base;
        return NDCN_OK;
    }
    int rc = rk_combine_f32(out, base, kp, cp, m, t->n, st);
    *res = out;
    return rc;
}

inline double half_on_tie_gt(double a, double b) { return a > b ? 1.0 : (a == b ? 0.5 : 0.0); }    // share of max(a, b)'s gradient that a receives

}  // namespace

// ====================================================================================================== C ABI
extern "C" {

int ndcn_tape_dopri5_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, const float *y0,
                         const double *ticks, int64_t n_t, double rtol, double atol, const double *opts, float *out, ndcn_alloc_fn alloc,
                         void *alloc_ctx, ndcn_tape **tape, void *stream) {
    NDCN_CHECK_ARG(A && y0 && ticks && n_t >= 1 && out && alloc && tape && opts && H > 0, "bad argument");
    const bool no_graph = flags & NDCN_F_NO_GRAPH, no_control = flags & NDCN_F_NO_CONTROL;
    NDCN_CHECK_ARG(no_graph || (At && A->n_rows == A->n_cols), "a square operator and its transpose are required");
    NDCN_CHECK_ARG(no_control || W, "weight missing");
    for (int64_t i = 1; i < n_t; ++i) NDCN_CHECK_ARG(ticks[i] > ticks[i - 1], "t must be strictly increasing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HostScratch *hs;
    int rc = host_scratch(&hs);
    if (rc) return rc;
    ndcn_tape *t = new (std::nothrow) ndcn_tape();
    if (!t) { set_error("out of host memory"); return NDCN_EINVAL; }
    *tape = t;                                     // (the caller destroys it on every path, errors included)
    t->A = *A;
    if (At) t->At = *At;
    t->W = W;
    t->b = b;
    t->H = H;
    t->flags = flags;
    t->n_rows = A->n_rows;
    t->n = A->n_rows * (int64_t)H;
    t->rtol = rtol;
    t->atol = atol;
    t->first_given = opts[0] != 0.0;
    t->safety = opts[1];
    t->ifactor = opts[2];
    t->dfactor = opts[3];
    t->max_steps = (int64_t)opts[4];
    t->keep_s = opts[5] != 0.0;
    t->alloc = alloc;
    t->alloc_ctx = alloc_ctx;
    t->panel_bytes = (size_t)t->n * sizeof(float) + 16;
    t->ticks.assign(ticks, ticks + n_t);
    void *p;
    if (hs->rec_dev) {
        t->d_red = hs->rec_dev;                    // the record lands in pinned host memory, the host polls (hostrec.h)
    } else {
        if ((rc = arena(t, 2 * sizeof(double) + 256, &p))) return rc;
        t->d_red = static_cast<double *>(p);
    }
    if ((rc = arena(t, (size_t)reduce_ws_bytes(), &t->d_ws))) return rc;
    if ((rc = arena(t, (size_t)reduce_ws_bytes(), &t->d_ws2))) return rc;
    const int64_t wb = rhs_work_bytes(t->n_rows, H, flags);
    if (wb > 0) {
        if ((rc = arena(t, (size_t)wb, &p))) return rc;
        t->work = static_cast<float *>(p);
        if (H == 256 && !no_control && !no_graph) {
            // the packed image of W once per solve (every evaluation reads it: NDCN_F_PACKED)
            if ((rc = pack_weight_256(W, t->work, st))) return rc;
            if (weights_wide_range(t->work)) t->keep_s = false;       // range guard: the fp32 route has no S output (rhs.hip)
            t->packed = true;
        }
    }
    // ---- dopri5.py:76-83
    t->y_in = y0;
    NDCN_HIP(hipMemcpyAsync(out, y0, (size_t)t->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    float *f0;
    if ((rc = panel(t, &f0))) return rc;
    if ((rc = rhs_plain(t, y0, f0, st))) return rc;
    t->f0 = f0;
    t->nfe = 2;                                    // (the count the python path reports: autograd_path.py nfe = 2)
    int64_t pending_bad = 0;
    double dt;
    if (!t->first_given) {
        if ((rc = initial_step(t, st, hs, pending_bad))) return rc;
        t->nfe = 2;
        dt = t->dt_init;
    } else {
        dt = 0.01;                                 // dopri5.py:82
    }
    const float *y_cur = y0, *f_cur = f0;
    double t_lo = ticks[0], t_hi = ticks[0];
    int last_acc = -1;
    int64_t i = 0;
    while (i + 1 < n_t) {
        ++i;
        const double nxt = ticks[i];
        int64_t n_steps = 0;
        while (nxt > t_hi) {
            if (n_steps >= t->max_steps) { set_error("max_num_steps exceeded (%lld>=%lld)", (long long)n_steps, (long long)t->max_steps); return NDCN_EMAXSTEPS; }
            const double t0 = t_hi;
            if (!(t0 + dt > t0)) { set_error("underflow in dt %g", dt); return NDCN_EUNDERFLOW; }
            if (pending_bad != 0) { set_error("non-finite values in state `y`: %lld elements", (long long)pending_bad); return NDCN_ENONFINITE; }
            t->attempts.emplace_back();
            Attempt &a = t->attempts.back();
            a.t0 = t0;
            a.dt = dt;
            a.dts = (float)dt;
            a.y0 = y_cur;
            a.k[0] = f_cur;
            double bad = 0;
            if ((rc = forward_attempt(t, a, st, hs, bad))) return rc;
            const float ratio = a.ratio;
            a.accept = ratio <= 1.f;
            // misc.py:160-170 as the python path evaluates it (float64 scalars; sqrt in the ratio's float32)
            double dt_next;
            if (ratio == 0.f) {
                dt_next = dt * t->ifactor;
                a.factor = 0;
            } else {
                const double dfac = ratio < 1.f ? 1.0 : t->dfactor;
                const double er = (double)sqrtf(ratio);
                const double expo = (double)0.2f;
                a.factor = nan_max(1.0 / t->ifactor, nan_min(pow(er, expo) / t->safety, 1.0 / dfac));
                dt_next = dt / a.factor;
            }
            a.dt_next = dt_next;
            const double row[5] = {t0, dt, a.accept ? 1.0 : 0.0, (double)ratio, dt_next};
            t->log.insert(t->log.end(), row, row + 5);
            if (a.accept) {
                y_cur = a.u[7];
                f_cur = a.k[6];
                t_lo = t0;
                t_hi = t0 + dt;
                pending_bad = (int64_t)bad;
                last_acc = (int)t->attempts.size() - 1;
            } else {
                t_lo = t_hi = t0;
            }
            dt = dt_next;
            ++n_steps;
        }
        // interp.py:51-65: this tick and the following ones the accepted step covers (<= 7 per pass)
        if (last_acc < 0) { set_error("invalid interpolation: no accepted step covers t = %g", nxt); return NDCN_EINVAL; }
        const Attempt &a = t->attempts[(size_t)last_acc];
        DenseGroup g;
        g.attempt = last_acc;
        g.a0 = (float)t_lo;
        g.a1 = (float)t_hi;
        float xp[7 * 5];
        float *outs[7];
        for (;;) {
            const float at = (float)ticks[i];
            if (!(g.a0 <= at && at <= g.a1)) { set_error("invalid interpolation, fails `t0 <= t <= t1`: %g, %g, %g", (double)g.a0, (double)at, (double)g.a1); return NDCN_EINVAL; }
            const float x = (at - g.a0) / (g.a1 - g.a0);
            const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
            float *q = xp + 5 * g.nt;
            q[0] = x4; q[1] = x3; q[2] = x2; q[3] = x; q[4] = 1.f;
            outs[g.nt] = out + (size_t)i * t->n;
            g.tick[g.nt] = (int)i;
            g.x[g.nt] = x;
            ++g.nt;
            if (i + 1 < n_t && g.nt < 7 && !(ticks[i + 1] > t_hi)) ++i;
            else break;
        }
        float cm[7];
        for (int j = 0; j < 7; ++j) cm[j] = a.dts * (float)kCMid[j];
        rc = interp_direct_multi_f32(a.y0, a.u[7], a.k, cm, a.dts, xp, outs, g.nt, t->n, st);
        if (rc) return rc;
        t->attempts[(size_t)last_acc].dense.push_back((int)t->groups.size());
        t->groups.push_back(g);
    }
    return NDCN_OK;
}

int64_t ndcn_tape_steplog(const ndcn_tape *t, double *rows, int64_t cap) {
    if (!t) return NDCN_EINVAL;
    const int64_t n = (int64_t)t->log.size() / 5;
    if (rows)
        for (int64_t i = 0; i < n && i < cap; ++i) memcpy(rows + 5 * i, t->log.data() + 5 * i, 5 * sizeof(double));
    return n;
}

int64_t ndcn_tape_nfe(const ndcn_tape *t) { return t ? t->nfe : NDCN_EINVAL; }

void ndcn_tape_destroy(ndcn_tape *t) { delete t; }

int ndcn_tape_backward_f32(ndcn_tape *t, const float *g_out, float *g_y0, float *g_W, float *g_b, void *stream) {
    NDCN_CHECK_ARG(t && g_out && g_y0, "bad argument");
    const bool no_control = t->flags & NDCN_F_NO_CONTROL;
    NDCN_CHECK_ARG(no_control || (g_W && (g_b || !t->b)), "g_W / g_b missing");
    // The reverse pass may run more than once over one forward record (loss.backward(retain_graph=True), several torch.autograd.grad
    // calls, a retry after a failed pass): the record is only read; the scratch of a pass is carved from the arena at the mark the FIRST
    // pass found, the caller frees the blocks a pass asked for when it returns (the forward record's blocks live until ndcn_tape_destroy).
    if (!t->bwd_marked) {
        t->bwd_marked = true;
        t->mark_chunk = t->chunk;
        t->mark_left = t->chunk_left;
    } else {
        t->chunk = t->mark_chunk;
        t->chunk_left = t->mark_left;
    }
    t->bpacked = false;                            // (the packed W^T lives in this pass's scratch)
    hipStream_t st = static_cast<hipStream_t>(stream);
    HostScratch *hs;
    int rc = host_scratch(&hs);
    if (rc) return rc;
    const int64_t n = t->n;
    const int H = t->H;
    Bwd B;
    B.t = t;
    B.st = st;
    void *p;
    // ---- scratch
    if (hs->h_dev) {
        t->d_dots = hs->h_dev;
    } else {
        if ((rc = arena(t, (size_t)kDotSlots * 8 * sizeof(double), &p))) return rc;
        t->d_dots = static_cast<double *>(p);
    }
    if ((rc = arena(t, (size_t)rk_bwd_ws_bytes(), &t->d_bws))) return rc;
    if (!no_control) {
        if ((rc = arena(t, (size_t)linear_bwd_work_bytes(t->n_rows, H, H), &t->bwork))) return rc;
        B.gW_acc = g_W;
        B.gb_acc = g_b;
    }
    float *own_gk[7], *own_gy0, *own_gy1, *GU[8] = {}, *T, *CY[2], *CF[2];
    for (int j = 0; j < 7; ++j)
        if ((rc = panel(t, &own_gk[j]))) return rc;
    if ((rc = panel(t, &own_gy0)) || (rc = panel(t, &own_gy1)) || (rc = panel(t, &T))) return rc;
    for (int e = 2; e <= 7; ++e)
        if ((rc = panel(t, &GU[e]))) return rc;
    for (int q = 0; q < 2; ++q)
        if ((rc = panel(t, &CY[q])) || (rc = panel(t, &CF[q]))) return rc;
    if ((rc = panel(t, &B.tmpS)) || (rc = panel(t, &B.tmpG))) return rc;

    const float *Gy = nullptr, *Gf = nullptr;      // gradients of the state / its derivative as they stand behind the attempt being processed
    int pp = 0;
    double g_dt_next = 0.0, g_t = 0.0;             // adjoints of the step size / the time the NEXT attempt started from
    const float rtol = (float)t->rtol, atol = (float)t->atol;

    for (int s = (int)t->attempts.size() - 1; s >= 0; --s) {
        const Attempt &a = t->attempts[(size_t)s];
        int slot = 0;
        if (hs->h_dev) rec_arm(hs->h, kDotSlots * 8);
        auto dots_at = [&](int sl) { return t->d_dots + 8 * sl; };
        // ---- scalar chain, part 1: the error ratio's gradient comes from the NEXT step size (misc.py:160-170)
        double g_dt = 0.0;                         // adjoint of this attempt's dt (float64)
        float g_r = 0.f;
        if (a.ratio == 0.f) {
            g_dt += g_dt_next * t->ifactor;
        } else {
            g_dt += g_dt_next / a.factor;
            const double g_factor = -g_dt_next * a.dt / (a.factor * a.factor);
            const double dfac = a.ratio < 1.f ? 1.0 : t->dfactor;
            const double er = (double)sqrtf(a.ratio), expo = (double)0.2f;
            const double Bv = pow(er, expo) / t->safety, Cv = 1.0 / dfac, Av = 1.0 / t->ifactor;
            const double inner = nan_min(Bv, Cv);
            // torch.max(A, min(B, C)): maximum / minimum split the gradient evenly on ties
            const double share_inner = half_on_tie_gt(inner, Av);
            const double share_B = Bv < Cv ? 1.0 : (Bv == Cv ? 0.5 : 0.0);
            const double g_B = g_factor * share_inner * share_B;
            const double g_er = g_B * expo * pow(er, expo - 1.0) / t->safety;
            g_r = (float)g_er * (0.5f / sqrtf(a.ratio));         // sqrt's backward in the ratio's float32
        }
        // ---- received gradients
        const float *cur_gk[7] = {}, *cur_gy0 = nullptr, *cur_gy1 = nullptr;
        if (a.accept) {
            cur_gy1 = Gy;
            cur_gk[6] = Gf;
        } else {
            cur_gy0 = Gy;
            cur_gk[0] = Gf;
        }
        // ---- dense output of the ticks this step covers (interp.py:21-65)
        // (a step that covers more ticks than the slots hold - a dense time grid over a long accepted step - hands its inner products
        // over in batches: read, added to the scalar adjoints, slots re-armed)
        double g_dts = 0.0;                        // adjoint of the float32 step size dts = float(dt)
        double g_a0 = 0.0, g_a1 = 0.0;
        int dense_slot[kDenseBatch], dense_group[kDenseBatch], n_dense = 0;
        auto take_dense = [&](const double *h) {
            for (int di = 0; di < n_dense; ++di) {
                const DenseGroup &g = t->groups[(size_t)dense_group[di]];
                const double *d = h + 8 * dense_slot[di];
                g_dts += (double)(float)d[7];
                const float w = g.a1 - g.a0;
                for (int q = 0; q < g.nt; ++q) {
                    const float gx = (float)d[q];
                    // x = (at - a0) / (a1 - a0):  dx/da0 = (x - 1) / (a1 - a0),  dx/da1 = -x / (a1 - a0)
                    g_a0 += (double)(gx * ((g.x[q] - 1.f) / w));
                    g_a1 += (double)(gx * (-g.x[q] / w));
                }
            }
            n_dense = 0;
        };
        for (int gi = (int)a.dense.size() - 1; gi >= 0; --gi) {
            const DenseGroup &g = t->groups[(size_t)a.dense[(size_t)gi]];
            const float *hg[7];
            for (int q = 0; q < g.nt; ++q) hg[q] = g_out + (size_t)g.tick[q] * n;
            if (n_dense == kDenseBatch) {
                rc = fetch(t, t->d_dots, 8 * slot, st, hs);
                if (rc) return rc;
                take_dense(hs->h);
                slot = 0;
                if (hs->h_dev) rec_arm(hs->h, kDotSlots * 8);
            }
            rc = rk_dense_bwd_multi_f32(hg, g.nt, a.y0, a.u[7], a.k, a.dts, g.x, own_gy0, own_gy1, own_gk, cur_gy0, cur_gy1, cur_gk,
                                        dots_at(slot), t->d_bws, n, st);
            if (rc) return rc;
            dense_group[n_dense] = a.dense[(size_t)gi];
            dense_slot[n_dense++] = slot++;
            cur_gy0 = own_gy0;
            cur_gy1 = own_gy1;
            for (int j = 0; j < 7; ++j) cur_gk[j] = own_gk[j];
        }
        // ---- error ratio (misc.py:146-157)
        int err_slot = -1, err_idx[8], err_m = 0;
        float err_c[8];
        if (g_r != 0.f) {
            const float *kp[8];
            float *hgk[8];
            const float *hacc[8];
            err_m = terms(a.dts, kCErr, 7, a.k, kp, err_c, err_idx);
            for (int q = 0; q < err_m; ++q) {
                hgk[q] = own_gk[err_idx[q]];
                hacc[q] = cur_gk[err_idx[q]];
            }
            err_slot = slot++;
            rc = rk_error_bwd_f32(a.y0, a.u[7], kp, err_c, err_m, rtol, atol, g_r, 1.0 / (double)n, own_gy0, own_gy1, hgk, cur_gy0, cur_gy1,
                                  hacc, dots_at(err_slot), t->d_bws, n, st);
            if (rc) return rc;
            cur_gy0 = own_gy0;
            cur_gy1 = own_gy1;
            for (int q = 0; q < err_m; ++q) cur_gk[err_idx[q]] = own_gk[err_idx[q]];
        }
        // ---- evaluations 7 .. 2 and the stage sums, pull form
        const float *gu[8] = {};
        {
            const float *g7 = cur_gk[6];
            if (g7) {
                rc = rhs_vjp(B, a.u[7], a.k[6], g7, GU[7], a.S[7]);
                if (rc) return rc;
                if (cur_gy1) {
                    if ((rc = add_into(GU[7], cur_gy1, n, st))) return rc;     // gx + what y1 received (a + b in float32: order-free)
                }
                gu[7] = GU[7];
            } else {
                gu[7] = cur_gy1;
            }
        }
        int row_slot[8];
        for (int m = 2; m <= 7; ++m) row_slot[m] = -1;
        static const bool fused_pull = [] { const char *e = getenv("NDCN_TAPE_PULL_FUSED"); return !(e && e[0] == '0'); }();
        for (int e = 6; e >= 1; --e) {
            // total gradient of k_e: received + sum_{m' > e} dt beta[m' - 2][e - 1] g_u_m', and the step size through stage sum
            // m = e + 1 (all its g_u is known now): <g_u_m, u_m - y0> / dt
            const int m = e + 1;
            const float *pp_[8];
            float cc_[8];
            int np = 0;
            for (int m2 = e + 1; m2 <= 7; ++m2) {
                pp_[np] = gu[m2];
                cc_[np] = a.dts * (float)kBeta[m2 - 2][e - 1];
                ++np;
            }
            float *dst = (e == 1) ? CF[1 - pp] : T;
            const float *tot;
            bool premasked = false;
            if (fused_pull && gu[m]) {
                // one pass: the sum, the ReLU mask of evaluation e (none for k_1: that evaluation belongs to the previous attempt), the product
                const float *kp[8];
                float cp[8];
                int mm = 0;
                for (int q = 0; q < np; ++q)
                    if (pp_[q] && cc_[q] != 0.f) { kp[mm] = pp_[q]; cp[mm] = cc_[q]; ++mm; }
                const float *mask = (e > 1 && (t->flags & NDCN_F_RELU)) ? a.k[e - 1] : nullptr;
                row_slot[m] = slot++;
                rc = rk_pull_f32(dst, cur_gk[e - 1], kp, cp, mm, mask, a.u[m], a.y0, dots_at(row_slot[m]), t->d_bws, n, st);
                if (rc) return rc;
                tot = dst;
                premasked = mask != nullptr;
            } else {
                if (gu[m]) {
                    row_slot[m] = slot++;
                    rc = rk_dot_diff_f32(gu[m], a.u[m], a.y0, dots_at(row_slot[m]), t->d_bws, n, st);
                    if (rc) return rc;
                }
                rc = gather_sum(t, dst, cur_gk[e - 1], pp_, cc_, np, st, &tot, e == 1);
                if (rc) return rc;
            }
            if (e == 1) {
                Gf = tot;
                break;
            }
            if (tot) {
                rc = rhs_vjp(B, a.u[e], a.k[e - 1], tot, GU[e], a.S[e], premasked);
                if (rc) return rc;
                gu[e] = GU[e];
            }
        }
        {
            const float *pp_[8];
            float cc_[8];
            int np = 0;
            for (int m2 = 2; m2 <= 7; ++m2) {
                pp_[np] = gu[m2];
                cc_[np] = 1.f;
                ++np;
            }
            const float *tot;
            rc = gather_sum(t, CY[1 - pp], cur_gy0, pp_, cc_, np, st, &tot, true);
            if (rc) return rc;
            Gy = tot;
        }
        pp = 1 - pp;
        // ---- scalar chain, part 2: one read of the attempt's inner products
        if (slot > 0) {
            rc = fetch(t, t->d_dots, 8 * slot, st, hs);
            if (rc) return rc;
            const double *h = hs->h;
            take_dense(h);
            if (err_slot >= 0) {
                const double *d = h + 8 * err_slot;
                for (int q = 0; q < err_m; ++q) g_dts += (double)((float)((double)g_r * d[q]) * (float)kCErr[err_idx[q]]);
            }
            for (int m = 2; m <= 7; ++m)
                if (row_slot[m] >= 0) g_dts += (double)(float)(h[8 * row_slot[m]] / (double)a.dts);
        }
        g_dt += g_dts;
        double g_t0 = g_t;                         // t_next = t0 (+ dt): the time passes through
        if (a.accept) {
            g_dt += g_t + g_a1;                    // t_hi = t0 + dt feeds the next attempt's t0 and the abscissae's a1
            g_t0 += g_a1 + g_a0;
        }
        g_t = g_t0;
        g_dt_next = g_dt;
    }

    // ---- the initial step and the first evaluation (misc.py:84-143, dopri5.py:76-83)
    // gradients so far: Gy (state the first attempt started from), Gf (its derivative f0), g_dt_next (the initial step size)
    const float *acc_y = Gy;                       // everything y0 receives is added up in `accY`
    float *accY = own_gy0, *accF = own_gk[0];
    const float *acc_f = Gf;
    auto add_y = [&](const float *x) -> int {
        if (!x) return NDCN_OK;
        const float *kp[1] = {x};
        const float one[1] = {1.f};
        int r = rk_combine_f32(accY, acc_y, kp, one, 1, n, st);      // (acc_y null: a copy)
        acc_y = accY;
        return r;
    };
    auto add_f = [&](const float *x, float c) -> int {
        if (!x) return NDCN_OK;
        const float *kp[1] = {x};
        const float cc[1] = {c};
        int r = rk_combine_f32(accF, acc_f, kp, cc, 1, n, st);
        acc_f = accF;
        return r;
    };
    if ((rc = add_y(g_out))) return rc;            // the trajectory's first tick IS y0
    if (!t->first_given && g_dt_next != 0.0) {
        const float g_dt0 = (float)g_dt_next;      // dt = min(100 h0, h1).to(float64)
        const float h100 = 100.f * t->h0;
        const float g_h100 = g_dt0 * (float)(h100 < t->h1 ? 1.0 : (h100 == t->h1 ? 0.5 : 0.0));
        const float g_h1 = g_dt0 * (float)(t->h1 < h100 ? 1.0 : (h100 == t->h1 ? 0.5 : 0.0));
        float g_h0 = 100.f * g_h100;
        float g_d1 = 0.f, g_d2 = 0.f;
        if (t->h1_alt) {
            const float b1 = t->h0 * 1e-3f;
            g_h0 += 1e-3f * g_h1 * (float)half_on_tie_gt((double)b1, (double)1e-6f);
        } else {
            const float m = t->d1 >= t->d2 ? t->d1 : t->d2;
            const float av = (1.0f / m) * 0.01f;
            const float g_av = g_h1 * (float)(0.2 * pow((double)av, 0.2 - 1.0));
            const float g_m = -g_av * 0.01f / (m * m);
            if (t->d1 >= t->d2) g_d1 += g_m;
            else g_d2 += g_m;
        }
        // d2 = rms(f1 - f0 over scale(y0)) / h0
        const float g_rms2 = g_d2 / t->h0;
        g_h0 += -g_d2 * t->d2r / (t->h0 * t->h0);
        float *ga = own_gk[1], *gb_ = own_gk[2], *gy = own_gk[3], *gyh = own_gk[4];
        bool have_yh = false;
        if (g_rms2 != 0.f && t->s2 > 0) {
            const float coef = (float)((double)g_rms2 / (sqrt(t->s2) * sqrt((double)n)));
            rc = rk_rms_bwd_f32(t->f1, t->f0, t->y_in, rtol, atol, coef, ga, gb_, gy, n, st);
            if (rc) return rc;
            if ((rc = add_f(gb_, 1.f)) || (rc = add_y(gy))) return rc;
            // f1 = f(y0 + h0 f0)
            rc = rhs_vjp(B, t->yh, t->f1, ga, gyh);
            if (rc) return rc;
            have_yh = true;
        }
        if (have_yh) {
            if (hs->h_dev) rec_arm(hs->h, 8);
            rc = rk_dot_diff_f32(gyh, t->f0, nullptr, t->d_dots, t->d_bws, n, st);
            if (rc) return rc;
            rc = fetch(t, t->d_dots, 8, st, hs);
            if (rc) return rc;
            g_h0 += (float)hs->h[0];
            if ((rc = add_y(gyh)) || (rc = add_f(gyh, t->h0))) return rc;
        }
        float g_d0 = 0.f;
        if (!t->h0_const) {
            g_d0 = g_h0 * 0.01f / t->d1;
            g_d1 += -g_h0 * 0.01f * t->d0 / (t->d1 * t->d1);
        }
        if (g_d1 != 0.f && t->s1 > 0) {
            const float coef = (float)((double)g_d1 / (sqrt(t->s1) * sqrt((double)n)));
            rc = rk_rms_bwd_f32(t->f0, nullptr, t->y_in, rtol, atol, coef, ga, nullptr, gy, n, st);
            if (rc) return rc;
            if ((rc = add_f(ga, 1.f)) || (rc = add_y(gy))) return rc;
        }
        if (g_d0 != 0.f && t->s0 > 0) {
            const float coef = (float)((double)g_d0 / (sqrt(t->s0) * sqrt((double)n)));
            rc = rk_rms_bwd_f32(t->y_in, nullptr, t->y_in, rtol, atol, coef, ga, nullptr, gy, n, st);
            if (rc) return rc;
            if ((rc = add_y(ga)) || (rc = add_y(gy))) return rc;
        }
    }
    // f0 = f(y0)
    if (acc_f) {
        rc = rhs_vjp(B, t->y_in, t->f0, acc_f, own_gk[5]);
        if (rc) return rc;
        if ((rc = add_y(own_gk[5]))) return rc;
    }
    if (acc_y) NDCN_HIP(hipMemcpyAsync(g_y0, acc_y, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
    else NDCN_HIP(hipMemsetAsync(g_y0, 0, (size_t)n * sizeof(float), st));
    if (!no_control && !B.have_w) {
        NDCN_HIP(hipMemsetAsync(g_W, 0, (size_t)H * H * sizeof(float), st));
        if (g_b) NDCN_HIP(hipMemsetAsync(g_b, 0, (size_t)H * sizeof(float), st));
    }
    return NDCN_OK;
}


// ====================================================================================================== fixed grids
// FixedGridODESolver.integrate (solvers.py:79-99; fixed_grid.py:7-29, rk_common.py:72-78) over ODEFunc, differentiated as the drivers
// train (heat_dynamics.py:313-334: plain backpropagation through every step), as two calls.  The launches and their order are those of
// _impl/odeint.py::_FixedGridSolve (kept as the A/B partner: results are bit-identical): forward - every evaluation carries the stage
// algebra that consumes it in its epilogue (ndcn_rhs_rk_f32), only the trajectory is kept; backward - per step, in reverse: the stages
// re-formed from the stored state by the same launches, then the closed-form VJPs (S = A u, the masked Linear backward, A^T with the
// step's factor folded into alpha) and the stage recurrences as one linear-combination launch each.  README-sized RK4 training was 34 ms
// per Adam step in the interpreter's hands (320 evaluations forward, 640 launches backward), midpoint 17 ms.

namespace {

struct Fixed {
    ndcn_tape t;                       // arena + operator + weights (no attempts)
    hipStream_t st;
    int method;
};

int fixed_init(Fixed &F, const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, int method,
               ndcn_alloc_fn alloc, void *alloc_ctx, void *stream) {
    ndcn_tape &t = F.t;
    t.A = *A;
    if (At) t.At = *At;
    t.W = W;
    t.b = b;
    t.H = H;
    t.flags = flags;
    t.n_rows = A->n_rows;
    t.n = A->n_rows * (int64_t)H;
    t.alloc = alloc;
    t.alloc_ctx = alloc_ctx;
    t.panel_bytes = (size_t)t.n * sizeof(float) + 16;
    F.st = static_cast<hipStream_t>(stream);
    F.method = method;
    const int64_t wb = rhs_work_bytes(t.n_rows, H, flags);
    if (wb > 0) {
        void *p;
        int rc = arena(&t, (size_t)wb, &p);
        if (rc) return rc;
        t.work = static_cast<float *>(p);
        if (H == 256 && !(flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL))) {
            if ((rc = pack_weight_256(W, t.work, F.st))) return rc;
            t.packed = true;
        }
    }
    return NDCN_OK;
}

// one step from y into out_y; u / K (nullable arrays of 4): the stage inputs and derivatives, in caller-provided panels
// last_k: the step's last stage derivative is wanted too (the reverse pass needs its sign pattern; the forward pass does not)
int fixed_step(Fixed &F, const float *y, float dt, float *out_y, float *const *u, float *const *K, bool last_k = true) {
    ndcn_tape *t = &F.t;
    const uint32_t fl = t->flags | (t->packed ? NDCN_F_PACKED : 0u);
    auto eval = [&](const float *x, float *k, int mode, const float *const *kp, const float *cp, int n_prev, float *y_next, bool need_k = true) {
        RkOpt opt = {};
        opt.no_k = need_k ? 0 : 1;
        return rhs_rk_f32(&t->A, x, nullptr, t->A.n_cols, t->W, t->b, k, t->work, t->H, fl, mode, y, kp, cp, n_prev, y_next, 0.f, 0.f, nullptr,
                          nullptr, F.st, need_k ? nullptr : &opt);
    };
    if (F.method == NDCN_M_EULER) {
        const float c[1] = {dt};
        return eval(y, K[0], NDCN_RK_COMBINE, nullptr, c, 0, out_y, last_k);          // y + dt k1
    }
    if (F.method == NDCN_M_MIDPOINT) {
        const float c1[1] = {(float)((double)dt / 2.0)}, c2[1] = {dt};
        int rc = eval(y, K[0], NDCN_RK_COMBINE, nullptr, c1, 0, u[1]);               // ym = y + k1 dt / 2
        if (rc) return rc;
        return eval(u[1], K[1], NDCN_RK_COMBINE, nullptr, c2, 0, out_y, last_k);      // y + dt k2
    }
    const float c[1] = {dt};
    const float *x = y;
    for (int i = 0; i < 4; ++i) {
        const float *kp[3] = {K[0], K[1], K[2]};
        int rc = eval(x, K[i], NDCN_RK_RK4, kp, c, i, i == 3 ? out_y : u[i + 1], i < 3 || last_k);
        if (rc) return rc;
        x = u[i + 1 < 4 ? i + 1 : 3];
    }
    return NDCN_OK;
}

}  // namespace

int ndcn_fixed_grid_train_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method, const float *y0,
                              const float *h_dt, int64_t n_ticks, float *out, ndcn_alloc_fn alloc, void *alloc_ctx, void *stream) {
    NDCN_CHECK_ARG(A && y0 && h_dt && n_ticks >= 1 && out && alloc && H > 0, "bad argument");
    NDCN_CHECK_ARG(method == NDCN_M_EULER || method == NDCN_M_MIDPOINT || method == NDCN_M_RK4, "method must be euler, midpoint or rk4");
    NDCN_CHECK_ARG((flags & NDCN_F_NO_CONTROL) || W, "weight missing");
    Fixed F;
    int rc = fixed_init(F, A, nullptr, W, b, H, flags, method, alloc, alloc_ctx, stream);
    if (rc) return rc;
    float *u[4] = {}, *K[4] = {};
    for (int i = 0; i < 4; ++i)
        if ((rc = panel(&F.t, &u[i])) || (rc = panel(&F.t, &K[i]))) return rc;
    const int64_t n = F.t.n;
    NDCN_HIP(hipMemcpyAsync(out, y0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, F.st));
    for (int64_t i = 0; i < n_ticks; ++i)
        if ((rc = fixed_step(F, out + (size_t)i * n, h_dt[i], out + (size_t)(i + 1) * n, u, K, false))) return rc;
    return NDCN_OK;
}

int ndcn_fixed_grid_backward_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, int method,
                                 const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, float *g_y0, float *g_W,
                                 float *g_b, ndcn_alloc_fn alloc, void *alloc_ctx, void *stream) {
    NDCN_CHECK_ARG(A && traj && g_out && h_dt && n_ticks >= 1 && g_y0 && alloc && H > 0, "bad argument");
    NDCN_CHECK_ARG(method == NDCN_M_EULER || method == NDCN_M_MIDPOINT || method == NDCN_M_RK4, "method must be euler, midpoint or rk4");
    const bool no_graph = flags & NDCN_F_NO_GRAPH, no_control = flags & NDCN_F_NO_CONTROL;
    NDCN_CHECK_ARG(no_graph || At, "the transposed operator is required");
    NDCN_CHECK_ARG(no_control || (W && g_W && (g_b || !b)), "weight / g_W / g_b missing");
    Fixed F;
    int rc = fixed_init(F, A, At, W, b, H, flags, method, alloc, alloc_ctx, stream);
    if (rc) return rc;
    ndcn_tape *t = &F.t;
    hipStream_t st = F.st;
    const int64_t n = t->n;
    void *p;
    if (!no_control) {
        if ((rc = arena(t, (size_t)linear_bwd_work_bytes(t->n_rows, H, H), &t->bwork))) return rc;
        NDCN_HIP(hipMemsetAsync(g_W, 0, (size_t)H * H * sizeof(float), st));
        if (g_b) NDCN_HIP(hipMemsetAsync(g_b, 0, (size_t)H * sizeof(float), st));
    }
    float *u[4] = {}, *K[4] = {}, *gu[4] = {}, *a2[2], *gk, *tmpS, *tmpG, *scratch;
    for (int i = 0; i < 4; ++i)
        if ((rc = panel(t, &u[i])) || (rc = panel(t, &K[i])) || (rc = panel(t, &gu[i]))) return rc;
    if ((rc = panel(t, &a2[0])) || (rc = panel(t, &a2[1])) || (rc = panel(t, &gk)) || (rc = panel(t, &tmpS)) || (rc = panel(t, &tmpG)) ||
        (rc = panel(t, &scratch)))
        return rc;
    // alpha J(x)^T g for K = relu(W (A x) + b) into out (_FixedGridSolve._vjp); g_W / g_b += scale * (this evaluation's)
    auto vjp = [&](const float *x, const float *Kx, const float *g, float alpha, float scale, float *out) -> int {
        const float *mask = (flags & NDCN_F_RELU) ? Kx : nullptr;
        const float *gS;
        int r;
        if (no_control) {
            if (mask) {
                if ((r = relu_bwd_f32(tmpG, g, mask, n, st))) return r;
                gS = tmpG;
            } else {
                gS = g;
            }
        } else {
            const float *S = x;
            if (!no_graph) {
                if ((r = spmm_f32(&t->A, x, nullptr, t->A.n_cols, tmpS, H, 1.f, 0, st))) return r;
                S = tmpS;
            }
            r = linear_bwd_f32(g, mask, S, t->W, tmpG, g_W, t->b ? g_b : nullptr, t->bwork, t->n_rows, H, H, st,
                               t->bpacked ? NDCN_F_PACKED : 0u, scale, true);                // gW_tot.add_(gW, alpha = scale)
            if (r) return r;
            if (H == 256) t->bpacked = true;
            gS = tmpG;
        }
        if (no_graph) return scale_f32(out, gS, alpha, n, st);
        return spmm_f32(&t->At, gS, nullptr, t->At.n_cols, out, H, alpha, 0, st);
    };
    auto lincomb = [&](float *out, const float *y0p, std::initializer_list<const float *> ks, std::initializer_list<float> cs) -> int {
        const float *kp[8];
        float cp[8];
        int m = 0;
        for (const float *k : ks) kp[m++] = k;
        m = 0;
        for (float c : cs) cp[m++] = c;
        return rk_combine_f32(out, y0p, kp, cp, m, n, st);
    };
    const float *a = g_out + (size_t)n_ticks * n;
    int pp = 0;
    for (int64_t i = n_ticks - 1; i >= 0; --i) {
        const float dt = h_dt[i];
        const float *y = traj + (size_t)i * n, *gi = g_out + (size_t)i * n;
        u[0] = const_cast<float *>(y);
        if ((rc = fixed_step(F, y, dt, scratch, u, K))) return rc;
        float *a_new = a2[pp];
        if (method == NDCN_M_EULER) {                         // y1 = y + dt k1
            if ((rc = vjp(y, K[0], a, dt, dt, gu[0]))) return rc;
            rc = lincomb(a_new, a, {gu[0], gi}, {1.f, 1.f});
        } else if (method == NDCN_M_MIDPOINT) {               // ym = y + (dt / 2) k1 ; y1 = y + dt k2
            const float h = (float)((double)dt / 2.0);
            if ((rc = vjp(u[1], K[1], a, dt, dt, gu[1]))) return rc;          // dL / d ym
            if ((rc = vjp(y, K[0], gu[1], h, h, gu[0]))) return rc;
            rc = lincomb(a_new, a, {gu[1], gu[0], gi}, {1.f, 1.f, 1.f});
        } else {                                              // the 3/8 rule, rk_common.py:72-78
            const float c8 = (float)((double)dt / 8.0), c38 = (float)(3.0 * (double)c8), d3 = (float)((double)dt / 3.0);
            if ((rc = vjp(u[3], K[3], a, c8, c8, gu[3]))) return rc;          // J4^T (c8 a)
            if ((rc = lincomb(gk, nullptr, {a, gu[3]}, {c38, dt}))) return rc;
            if ((rc = vjp(u[2], K[2], gk, 1.f, 1.f, gu[2]))) return rc;
            if ((rc = lincomb(gk, nullptr, {a, gu[3], gu[2]}, {c38, -dt, dt}))) return rc;
            if ((rc = vjp(u[1], K[1], gk, 1.f, 1.f, gu[1]))) return rc;
            if ((rc = lincomb(gk, nullptr, {a, gu[3], gu[2], gu[1]}, {c8, dt, -d3, d3}))) return rc;
            if ((rc = vjp(y, K[0], gk, 1.f, 1.f, gu[0]))) return rc;
            rc = lincomb(a_new, a, {gu[3], gu[2], gu[1], gu[0], gi}, {1.f, 1.f, 1.f, 1.f, 1.f});
        }
        if (rc) return rc;
        a = a_new;
        pp = 1 - pp;
    }
    NDCN_HIP(hipMemcpyAsync(g_y0, a, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return NDCN_OK;
}

}  // extern "C"
