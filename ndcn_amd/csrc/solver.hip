// Device-resident integrator for the ODEFunc right-hand side.
//
// State, stage derivatives and dense-output coefficients live in HBM for the whole solve; the host runs
// only the scalar control flow of the reference's solvers:
//   fixed grid (euler / midpoint / rk4 3-8 rule)  torchdiffeq/_impl/solvers.py:79-99, fixed_grid.py, rk_common.py:72-78
//   dopri5                                         torchdiffeq/_impl/dopri5.py:58-122, rk_common.py:22-61,
//                                                  misc.py:84-170, interp.py:5-65
// and reads back one 16-byte record {sum r^2, non-finite count} per adaptive step.
// Scalar arithmetic mirrors the reference's dtypes: time/step-size controller in float64, everything
// that the reference forms as a 0-d tensor of the state dtype (dt*beta, stage times, the initial-step
// heuristic, the interpolation abscissa) in float32.
#include <math.h>
#include <stdlib.h>
#include <new>
#include <algorithm>
#include <utility>
#include <vector>

#include "common.h"
#include "dopri5_tableau.h"
#include "hostrec.h"
#include "kernels.h"

using namespace ndcn;

struct ndcn_solver {
    ndcn_solver_desc d;
    int64_t n_rows = 0, n_elem = 0;
    // panels
    float *ycur = nullptr, *ynext = nullptr, *yold = nullptr, *ytmp = nullptr, *work = nullptr;
    float *k[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *ca = nullptr, *cb = nullptr, *cc = nullptr, *cd = nullptr;
    const float *ce = nullptr;     // interpolation "e" = y at the start of the fitted step
    void *slab = nullptr;          // one workspace: caller-provided or hipMalloc'ed here
    size_t slab_bytes = 0, slab_used = 0;
    bool slab_owned = false;
    double *d_red = nullptr;       // device {sum, nonfinite}
    void *d_ws = nullptr;
    void *d_ws2 = nullptr;         // fused2 error partials
    double *h_red = nullptr;       // pinned host mirror
    bool poll = false;             // d_red IS the device alias of h_red: the kernels store the record into host memory, the host polls (hostrec.h)
    hipEvent_t ev = nullptr;
    // scalar state
    bool begun = false;
    bool fused = false;            // H = 256 fused RHS: `work` holds the packed weights
    bool fused2 = false;           // the RK algebra rides in the epilogue of the RHS launches (rhs_epi)
    bool rec_epi = false;          // ... of an SpMM kernel (no_control RHS) instead of the fused MFMA kernel:
    bool wide_epi = false;         //     the group-record kernel, or (no plan) the row kernel
    bool small_epi = false;        // ... of the narrow-panel kernel (H <= 128, rhs_small.hip)
    float *ytmp2 = nullptr;        // second stage-input panel (fused2: a stage's input must outlive its epilogue)
    double t0 = 0, t1 = 0, dt = 0; // dopri5: last interval [t0, t1], next step size
    float tf = 0;                  // fixed grid: current time in the state dtype
    bool fit_pending = false;      // last accepted step not yet fitted
    bool fit_valid = false;
    int evals_in_step = 0;         // ticks already evaluated inside the current accepted step
    float fit_dt = 0;
    bool cur_is_borrowed = false;  // ycur points into a caller buffer (fixed grid)
    float *ycur_own = nullptr;
    float *own3[3] = {};           // dopri5: the solver's three state panels {ycur_own, ynext, yold as allocated}
    const float *borrowed = nullptr;   // dopri5, ndcn_solver_begin_borrowed: the caller's y0 stands in for a state panel ...
    float *spare = nullptr;        // ... and this one of the solver's own takes its place when it leaves the rotation
    int64_t n_attempt = 0, n_accept = 0, n_rhs = 0;
    double last_ratio = 0;
    int64_t pending_bad = 0;       // non-finite elements seen in the state that starts the next step
    std::vector<double> log;       // 5 doubles per attempt
    // fixed-grid hipGraph replay (launch-bound graphs): one captured step, dt in device memory
    bool graph_on = false;
    hipStream_t gstream = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipEvent_t gev_in = nullptr, gev_out = nullptr;
    float *d_dt = nullptr;         // device step size read by the captured stage kernels
    float *d_coef = nullptr, *d_beta = nullptr;   // replayed adaptive step: tableau entries and their dt-scaled image
    float h_beta[64] = {0};
    // node-range sharded graph (ndcn_solver_desc::shard): exchange on a side stream, global reductions
    bool sharded = false;
    ndcn_shard shard;
    int64_t n_own = 0, n_halo = 0, n_send = 0;
    float *halo = nullptr, *pack = nullptr, *sbuf = nullptr;   // halo panel, send buffer, S = A_own X (two-phase)
    hipStream_t cstream = nullptr;
    hipEvent_t ev_x = nullptr, ev_halo = nullptr;
    // row-split shards: the own rows whose first stage input the boundary launches and the exchange read (merged ranges);
    // everything else of it is formed inside the interior launch (RkOpt::xadd)
    std::vector<std::pair<int64_t, int64_t>> xadd_ranges;
    int xadd_block = -1;           // an interior block whose operator has the XADD kernel (-1: none)
    bool packed = false;           // `work` holds the packed weights of this solve
    bool exact32 = false;          // ... and they are outside the split product's guarantee (range guard): fp32 matrix cores, composed stage algebra
    double n_mean = 0;             // element count behind the controller's means (global for a shard)
    int n_coef = 0;
    float *h_dt = nullptr;         // pinned ring of step sizes (async H2D source must stay untouched until consumed)
    int64_t g_step = 0;
};

namespace {

constexpr int kDtRing = 4096;
constexpr int kCoefCap = 64;

inline size_t align_up(size_t v) { return (v + 255u) & ~(size_t)255u; }

int n_panels(const ndcn_solver_desc *d) {
    const int nk = d->method == NDCN_M_DOPRI5 ? 7 : d->method == NDCN_M_RK4 ? 4 : 1;
    int n = 3 + nk;                                       // ycur, ytmp, ytmp2, k[]
    if (d->method == NDCN_M_DOPRI5) n += 6;               // ynext, yold, a, b, c, d
    return n;
}

size_t workspace_bytes(const ndcn_solver_desc *d) {
    const size_t panel = align_up((size_t)d->A.n_rows * (size_t)d->H * sizeof(float) + 16);
    const size_t work = align_up((size_t)rhs_work_bytes(d->A.n_rows, d->H, d->rhs_flags) + 16);
    size_t shard = 0;
    if (d->shard) {
        const size_t rows = (size_t)halo_plan_n_halo(d->shard->halo) + (size_t)halo_plan_n_send(d->shard->halo) +
                            (d->shard->A_own.n_rows > 0 ? (size_t)d->A.n_rows : 0);
        shard = 3 * 256 + align_up(rows * (size_t)d->H * sizeof(float) + 48);
    }
    return (size_t)n_panels(d) * panel + work + 2 * align_up((size_t)reduce_ws_bytes()) + 4096 + shard;
}

int carve(ndcn_solver *s, size_t bytes, void **p) {
    const size_t off = align_up(s->slab_used);
    if (off + bytes > s->slab_bytes) {
        set_error("solver workspace too small: need %zu more bytes", off + bytes - s->slab_bytes);
        return NDCN_EINVAL;
    }
    *p = static_cast<char *>(s->slab) + off;
    s->slab_used = off + bytes;
    return NDCN_OK;
}

int alloc_panel(ndcn_solver *s, float **p) {
    void *q = nullptr;
    int rc = carve(s, (size_t)s->n_elem * sizeof(float) + 16, &q);
    if (rc) return rc;
    *p = static_cast<float *>(q);
    return NDCN_OK;
}

int rhs_sharded(ndcn_solver *s, const float *x, float *K, int mode, const float *y0, const float *const *kp, const float *cp,
                int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const RkOpt *opt);

int rhs(ndcn_solver *s, const float *x, float *out, hipStream_t st) {
    if (s->sharded) return rhs_sharded(s, x, out, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, st, nullptr);
    s->n_rhs++;
    if (s->fused2 && !s->rec_epi && !s->exact32)
        return rhs_fused2_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->work, s->d.b, out, s->d.rhs_flags, 0, nullptr,
                              nullptr, nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, st);
    if (s->fused2 && !s->rec_epi)       // exact32: the range guard's route - the same launch on the fp32 matrix cores
        return rhs_fused2_exact_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->work, s->d.b, out, s->d.rhs_flags, 0, nullptr,
                                    nullptr, nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, st);
    if (s->fused) {     // weights were packed once in solver_begin
        return rhs_fused_packed_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->work, s->d.b, out, s->d.rhs_flags, st);
    }
    return rhs_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->d.W, s->d.b, out, s->work, s->d.H, s->d.rhs_flags, st);
}

// right-hand side whose epilogue carries the stage algebra: the fused MFMA kernel (default RHS, H = 256) or the
// group-record SpMM (no_control RHS)
int rhs_epi(ndcn_solver *s, const float *x, float *K, int mode, const float *y0, const float *const *kp, const float *cp,
            int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st,
            const float *dt_dev = nullptr, const RkOpt *opt = nullptr) {
    if (s->sharded) return rhs_sharded(s, x, K, mode, y0, kp, cp, n_prev, y_next, rtol, atol, d_out, d_ws, st, opt);
    s->n_rhs++;
    if (s->rec_epi) {
        // replay: cp holds the bare tableau entries; they join the solver's coefficient table, whose scaled image
        // (fl(dt * beta), written by the graph's first kernel) the launch reads instead of by-value coefficients
        const float *c_dev = nullptr;
        if (dt_dev) {
            if (s->n_coef + n_prev + 1 > kCoefCap) { set_error("coefficient table full"); return NDCN_EINVAL; }
            for (int m = 0; m <= n_prev; ++m) s->h_beta[s->n_coef + m] = cp[m];
            c_dev = s->d_coef + s->n_coef;
            s->n_coef += 8;                                   // slices stay 32-byte aligned
        }
        g_last_rhs_path = s->small_epi ? NDCN_PATH_SMALL : (s->wide_epi ? NDCN_PATH_WIDE : NDCN_PATH_REC);
        if (s->small_epi)
            return rhs_small_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->d.W, s->d.b, K, s->d.H, s->d.rhs_flags, mode, y0, kp, cp,
                                 n_prev, y_next, rtol, atol, d_out, d_ws, st, c_dev, opt);
        if (s->wide_epi)
            return spmm_wide_rk_f32(&s->d.A, x, nullptr, s->d.A.n_cols, K, s->d.rhs_flags, mode, y0, kp, cp, n_prev, y_next,
                                    rtol, atol, d_out, d_ws, st, c_dev, opt);
        return spmm_rec_f32(&s->d.A, x, nullptr, s->d.A.n_cols, K, 1.f, s->d.rhs_flags, mode, y0, kp, cp, n_prev, y_next, rtol,
                            atol, d_out, d_ws, st, c_dev, opt);
    }
    if (s->exact32)     // range guard: the same launch with the fp32 matrix cores as its consumer (rhs_fused2_exact.hip)
        return rhs_fused2_exact_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->work, s->d.b, K, s->d.rhs_flags, mode, y0, kp, cp, n_prev,
                                    y_next, rtol, atol, d_out, d_ws, st, opt);
    return rhs_fused2_f32(&s->d.A, x, nullptr, s->d.A.n_cols, s->work, s->d.b, K, s->d.rhs_flags, mode, y0, kp, cp, n_prev,
                          y_next, rtol, atol, d_out, d_ws, st, opt);
}

// wait for the reduction record enqueued last on `st`
int fetch_record(ndcn_solver *s, hipStream_t st, double &sum, double &bad) {
    if (s->sharded) {                                   // every rank sees the same record: identical controller decisions
        int rc = comm_allreduce_sum_f64(s->shard.comm, s->d_red, 2, st);
        if (rc) return rc;
    }
    if (s->poll) {
        int rc = rec_wait(s->h_red, 2, st);
        if (rc) return rc;
    } else {
        NDCN_HIP(hipMemcpyAsync(s->h_red, s->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipEventRecord(s->ev, st));
        NDCN_HIP(hipEventSynchronize(s->ev));
    }
    sum = s->h_red[0];
    bad = s->h_red[1];
    return NDCN_OK;
}

// before the launches whose last kernel writes an n-double record (polling mode: the sentinel the host waits on)
inline void arm(ndcn_solver *s, int n = 2) {
    if (s->poll) rec_arm(s->h_red, n);
}

int rms_scaled(ndcn_solver *s, const float *a, const float *b, const float *y, float rtol, float atol, hipStream_t st,
               float &rms, double &bad) {
    arm(s);
    int rc = scaled_sumsq_f32(a, b, y, rtol, atol, s->n_elem, s->d_red, s->d_ws, st);
    if (rc) return rc;
    double sum;
    rc = fetch_record(s, st, sum, bad);
    if (rc) return rc;
    // misc.py:71-76: x.norm() / numel ** 0.5, a float32 0-d tensor divided by a python float
    const float nrm = (float)sqrt(sum);
    rms = nrm / (float)sqrt(s->n_mean);
    return NDCN_OK;
}

// misc.py:84-143 with order = 4 (dopri5.py:80)
int initial_step(ndcn_solver *s, hipStream_t st, double &h_out) {
    const float rtol = (float)s->d.rtol, atol = (float)s->d.atol;
    float d0, d1, d2;
    double bad0, bad;
    int rc;
    static const bool fuse_on = [] { const char *e = getenv("NDCN_INIT_FUSED"); return !(e && e[0] == '0'); }();
    if (fuse_on && (s->sharded || s->n_elem > aten_order_max_elems())) {
        // d0 = || y0 / scale || and d1 = || f0 / scale || in one pass over {y0, f0}, one read-back (large panels: the
        // double-precision sums of scaled_sumsq_f32; small ones keep ATen's float32 order, section 2)
        arm(s, 4);
        rc = scaled_sumsq_pair_f32(s->k[0], s->ycur, rtol, atol, s->n_elem, s->d_red, s->d_ws, s->d_ws2, st);
        if (rc) return rc;
        if (s->sharded) {
            rc = comm_allreduce_sum_f64(s->shard.comm, s->d_red, 4, st);
            if (rc) return rc;
        }
        if (s->poll) {
            if ((rc = rec_wait(s->h_red, 4, st))) return rc;
        } else {
            NDCN_HIP(hipMemcpyAsync(s->h_red, s->d_red, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
            NDCN_HIP(hipEventRecord(s->ev, st));
            NDCN_HIP(hipEventSynchronize(s->ev));
        }
        bad0 = s->h_red[1];
        d0 = (float)sqrt(s->h_red[0]) / (float)sqrt(s->n_mean);
        d1 = (float)sqrt(s->h_red[2]) / (float)sqrt(s->n_mean);
        s->pending_bad = (int64_t)bad0;
    } else {
        rc = rms_scaled(s, s->ycur, nullptr, s->ycur, rtol, atol, st, d0, bad0);
        if (rc) return rc;
        s->pending_bad = (int64_t)bad0;
        rc = rms_scaled(s, s->k[0], nullptr, s->ycur, rtol, atol, st, d1, bad);
        if (rc) return rc;
    }
    float h0;
    if (d0 < 1e-5 || d1 < 1e-5) h0 = 1e-6f;
    else h0 = 0.01f * (d0 / d1);
    // y1 = y0 + h0 * f0 ; f1 = f(t0 + h0, y1)
    const float *kp[1] = {s->k[0]};
    const float cp[1] = {h0};
    // (on the lattice plan the launch that produces f1 forms y0 + h0 f0 on the rows it stages: RkOpt::xadd)
    const bool fuse_d2 = fuse_on && s->fused2 && (s->sharded || s->n_elem > aten_order_max_elems());
    static const bool xadd_on = [] { const char *e = getenv("NDCN_STAGE_XADD"); return !(e && e[0] == '0'); }();
    const bool xadd = xadd_on && fuse_d2 && !s->sharded && !s->rec_epi && !s->exact32 && rhs_xadd_supported(&s->d.A, s->d.H, s->d.rhs_flags, 2, 1);
    if (!xadd) {
        rc = rk_combine_f32(s->ytmp, s->ycur, kp, cp, 1, s->n_elem, st);
        if (rc) return rc;
    }
    // d2 = || (f1 - f0) / scale || (misc.py:131-134).  On the paths whose RHS carries an epilogue the sum rides in the launch
    // that produces f1, as its error record: stages {f0, f1} with coefficients {-1, 1} - (-f0) + f1 is the float32 value of
    // f1 - f0 - over the tolerance of the pair (y0, y0), i.e. atol + rtol |y0|; squares summed in double like
    // scaled_sumsq_f32.  (Panels small enough for ATen's float32 summation order keep the separate launch: section 2.)
    if (fuse_d2) {
        const float *kq[1] = {s->k[0]};
        const float cq[2] = {-1.f, 1.f};
        const RkOpt opt = {s->ycur, 0, nullptr, nullptr, xadd ? s->k[0] : nullptr, xadd ? h0 : 0.f};
        arm(s);
        rc = rhs_epi(s, xadd ? s->ycur : s->ytmp, s->k[1], 2, s->ycur, kq, cq, 1, nullptr, rtol, atol, s->d_red, s->d_ws2, st, nullptr,
                     &opt);
        if (rc) return rc;
        double sum;
        rc = fetch_record(s, st, sum, bad);
        if (rc) return rc;
        d2 = (float)sqrt(sum) / (float)sqrt(s->n_mean);
    } else {
        rc = rhs(s, s->ytmp, s->k[1], st);
        if (rc) return rc;
        rc = rms_scaled(s, s->k[1], s->k[0], s->ycur, rtol, atol, st, d2, bad);
        if (rc) return rc;
    }
    d2 = d2 / h0;
    float h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) {
        const float a = 1e-6f, b = h0 * 1e-3f;
        h1 = a > b ? a : b;
    } else {
        // `(0.01 / m) ** (1 / 5)` as torch evaluates it on a float32 0-d tensor (misc.py:141): python_scalar / tensor is
        // tensor.reciprocal() * scalar - two float32 roundings - and tensor ** python_float runs std::pow in double with
        // the exponent at full double precision, rounded to float32 once
        const float m = d1 > d2 ? d1 : d2;
        const float a = (1.0f / m) * 0.01f;
        h1 = (float)pow((double)a, 1. / 5.);
    }
    const float h100 = 100.f * h0;
    h_out = (double)(h100 < h1 ? h100 : h1);
    if (isnan(h100) || isnan(h1)) h_out = NAN;
    return NDCN_OK;
}

// One right-hand side (+ RK epilogue) of a shard: the halo exchange runs on the side stream while the launches that
// need no halo run on `st`; see struct ndcn_shard for the three forms.  Every launch goes through rhs_rk_f32, i.e. the
// same kernel selection as ndcn_rhs_rk_f32; split launches take their rows of every row-local panel by offset, the error
// record by `y1` + accumulation (NDCN_F_ACCUM semantics).
int rhs_sharded(ndcn_solver *s, const float *x, float *K, int mode, const float *y0, const float *const *kp, const float *cp,
                int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const RkOpt *opt) {
    s->n_rhs++;
    const int H = s->d.H;
    const uint32_t flags = s->d.rhs_flags | (s->packed ? NDCN_F_PACKED : 0u);
    const ndcn_shard &sh = s->shard;
    float *y_aux = opt ? opt->y_aux : nullptr;
    const float *c_aux = opt ? opt->c_aux : nullptr;
    const bool xadd = opt && opt->xadd;                       // x = y0, the input is x + xadd_c * xadd (row-split shards only)
    auto launch = [&](const ndcn_csr *A, const float *X, const float *Xh, int64_t lo, bool first, const float *y1,
                      bool with_xadd = false) -> int {
        const size_t off = (size_t)lo * H;
        if (opt && opt->y1) y1 = opt->y1;                     // the caller names the second state of the error tolerance
        const float *kpo[8];
        for (int m = 0; m < n_prev; ++m) kpo[m] = kp[m] + off;
        RkOpt o = {mode == NDCN_RK_ERROR ? y1 + off : nullptr, (mode == NDCN_RK_ERROR && !first) ? 1 : 0,
                   y_aux ? y_aux + off : nullptr, c_aux, with_xadd ? opt->xadd : nullptr, with_xadd ? opt->xadd_c : 0.f};
        return rhs_rk_f32(A, X, Xh, s->n_own, s->d.W, s->d.b, K + off, s->work, H, flags, mode, y0 ? y0 + off : nullptr, kpo, cp,
                          n_prev, y_next ? y_next + off : nullptr, rtol, atol, d_out, d_ws, st, &o);
    };
    if (s->d.rhs_flags & NDCN_F_NO_GRAPH) return launch(&s->d.A, x, nullptr, 0, true, x);     // row-local: nothing to exchange
    // x is complete on `st` -> the exchange may start on the side stream
    NDCN_HIP(hipEventRecord(s->ev_x, st));
    NDCN_HIP(hipStreamWaitEvent(s->cstream, s->ev_x, 0));
    // xadd: the rows the exchange packs and the boundary launches gather were formed in s->ytmp by the caller
    const float *xs = xadd ? s->ytmp : x;
    int rc = halo_exchange_f32(sh.halo, xs, H, s->pack, s->halo, s->cstream);
    if (rc) return rc;
    NDCN_HIP(hipEventRecord(s->ev_halo, s->cstream));
    if (sh.n_blocks > 0) {
        bool first = true;
        for (int b = 0; b < sh.n_blocks; ++b)
            if (!sh.blocks[b].needs_halo) {
                rc = launch(&sh.blocks[b].A, x, nullptr, sh.blocks[b].row_lo, first, x, xadd);
                if (rc) return rc;
                first = false;
            }
        NDCN_HIP(hipStreamWaitEvent(st, s->ev_halo, 0));
        for (int b = 0; b < sh.n_blocks; ++b)
            if (sh.blocks[b].needs_halo) {
                rc = launch(&sh.blocks[b].A, xs, s->halo, sh.blocks[b].row_lo, first, x);
                if (rc) return rc;
                first = false;
            }
        return NDCN_OK;
    }
    if (sh.A_own.n_rows > 0) {
        rc = spmm_f32(&sh.A_own, x, nullptr, s->n_own, s->sbuf, H, 1.f, 0, st);               // phase 1 under the exchange
        if (rc) return rc;
        NDCN_HIP(hipStreamWaitEvent(st, s->ev_halo, 0));
        return launch(&s->d.A, s->sbuf, s->halo, 0, true, x);                                  // phase 2: [I | A_halo] over [S | halo]
    }
    NDCN_HIP(hipStreamWaitEvent(st, s->ev_halo, 0));
    return launch(&s->d.A, x, s->halo, 0, true, x);
}

void dt_coeffs(float dt32, const double *beta, int n, const float *const *kall, const float **kp, float *cp, int &m) {
    // (scale * x) of misc.py:25 in float32; exact-zero tableau entries are dropped (their product is 0)
    m = 0;
    for (int j = 0; j < n; ++j) {
        const float bj = (float)beta[j];
        if (bj == 0.f) continue;
        kp[m] = kall[j];
        cp[m] = dt32 * bj;
        ++m;
    }
}

// The launches of one attempted step (rk_common.py:41-61): six stage inputs / evaluations and the error record.
// dt_dev == nullptr: dt32 is the step size and rides in the kernel arguments.  dt_dev != nullptr (hipGraph capture):
// dt32 must be 1 - the coefficients passed are the bare tableau entries and every kernel forms fl(dt * c) from the
// device-resident step size, the same single rounding.
int enqueue_attempt(ndcn_solver *s, hipStream_t st, float dt32, const float *dt_dev) {
    const float *kp[8];
    float cp[8];
    int m, rc;
    if (s->fused2) {
        // Stage algebra rides in the RHS epilogues: the evaluation that produces k[i+1] also forms the NEXT stage
        // input y0 + dt * sum_m beta[i+1][m] k[m] (its own K as the last term), the last one the error record.
        // The first stage input y0 + dt beta_21 k1 is a kernel of its own (3 panels) - unless the launch that produces k2 can
        // form it on the rows it stages (RkOpt::xadd: the lattice plan of rhs_fused3.hip, one more gather instead)
        static const bool xadd_on = [] { const char *e = getenv("NDCN_STAGE_XADD"); return !(e && e[0] == '0'); }();
        const bool xadd = xadd_on && !dt_dev && !s->rec_epi && !s->exact32 &&
                          (s->sharded ? s->xadd_block >= 0 : rhs_xadd_supported(&s->d.A, s->d.H, s->d.rhs_flags, 1, 1));
        dt_coeffs(dt32, kBeta[0], 1, s->k, kp, cp, m);
        const float xadd_c = cp[0];
        if (!xadd) {
            rc = rk_combine_f32(s->ytmp, s->ycur, kp, cp, m, s->n_elem, st, dt_dev);
            if (rc) return rc;
        } else if (s->sharded) {
            // row-split shard: the boundary launches and the exchange still read the stage input from a panel - formed on the
            // few lattice rows they touch; the interior launch forms the rest on its staged rows
            for (const auto &r : s->xadd_ranges) {
                const size_t off = (size_t)r.first * s->d.H;
                const float *kq[1] = {s->k[0] + off};
                rc = rk_combine_f32(s->ytmp + off, s->ycur + off, kq, cp, 1, (r.second - r.first) * (int64_t)s->d.H, st, nullptr);
                if (rc) return rc;
            }
        }
        // The launch that produces k6 (i == 4) holds k1, k3, k4, k5 in its epilogue: it also forms the partial error sum
        // E = dt (c_err1 k1 + c_err3 k3 + c_err4 k4 + c_err5 k5 + c_err6 k6) - left to right like the reference's sum - into
        // the stage-input buffer that is free at that point; the error launch reads {y0, E, y1} instead of 7 panels.
        // (Replayed steps keep the one-launch form: their coefficients are fl(dt * c) of a device-resident dt, and E's
        // coefficient in the error launch is the constant 1.)
        static const bool aux_on = [] { const char *e = getenv("NDCN_ERR_PARTIAL"); return !(e && e[0] == '0'); }();
        // Small panels (<= 2^20 elements: the reference's own drivers, every fixture) form the error record in a launch
        // of its own, in the summation order of ATen's float32 mean (rk.hip: rk_error_aten_kernel): at the truth solves'
        // rtol 1e-7 the accept / reject decision hangs on the last bit of that mean.
        const bool split_error = !s->sharded && s->n_elem >= 8 && s->n_elem <= aten_order_max_elems();
        const bool use_aux = aux_on && !dt_dev && !split_error;
        // The launch that produces k4 (i == 2) holds k1, k2, k3 in its epilogue: besides the stage-5 input it writes
        // P = dt (beta_61 k1 + beta_62 k2 + beta_63 k3 + beta_64 k4) - the first four terms of the stage-6 sum, left to right -
        // into the panel that will hold y1 two launches later; the launch that produces k5 then reads {y0, P} and forms
        // y0 + (1 * P + dt beta_65 k5), the same roundings, instead of {y0, k1, k2, k3, k4}: 2 panels less per step.
        static const bool partial_on = [] { const char *e = getenv("NDCN_STAGE_PARTIAL"); return !(e && e[0] == '0'); }();
        const bool use_partial = partial_on && !dt_dev;
        float *e_panel = nullptr, *p_panel = nullptr;
        float *in = xadd ? s->ycur : s->ytmp;
        for (int i = 0; i < 6; ++i) {
            if (i < 5) {
                // stage-6 input IS y1; a sharded XADD launch still gathers its boundary rows from ytmp: its output goes to ytmp2
                float *out = (i == 4) ? s->ynext : ((in == s->ytmp || (i == 0 && xadd && s->sharded)) ? s->ytmp2 : s->ytmp);
                int mp = 0;
                if (i == 3 && p_panel) {
                    kp[0] = p_panel;
                    cp[0] = 1.f;
                    mp = 1;
                } else {
                    for (int j = 0; j <= i; ++j) {             // previous stages k[0..i] with non-zero coefficients
                        const float bj = (float)kBeta[i + 1][j];
                        if (bj == 0.f) continue;
                        kp[mp] = s->k[j];
                        cp[mp] = dt32 * bj;
                        ++mp;
                    }
                }
                cp[mp] = dt32 * (float)kBeta[i + 1][i + 1];   // the K being produced, last term
                RkOpt opt = {nullptr, 0, nullptr, nullptr, nullptr, 0.f};
                float c2[8];
                if (i == 0 && xadd) {
                    opt.xadd = s->k[0];
                    opt.xadd_c = xadd_c;
                }
                if (i == 2 && use_partial) {
                    // rows 4 and 5 of the tableau are dense: the same stages in the same order as this launch's own sum
                    bool dense = true;
                    for (int j = 0; j <= 3; ++j) dense = dense && (float)kBeta[3][j] != 0.f && (float)kBeta[4][j] != 0.f;
                    if (dense) {
                        for (int j = 0; j <= 3; ++j) c2[j] = dt32 * (float)kBeta[4][j];
                        p_panel = s->ynext;
                        opt.y_aux = p_panel;
                        opt.c_aux = c2;
                    }
                }
                if (i == 4 && use_aux) {
                    // the same stages in the same order: beta[5][j] and c_err[j] vanish for the same j (= 1) only
                    int m2 = 0;
                    for (int j = 0; j <= i; ++j) {
                        if ((float)kBeta[i + 1][j] == 0.f) continue;
                        c2[m2++] = dt32 * (float)kCErr[j];
                    }
                    c2[m2] = dt32 * (float)kCErr[i + 1];
                    e_panel = (in == s->ytmp) ? s->ytmp2 : s->ytmp;
                    opt.y_aux = e_panel;
                    opt.c_aux = c2;
                }
                rc = rhs_epi(s, in, s->k[i + 1], 1, s->ycur, kp, cp, mp, out, 0.f, 0.f, nullptr, nullptr, st, dt_dev,
                             (opt.y_aux || opt.xadd) ? &opt : nullptr);
                if (rc) return rc;
                in = out;
            } else if (split_error) {
                rc = rhs(s, in, s->k[6], st);
                if (rc) return rc;
                dt_coeffs(dt32, kCErr, 7, s->k, kp, cp, m);
                rc = rk_error_f32(s->ycur, s->ynext, kp, cp, m, (float)s->d.rtol, (float)s->d.atol, s->n_elem, s->d_red, s->d_ws, st,
                                  dt_dev);
                if (rc) return rc;
            } else {
                int mp = 0;
                if (e_panel) {
                    kp[0] = e_panel;
                    cp[0] = 1.f;
                    mp = 1;
                } else {
                    for (int j = 0; j < 6; ++j) {
                        const float cj = (float)kCErr[j];
                        if (cj == 0.f) continue;
                        kp[mp] = s->k[j];
                        cp[mp] = dt32 * cj;
                        ++mp;
                    }
                }
                cp[mp] = dt32 * (float)kCErr[6];
                rc = rhs_epi(s, in, s->k[6], 2, s->ycur, kp, cp, mp, nullptr, (float)s->d.rtol, (float)s->d.atol, s->d_red,
                             s->d_ws2, st, dt_dev);
                if (rc) return rc;
            }
        }
        return NDCN_OK;
    }
    for (int i = 0; i < 6; ++i) {
        dt_coeffs(dt32, kBeta[i], i + 1, s->k, kp, cp, m);
        float *dst = (i == 5) ? s->ynext : s->ytmp;          // the 6th stage input IS y1 (rk_common.py:54-58)
        rc = rk_combine_f32(dst, s->ycur, kp, cp, m, s->n_elem, st, dt_dev);
        if (rc) return rc;
        rc = rhs(s, dst, s->k[i + 1], st);
        if (rc) return rc;
    }
    dt_coeffs(dt32, kCErr, 7, s->k, kp, cp, m);
    return rk_error_f32(s->ycur, s->ynext, kp, cp, m, (float)s->d.rtol, (float)s->d.atol, s->n_elem, s->d_red, s->d_ws, st,
                        dt_dev);
}

int graph_setup_dopri5(ndcn_solver *s);

// dopri5.py:94-122
int dopri5_step(ndcn_solver *s, hipStream_t st) {
    const double t_start = s->t1, dt = s->dt;
    if (!(t_start + dt > t_start)) {
        set_error("underflow in dt %g", dt);
        return NDCN_EUNDERFLOW;
    }
    if (s->pending_bad > 0) {
        set_error("non-finite values in state `y` (%lld elements)", (long long)s->pending_bad);
        return NDCN_ENONFINITE;
    }
    const float dt32 = (float)dt;
    int rc;
    arm(s);
    if (s->graph_on) {
        // one captured graph per attempt, the step size handed over through device memory
        if (!s->gexec && (rc = graph_setup_dopri5(s))) return rc;
        float *slot = s->h_dt + (s->g_step++ % kDtRing);
        *slot = dt32;
        NDCN_HIP(hipMemcpyAsync(s->d_dt, slot, sizeof(float), hipMemcpyHostToDevice, st));
        NDCN_HIP(hipGraphLaunch(s->gexec, st));
        s->n_rhs += 6;
    } else {
        rc = enqueue_attempt(s, st, dt32, nullptr);
        if (rc) return rc;
    }
    double sum, bad;
    rc = fetch_record(s, st, sum, bad);
    if (rc) return rc;
    // misc.py:156 mean in the state dtype; dopri5.py:109
    const float ratio = (float)(sum / s->n_mean);
    const bool accept = ratio <= 1.f;
    // misc.py:160-170.  safety / dfactor passed through a float32 tensor in the reference (dopri5.py:72-74)
    const double safety = s->d.safety > 0 ? s->d.safety : (double)0.9f;
    const double ifactor = s->d.ifactor > 0 ? s->d.ifactor : 10.0;
    const double dfactor = s->d.dfactor > 0 ? s->d.dfactor : (double)0.2f;
    double dt_next;
    if (ratio == 0.f) {
        dt_next = dt * ifactor;
    } else {
        const double dfac = ratio < 1.f ? 1.0 : dfactor;
        const double er = (double)sqrtf(ratio);
        const double expo = (double)0.2f;
        const double factor = nan_max(1.0 / ifactor, nan_min(pow(er, expo) / safety, 1.0 / dfac));
        dt_next = dt / factor;
    }
    s->n_attempt++;
    s->last_ratio = ratio;
    const double row[5] = {t_start, dt, accept ? 1.0 : 0.0, (double)ratio, dt_next};
    s->log.insert(s->log.end(), row, row + 5);
    if (accept) {
        s->n_accept++;
        // keep {y0, y1, k[0..6]} of this step intact for a lazy dense-output fit; rotate the state
        s->fit_pending = true;
        s->fit_valid = false;
        s->evals_in_step = 0;
        s->fit_dt = dt32;
        s->t0 = t_start;
        s->t1 = t_start + dt;
        s->pending_bad = (int64_t)bad;
    } else {
        s->t0 = s->t1 = t_start;
    }
    s->dt = dt_next;
    return NDCN_OK;
}

// After an accepted step: y0 = ycur, y1 = ynext, f0 = k[0], f1 = k[6].  Rotate for the next step.
int rotate_after_accept(ndcn_solver *s, hipStream_t st) {
    if (s->graph_on) {
        // the captured graph holds fixed pointers: the state moves instead of the names (launch-bound sizes only, the
        // three copies are a few microseconds)
        const size_t bytes = (size_t)s->n_elem * sizeof(float);
        if (s->fit_valid) {                 // the stored fit's "e" is the step's y0: keep it
            NDCN_HIP(hipMemcpyAsync(s->yold, s->ycur, bytes, hipMemcpyDeviceToDevice, st));
            s->ce = s->yold;
        }
        NDCN_HIP(hipMemcpyAsync(s->ycur, s->ynext, bytes, hipMemcpyDeviceToDevice, st));
        NDCN_HIP(hipMemcpyAsync(s->k[0], s->k[6], bytes, hipMemcpyDeviceToDevice, st));   // FSAL
        return NDCN_OK;
    }
    float *old = s->yold;
    if (s->borrowed && old == s->borrowed) {          // the caller's y0 leaves the rotation: never written by the solver
        old = s->spare;
        s->borrowed = nullptr;
        s->spare = nullptr;
    }
    s->yold = s->ycur;       // becomes "e" if this step gets fitted
    s->ycur = s->ynext;
    s->ynext = old;
    float *f = s->k[0];
    s->k[0] = s->k[6];       // FSAL
    s->k[6] = f;
    return NDCN_OK;
}

int do_fit(ndcn_solver *s, hipStream_t st) {
    // called BEFORE the rotation: ycur = y0, ynext = y1
    float cm[7];
    for (int j = 0; j < 7; ++j) cm[j] = s->fit_dt * (float)kCMid[j];
    int rc = interp_fit_f32(s->ycur, s->ynext, s->k, cm, s->fit_dt, s->ca, s->cb, s->cc, s->cd, s->n_elem, st);
    if (rc) return rc;
    return NDCN_OK;
}

}  // namespace

namespace ndcn {

int64_t solver_workspace_bytes(const ndcn_solver_desc *desc) {
    if (!desc || desc->H <= 0 || desc->A.n_rows < 0) return NDCN_EINVAL;
    return (int64_t)workspace_bytes(desc);
}

int solver_create(const ndcn_solver_desc *desc, void *workspace, int64_t ws_bytes, ndcn_solver **out) {
    NDCN_CHECK_ARG(desc && out, "null argument");
    NDCN_CHECK_ARG(desc->method >= NDCN_M_EULER && desc->method <= NDCN_M_DOPRI5, "unknown method");
    NDCN_CHECK_ARG(desc->H > 0, "H must be positive");
    const bool no_graph = desc->rhs_flags & NDCN_F_NO_GRAPH, no_ctl = desc->rhs_flags & NDCN_F_NO_CONTROL;
    NDCN_CHECK_ARG(no_graph || (desc->A.rowptr && (desc->A.nnz == 0 || (desc->A.colidx && desc->A.val))), "operator missing");
    NDCN_CHECK_ARG(no_ctl || desc->W, "weight missing");
    ndcn_solver *s = new (std::nothrow) ndcn_solver();
    if (!s) { set_error("out of host memory"); return NDCN_EINVAL; }
    s->d = *desc;
    s->n_rows = desc->A.n_rows;
    s->n_elem = s->n_rows * (int64_t)desc->H;
    s->n_mean = (double)s->n_elem;
    if (desc->shard) {
        NDCN_CHECK_ARG(desc->shard->comm && desc->shard->halo && desc->shard->n_global_rows >= desc->A.n_rows &&
                       desc->shard->n_blocks >= 0 && desc->shard->n_blocks <= 4, "bad shard descriptor");
        s->sharded = true;
        s->shard = *desc->shard;
        s->d.shard = &s->shard;
        s->n_halo = halo_plan_n_halo(s->shard.halo);
        s->n_send = halo_plan_n_send(s->shard.halo);
        s->n_own = desc->A.n_cols - s->n_halo;
        s->n_mean = (double)s->shard.n_global_rows * (double)desc->H;
        if (s->n_own != s->n_rows) { set_error("shard: operator has %lld rows, %lld own columns", (long long)s->n_rows, (long long)s->n_own); delete s; return NDCN_EINVAL; }
        // Row-split shards (lattices): which own rows of a stage input do the boundary launches and the exchange read?  One-off,
        // from the boundary blocks' column lists and the send list (a few thousand entries).  With an interior block on the
        // XADD kernel the first stage input of a dopri5 step is then formed on those rows only (enqueue_attempt).
        if (s->shard.n_blocks > 0 && desc->method == NDCN_M_DOPRI5 && !desc->use_graph) {
            std::vector<std::pair<int64_t, int64_t>> rs;
            auto add_range = [&](const int32_t *d_idx, int64_t n_idx) -> int {
                if (n_idx <= 0 || !d_idx) return NDCN_OK;
                std::vector<int32_t> h((size_t)n_idx);
                NDCN_HIP(hipMemcpy(h.data(), d_idx, (size_t)n_idx * sizeof(int32_t), hipMemcpyDeviceToHost));
                // the own rows an index list touches, as bands: sorted, cut wherever two neighbours lie more than a few lattice
                // rows apart.  (An interior rank's send list holds rows at BOTH ends of the shard - for the upper and the
                // lower neighbour: one [min, max] range would cover the whole shard and turn the thin combine into a full pass.)
                std::vector<int32_t> own;
                own.reserve(h.size());
                for (int32_t c : h)
                    if (c >= 0 && c < s->n_own) own.push_back(c);
                std::sort(own.begin(), own.end());
                const int64_t gap = 8192;
                for (size_t i = 0; i < own.size();) {
                    size_t j = i;
                    while (j + 1 < own.size() && (int64_t)own[j + 1] - own[j] <= gap) ++j;
                    rs.emplace_back((int64_t)own[i], (int64_t)own[j] + 1);
                    i = j + 1;
                }
                return NDCN_OK;
            };
            int rcx = add_range(halo_plan_send_idx(s->shard.halo), s->n_send);
            for (int b = 0; b < s->shard.n_blocks && !rcx; ++b) {
                const auto &blk = s->shard.blocks[b];
                if (blk.needs_halo) rcx = add_range(blk.A.colidx, blk.A.nnz);
                else if (s->xadd_block < 0 && rhs_xadd_supported(&blk.A, desc->H, desc->rhs_flags, 1, 1)) s->xadd_block = b;
            }
            if (rcx) { delete s; return rcx; }
            std::sort(rs.begin(), rs.end());
            for (const auto &r : rs) {
                if (!s->xadd_ranges.empty() && r.first <= s->xadd_ranges.back().second)
                    s->xadd_ranges.back().second = std::max(s->xadd_ranges.back().second, r.second);
                else s->xadd_ranges.push_back(r);
            }
            // every interior block must be able to form the input itself
            for (int b = 0; b < s->shard.n_blocks; ++b)
                if (!s->shard.blocks[b].needs_halo && !rhs_xadd_supported(&s->shard.blocks[b].A, desc->H, desc->rhs_flags, 1, 1)) s->xadd_block = -1;
            // bands that add up to most of the shard: the combine launch over the whole panel is cheaper than the gather
            // inside the interior launch plus a near-full band pass
            int64_t covered = 0;
            for (const auto &r : s->xadd_ranges) covered += r.second - r.first;
            if (2 * covered > s->n_own) { s->xadd_block = -1; s->xadd_ranges.clear(); }
        }
    }
    if (workspace) {
        s->slab = workspace;
        s->slab_bytes = (size_t)ws_bytes;
    } else {
        s->slab_bytes = workspace_bytes(desc);
        if (hipMalloc(&s->slab, s->slab_bytes) != hipSuccess) {
            set_error("hipMalloc of %zu workspace bytes failed", s->slab_bytes);
            delete s;
            return NDCN_EHIP;
        }
        s->slab_owned = true;
    }
    int rc = NDCN_OK;
    auto fail = [&](int code) { solver_destroy(s); return code; };
    if ((rc = alloc_panel(s, &s->ycur))) return fail(rc);
    s->ycur_own = s->ycur;
    if ((rc = alloc_panel(s, &s->ytmp))) return fail(rc);
    if ((rc = alloc_panel(s, &s->ytmp2))) return fail(rc);
    {
        const int64_t wb = rhs_work_bytes(s->n_rows, desc->H, desc->rhs_flags);
        if (wb > 0) {
            void *wq = nullptr;
            if ((rc = carve(s, (size_t)wb + 16, &wq))) return fail(rc);
            s->work = static_cast<float *>(wq);
        }
        const bool both = !(desc->rhs_flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL));
        if (s->sharded) {
            // the stage algebra rides in the epilogues of whatever kernel ndcn_rhs_rk_f32 selects per launch
            s->fused = both && rhs_fused_supported(desc->H, desc->rhs_flags);
            s->fused2 = true;
            void *q2 = nullptr;
            if (s->shard.X_halo) {
                s->halo = s->shard.X_halo;
            } else {
                if ((rc = carve(s, (size_t)s->n_halo * desc->H * sizeof(float) + 16, &q2))) return fail(rc);
                s->halo = static_cast<float *>(q2);
            }
            if ((rc = carve(s, (size_t)s->n_send * desc->H * sizeof(float) + 16, &q2))) return fail(rc);
            s->pack = static_cast<float *>(q2);
            if (s->shard.A_own.n_rows > 0) {
                if ((rc = carve(s, (size_t)s->n_elem * sizeof(float) + 16, &q2))) return fail(rc);
                s->sbuf = static_cast<float *>(q2);
            }
            if (hipStreamCreateWithFlags(&s->cstream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_x, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming) != hipSuccess) {
                set_error("creating the exchange stream failed");
                return fail(NDCN_EHIP);
            }
        } else {
        s->fused = both && rhs_fused_supported(desc->H, desc->rhs_flags);
        s->fused2 = s->fused && rhs_fused2_supported(&desc->A, desc->H, desc->rhs_flags);
        if (!no_graph && no_ctl && spmm_rec_supported(&desc->A, desc->H) && s->n_rows * (int64_t)1024 < (1ll << 32)) {
            s->fused2 = true;                                   // same stepping, different launch (rhs_epi)
            s->rec_epi = true;
        } else if (!no_graph && no_ctl && spmm_wide_rk_supported(&desc->A, desc->H)) {
            s->fused2 = s->rec_epi = s->wide_epi = true;
        } else if (both && !s->fused && desc->method == NDCN_M_DOPRI5 && rhs_small_supported(&desc->A, desc->H, desc->rhs_flags)) {
            // narrow panels: the adaptive step runs 1 combine + 6 launches; fixed-grid methods keep their replayed step,
            // whose right-hand sides go through the same kernel in plain mode (rhs_f32)
            s->fused2 = s->rec_epi = s->small_epi = true;
        }
        }
    }
    const int nk = desc->method == NDCN_M_DOPRI5 ? 7 : desc->method == NDCN_M_RK4 ? 4 : 1;
    for (int j = 0; j < nk; ++j)
        if ((rc = alloc_panel(s, &s->k[j]))) return fail(rc);
    if (desc->method == NDCN_M_DOPRI5) {
        if ((rc = alloc_panel(s, &s->ynext))) return fail(rc);
        if ((rc = alloc_panel(s, &s->yold))) return fail(rc);
        s->own3[0] = s->ycur_own; s->own3[1] = s->ynext; s->own3[2] = s->yold;
        if ((rc = alloc_panel(s, &s->ca))) return fail(rc);
        if ((rc = alloc_panel(s, &s->cb))) return fail(rc);
        if ((rc = alloc_panel(s, &s->cc))) return fail(rc);
        if ((rc = alloc_panel(s, &s->cd))) return fail(rc);
    }
    void *q = nullptr;
    if ((rc = carve(s, 4 * sizeof(double), &q))) return fail(rc);     // {sum, non-finite} (x 2 for the initial step's norm pair)
    s->d_red = static_cast<double *>(q);
    if ((rc = carve(s, (size_t)reduce_ws_bytes(), &q))) return fail(rc);
    s->d_ws = q;
    if ((rc = carve(s, (size_t)reduce_ws_bytes(), &q))) return fail(rc);      // partials of any RHS epilogue
    s->d_ws2 = q;
    if (hipHostMalloc(reinterpret_cast<void **>(&s->h_red), 8 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc failed");
        return fail(NDCN_EHIP);
    }
    if (!s->sharded && poll_records_enabled()) {      // (a shard's record passes through an all-reduce on the device first)
        void *alias = nullptr;
        if (hipHostGetDevicePointer(&alias, s->h_red, 0) == hipSuccess && alias) {
            s->d_red = static_cast<double *>(alias);
            s->poll = true;
        }
    }
    if (hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return fail(NDCN_EHIP); }
    if ((rc = carve(s, 256 + 2 * kCoefCap * sizeof(float), &q))) return fail(rc);
    s->d_dt = static_cast<float *>(q);
    s->d_coef = s->d_dt + 64;
    s->d_beta = s->d_coef + kCoefCap;
    // hipGraph replay pays off where the step is launch-bound.  The fused MFMA kernel takes dt by value (large
    // panels: never launch-bound); every other path reads it from device memory when replayed.
    s->graph_on = !s->sharded && desc->use_graph && !(s->fused2 && !s->rec_epi) &&
                  !(s->rec_epi && desc->method != NDCN_M_DOPRI5);
    if (s->graph_on) {
        if (hipStreamCreateWithFlags(&s->gstream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s->gev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->gev_out, hipEventDisableTiming) != hipSuccess) {
            set_error("creating the replay stream failed");
            return fail(NDCN_EHIP);
        }
    }
    *out = s;
    return NDCN_OK;
}

int solver_destroy(ndcn_solver *s) {
    if (!s) return NDCN_OK;
    if (s->slab_owned && s->slab) (void)hipFree(s->slab);
    if (s->h_red) (void)hipHostFree(s->h_red);
    if (s->ev) (void)hipEventDestroy(s->ev);
    if (s->gexec) (void)hipGraphExecDestroy(s->gexec);
    if (s->gev_in) (void)hipEventDestroy(s->gev_in);
    if (s->gev_out) (void)hipEventDestroy(s->gev_out);
    if (s->gstream) (void)hipStreamDestroy(s->gstream);
    if (s->cstream) (void)hipStreamDestroy(s->cstream);
    if (s->ev_x) (void)hipEventDestroy(s->ev_x);
    if (s->ev_halo) (void)hipEventDestroy(s->ev_halo);
    if (s->h_dt) (void)hipHostFree(s->h_dt);
    delete s;
    return NDCN_OK;
}

int solver_begin(ndcn_solver *s, const float *y0, double t0, hipStream_t st, bool borrow) {
    NDCN_CHECK_ARG(s && y0, "null argument");
    s->ycur = s->ycur_own;
    s->cur_is_borrowed = false;
    s->borrowed = nullptr;
    s->spare = nullptr;
    if (s->d.method == NDCN_M_DOPRI5) {               // the names go back to the panels they were allocated as
        s->ynext = s->own3[1];
        s->yold = s->own3[2];
    }
    if (borrow && s->d.method == NDCN_M_DOPRI5 && !s->d.use_graph) {
        // the first step reads y0 where the caller keeps it (one panel copy less per solve); when the panel would come
        // up for writing - two accepted steps later, as y1 - the solver's own spare takes its place (rotate_after_accept)
        s->ycur = const_cast<float *>(y0);
        s->borrowed = y0;
        s->spare = s->own3[0];
    } else {
        NDCN_HIP(hipMemcpyAsync(s->ycur, y0, (size_t)s->n_elem * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    s->n_attempt = s->n_accept = s->n_rhs = 0;
    s->log.clear();
    s->fit_pending = s->fit_valid = false;
    s->pending_bad = 0;
    s->last_ratio = 0;
    s->t0 = s->t1 = t0;
    s->tf = (float)t0;
    if (s->fused) {
        int rcp = pack_weight_256(s->d.W, s->work, st);
        if (rcp) return rcp;
        s->packed = true;
        s->exact32 = weights_wide_range(s->work);
    }
    if (s->d.method == NDCN_M_DOPRI5) {
        // dopri5.py:77-83
        int rc = rhs(s, s->ycur, s->k[0], st);
        if (rc) return rc;
        double h;
        rc = initial_step(s, st, h);
        if (rc) return rc;
        s->dt = h;
    }
    s->begun = true;
    return NDCN_OK;
}

// the kernel sequence of one fixed-grid step on the solver's own buffers (y updated in place), dt from d_dt
static int enqueue_fixed_step(ndcn_solver *s, hipStream_t st) {
    const int64_t n = s->n_elem;
    float *y = s->ycur_own;
    const float *dtp = s->d_dt;
    int rc;
    if ((rc = rhs(s, y, s->k[0], st))) return rc;
    switch (s->d.method) {
        case NDCN_M_EULER:
            return fixed_stage_f32(0, y, y, s->k[0], nullptr, nullptr, nullptr, 0.f, n, st, dtp);
        case NDCN_M_MIDPOINT:
            if ((rc = fixed_stage_f32(1, s->ytmp, y, s->k[0], nullptr, nullptr, nullptr, 0.f, n, st, dtp))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[0], st))) return rc;
            return fixed_stage_f32(0, y, y, s->k[0], nullptr, nullptr, nullptr, 0.f, n, st, dtp);
        default:
            if ((rc = fixed_stage_f32(2, s->ytmp, y, s->k[0], nullptr, nullptr, nullptr, 0.f, n, st, dtp))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[1], st))) return rc;
            if ((rc = fixed_stage_f32(3, s->ytmp, y, s->k[0], s->k[1], nullptr, nullptr, 0.f, n, st, dtp))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[2], st))) return rc;
            if ((rc = fixed_stage_f32(4, s->ytmp, y, s->k[0], s->k[1], s->k[2], nullptr, 0.f, n, st, dtp))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[3], st))) return rc;
            return fixed_stage_f32(5, y, y, s->k[0], s->k[1], s->k[2], s->k[3], 0.f, n, st, dtp);
    }
}

static int graph_setup(ndcn_solver *s) {
    NDCN_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_dt), kDtRing * sizeof(float), hipHostMallocDefault));
    hipGraph_t graph = nullptr;
    prof_pause(true);
    hipError_t eb = hipStreamBeginCapture(s->gstream, hipStreamCaptureModeThreadLocal);
    if (eb != hipSuccess) { prof_pause(false); set_error("hipStreamBeginCapture: %s", hipGetErrorString(eb)); return NDCN_EHIP; }
    const int64_t rhs_before = s->n_rhs;
    const int rc = enqueue_fixed_step(s, s->gstream);
    s->n_rhs = rhs_before;                                      // capturing is not evaluating
    hipError_t e = hipStreamEndCapture(s->gstream, &graph);
    prof_pause(false);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return NDCN_EHIP; }
    e = hipGraphInstantiate(&s->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return NDCN_EHIP; }
    return NDCN_OK;
}

}  // namespace ndcn
namespace {
int graph_setup_dopri5(ndcn_solver *s) {
    NDCN_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_dt), kDtRing * sizeof(float), hipHostMallocDefault));
    hipGraph_t graph = nullptr;
    prof_pause(true);
    hipError_t e = hipStreamBeginCapture(s->gstream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { prof_pause(false); set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return NDCN_EHIP; }
    const int64_t rhs_before = s->n_rhs;
    s->n_coef = 0;
    int rc = scale_coef_f32(s->d_coef, s->d_beta, s->d_dt, kCoefCap, s->gstream);    // first node: fl(dt * beta) table
    if (!rc) rc = enqueue_attempt(s, s->gstream, 1.f, s->d_dt);
    s->n_rhs = rhs_before;                                      // capturing is not evaluating
    e = hipStreamEndCapture(s->gstream, &graph);
    prof_pause(false);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return NDCN_EHIP; }
    // the tableau entries the captured launches refer to (constant for the life of the graph)
    NDCN_HIP(hipMemcpy(s->d_beta, s->h_beta, sizeof(s->h_beta), hipMemcpyHostToDevice));
    e = hipGraphInstantiate(&s->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return NDCN_EHIP; }
    return NDCN_OK;
}
}  // namespace
namespace ndcn {

static int graph_advance(ndcn_solver *s, double next_t, float *out, hipStream_t st) {
    const float t1 = (float)next_t;
    const float dt = t1 - s->tf;
    if (!s->gexec) {
        int rc = graph_setup(s);
        if (rc) return rc;
    }
    if (s->ycur != s->ycur_own) { set_error("graph mode: state must live in the solver"); return NDCN_ESTATE; }
    float *slot = s->h_dt + (s->g_step++ % kDtRing);
    *slot = dt;
    NDCN_HIP(hipEventRecord(s->gev_in, st));                    // order after the caller's stream (begin / previous copies)
    NDCN_HIP(hipStreamWaitEvent(s->gstream, s->gev_in, 0));
    NDCN_HIP(hipMemcpyAsync(s->d_dt, slot, sizeof(float), hipMemcpyHostToDevice, s->gstream));
    NDCN_HIP(hipGraphLaunch(s->gexec, s->gstream));
    if (out) NDCN_HIP(hipMemcpyAsync(out, s->ycur_own, (size_t)s->n_elem * sizeof(float), hipMemcpyDeviceToDevice, s->gstream));
    NDCN_HIP(hipEventRecord(s->gev_out, s->gstream));
    NDCN_HIP(hipStreamWaitEvent(st, s->gev_out, 0));            // the caller's stream sees the result
    const int per = s->d.method == NDCN_M_EULER ? 1 : s->d.method == NDCN_M_MIDPOINT ? 2 : 4;
    s->n_rhs += per;
    s->tf = t1;
    s->t0 = s->t1;
    s->t1 = next_t;
    s->n_attempt++;
    s->n_accept++;
    return NDCN_OK;
}

static int fixed_advance(ndcn_solver *s, double next_t, float *out, hipStream_t st) {
    if (s->graph_on) return graph_advance(s, next_t, out, st);
    // solvers.py:81-97 with grid == t: one step of size t1 - t0 formed in the state dtype
    const float t1 = (float)next_t;
    const float dt = t1 - s->tf;
    const int64_t n = s->n_elem;
    float *dst = out ? out : s->ycur_own;
    int rc;
    if (s->fused2 && s->d.method == NDCN_M_EULER && dst != s->ycur) {
        // y + dt * f in the RHS epilogue (one term: identical rounding to fixed_stage op 0)
        const float c1[1] = {dt};
        RkOpt opt = {};
        opt.no_k = 1;                                        // nothing reads an Euler step's K
        rc = rhs_epi(s, s->ycur, s->k[0], 1, s->ycur, nullptr, c1, 0, dst, 0.f, 0.f, nullptr, nullptr, st, nullptr, &opt);
        if (rc) return rc;
        s->ycur = dst;
        s->cur_is_borrowed = (dst != s->ycur_own);
        s->tf = t1;
        s->t0 = s->t1;
        s->t1 = next_t;
        s->n_attempt++;
        s->n_accept++;
        return NDCN_OK;
    }
    if (s->fused2 && s->d.method == NDCN_M_RK4) {
        // the 3/8-rule stage algebra in the RHS epilogues (rk_common.py:72-78): 4 launches instead of 8, each stage
        // input written by the launch that produced the stage it needs last
        const float c1[1] = {dt};
        const float *kp[3] = {s->k[0], s->k[1], s->k[2]};
        const float *in = s->ycur;
        for (int i = 0; i < 4; ++i) {
            float *nxt = (i == 3) ? dst : (in == s->ytmp ? s->ytmp2 : s->ytmp);
            RkOpt opt = {};
            opt.no_k = i == 3;                               // the fourth stage's derivative is consumed by its own epilogue
            rc = rhs_epi(s, in, s->k[i], 3, s->ycur, kp, c1, i, nxt, 0.f, 0.f, nullptr, nullptr, st, nullptr, &opt);
            if (rc) return rc;
            in = nxt;
        }
        s->ycur = dst;
        s->cur_is_borrowed = (dst != s->ycur_own);
        s->tf = t1;
        s->t0 = s->t1;
        s->t1 = next_t;
        s->n_attempt++;
        s->n_accept++;
        return NDCN_OK;
    }
    if ((rc = rhs(s, s->ycur, s->k[0], st))) return rc;
    switch (s->d.method) {
        case NDCN_M_EULER:
            rc = fixed_stage_f32(0, dst, s->ycur, s->k[0], nullptr, nullptr, nullptr, dt, n, st);
            break;
        case NDCN_M_MIDPOINT:
            if ((rc = fixed_stage_f32(1, s->ytmp, s->ycur, s->k[0], nullptr, nullptr, nullptr, dt, n, st))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[0], st))) return rc;
            rc = fixed_stage_f32(0, dst, s->ycur, s->k[0], nullptr, nullptr, nullptr, dt, n, st);
            break;
        default:  // rk4, 3/8 rule
            if ((rc = fixed_stage_f32(2, s->ytmp, s->ycur, s->k[0], nullptr, nullptr, nullptr, dt, n, st))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[1], st))) return rc;
            if ((rc = fixed_stage_f32(3, s->ytmp, s->ycur, s->k[0], s->k[1], nullptr, nullptr, dt, n, st))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[2], st))) return rc;
            if ((rc = fixed_stage_f32(4, s->ytmp, s->ycur, s->k[0], s->k[1], s->k[2], nullptr, dt, n, st))) return rc;
            if ((rc = rhs(s, s->ytmp, s->k[3], st))) return rc;
            rc = fixed_stage_f32(5, dst, s->ycur, s->k[0], s->k[1], s->k[2], s->k[3], dt, n, st);
            break;
    }
    if (rc) return rc;
    s->ycur = dst;                 // the next step reads the state from where it was written
    s->cur_is_borrowed = (dst != s->ycur_own);
    s->tf = t1;
    s->t0 = s->t1;
    s->t1 = next_t;
    s->n_attempt++;
    s->n_accept++;
    return NDCN_OK;
}

static int dopri5_advance(ndcn_solver *s, double next_t, float *out, int64_t budget, hipStream_t st);

// ndcn_solver_begin_borrowed: results must not land in the initial state the solver still reads
static bool overlaps_borrowed(const ndcn_solver *s, const float *out, int64_t n_panels) {
    if (!s->borrowed || !out || n_panels <= 0) return false;
    const float *lo = s->borrowed, *hi = s->borrowed + s->n_elem;
    if (out < hi && out + n_panels * s->n_elem > lo) {
        set_error("the output overlaps the initial state handed to ndcn_solver_begin_borrowed");
        return true;
    }
    return false;
}

int solver_advance(ndcn_solver *s, double next_t, float *out, int64_t budget, hipStream_t st) {
    NDCN_CHECK_ARG(s, "null solver");
    if (!s->begun) { set_error("ndcn_solver_advance before ndcn_solver_begin"); return NDCN_ESTATE; }
    if (overlaps_borrowed(s, out, 1)) return NDCN_EINVAL;
    if (s->d.method != NDCN_M_DOPRI5) return fixed_advance(s, next_t, out, st);
    // replay mode: the captured attempt is launched into the caller's stream like any other work (capture needed a
    // stream of its own, replay does not)
    return dopri5_advance(s, next_t, out, budget, st);
}

static int dopri5_advance(ndcn_solver *s, double next_t, float *out, int64_t budget, hipStream_t st) {
    int64_t done = 0;
    int64_t n_here = 0;
    while (next_t > s->t1) {                                   // dopri5.py:88
        if (budget > 0 && done >= budget) return 1;
        if (s->d.max_num_steps > 0 && n_here >= s->d.max_num_steps) {
            set_error("max_num_steps exceeded (%lld>=%lld)", (long long)n_here, (long long)s->d.max_num_steps);
            return NDCN_EMAXSTEPS;
        }
        if (s->fit_pending) {          // previous accepted step is being left without ever being evaluated
            int rcr = rotate_after_accept(s, st);
            if (rcr) return rcr;
            s->fit_pending = false;
        }
        int rc = dopri5_step(s, st);
        if (rc) return rc;
        ++done;
        ++n_here;
    }
    if (!out) return NDCN_OK;
    // interp.py:51-65: abscissa and its powers in the state dtype
    const float a0 = (float)s->t0, a1 = (float)s->t1, at = (float)next_t;
    if (!(a0 <= at && at <= a1)) {
        set_error("invalid interpolation, fails `t0 <= t <= t1`: %g, %g, %g", a0, at, a1);
        return NDCN_ESTATE;
    }
    const float x = (at - a0) / (a1 - a0);
    float xp[5];
    xp[4] = 1.f; xp[3] = x; xp[2] = xp[3] * x; xp[1] = xp[2] * x; xp[0] = xp[1] * x;
    if (!s->fit_valid) {
        if (!s->fit_pending) { set_error("no accepted step covers t=%g", next_t); return NDCN_ESTATE; }
        if (s->evals_in_step++ == 0) {
            // first tick inside this step: fit + evaluate in one pass, coefficients not materialised (most steps
            // are sampled at most once); a second tick in the same step pays for the stored fit below
            float cm[7];
            for (int j = 0; j < 7; ++j) cm[j] = s->fit_dt * (float)kCMid[j];
            return interp_direct_f32(s->ycur, s->ynext, s->k, cm, s->fit_dt, xp, out, s->n_elem, st);
        }
        int rc = do_fit(s, st);
        if (rc) return rc;
        s->ce = s->ycur;               // y0 of the fitted step; becomes yold at the rotation
        s->fit_valid = true;
    }
    return interp_eval_f32(s->ca, s->cb, s->cc, s->cd, s->ce, xp, out, s->n_elem, st);
}

// dopri5: all of `h_ticks` (increasing) in one call; out[i] = y(h_ticks[i]), panels back to back.  Ticks that fall
// into the same accepted step are evaluated together: the step's panels are read once per <= 8 ticks
// (interp_direct_multi_f32) instead of once per tick - the reference's drivers sample 16-120 ticks over a handful of
// steps, where the dense output is most of the solve (dgnn.py:173-182: 15 ticks in 2 steps).  Same arithmetic per
// element as the single-tick path (bit-identical).  Fixed-grid methods: one step per tick, as ndcn_solver_advance.
int solver_advance_many(ndcn_solver *s, const double *h_ticks, int64_t n_ticks, float *out, hipStream_t st) {
    NDCN_CHECK_ARG(s && (n_ticks == 0 || (h_ticks && out)), "null argument");
    if (!s->begun) { set_error("ndcn_solver_advance_many before ndcn_solver_begin"); return NDCN_ESTATE; }
    if (overlaps_borrowed(s, out, n_ticks)) return NDCN_EINVAL;
    const size_t stride = (size_t)s->n_elem;
    int64_t i = 0;
    // Fixed grid on a state that fits one compute unit: the whole time vector in ONE launch (solve_small.hip) - bit-identical
    // to the per-step kernels below, without their launch latency.
    if (s->d.method != NDCN_M_DOPRI5 && n_ticks > 0 && !s->sharded && solve_small_supported(&s->d.A, s->d.H, s->d.rhs_flags, s->d.method)) {
        std::vector<float> dts((size_t)n_ticks);
        float tf = s->tf;
        for (int64_t q = 0; q < n_ticks; ++q) {            // solvers.py:81-97 with grid == t: step sizes formed in the state dtype
            const float t1 = (float)h_ticks[q];
            dts[(size_t)q] = t1 - tf;
            tf = t1;
        }
        int rc = solve_small_f32(&s->d.A, s->d.W, s->d.b, s->d.H, s->d.rhs_flags, s->d.method, s->ycur, dts.data(), n_ticks, out, st);
        if (rc) return rc;
        float *last = out + (size_t)(n_ticks - 1) * stride;
        if (s->graph_on) {                                 // replayed steps keep the state inside the solver
            NDCN_HIP(hipMemcpyAsync(s->ycur_own, last, stride * sizeof(float), hipMemcpyDeviceToDevice, st));
            s->ycur = s->ycur_own;
            s->cur_is_borrowed = false;
        } else {
            s->ycur = last;
            s->cur_is_borrowed = true;
        }
        const int per = s->d.method == NDCN_M_EULER ? 1 : s->d.method == NDCN_M_MIDPOINT ? 2 : 4;
        s->n_rhs += per * n_ticks;
        s->tf = tf;
        s->t0 = n_ticks > 1 ? h_ticks[n_ticks - 2] : s->t1;
        s->t1 = h_ticks[n_ticks - 1];
        s->n_attempt += n_ticks;
        s->n_accept += n_ticks;
        return NDCN_OK;
    }
    while (i < n_ticks) {
        if (s->d.method != NDCN_M_DOPRI5) {
            int rc = fixed_advance(s, h_ticks[i], out + i * stride, st);
            if (rc) return rc;
            ++i;
            continue;
        }
        int rc = dopri5_advance(s, h_ticks[i], nullptr, 0, st);          // steps only (no evaluation)
        if (rc) return rc;
        if (s->fit_valid || !s->fit_pending) {                           // a stored fit / no fresh step: single-tick path
            rc = dopri5_advance(s, h_ticks[i], out + i * stride, 0, st);
            if (rc) return rc;
            ++i;
            continue;
        }
        int64_t j = i;
        while (j < n_ticks && !(h_ticks[j] > s->t1)) ++j;                // the ticks this accepted step covers
        const float a0 = (float)s->t0, a1 = (float)s->t1;
        float cm[7];
        for (int q = 0; q < 7; ++q) cm[q] = s->fit_dt * (float)kCMid[q];
        while (i < j) {
            const int nt = (int)((j - i) < 8 ? (j - i) : 8);
            float xp[8][5];
            float *outs[8];
            for (int t = 0; t < nt; ++t) {
                // interp.py:51-65: abscissa and its powers in the state dtype
                const float at = (float)h_ticks[i + t];
                if (!(a0 <= at && at <= a1)) {
                    set_error("invalid interpolation, fails `t0 <= t <= t1`: %g, %g, %g", a0, at, a1);
                    return NDCN_ESTATE;
                }
                const float x = (at - a0) / (a1 - a0);
                xp[t][4] = 1.f; xp[t][3] = x; xp[t][2] = xp[t][3] * x; xp[t][1] = xp[t][2] * x; xp[t][0] = xp[t][1] * x;
                outs[t] = out + (i + t) * stride;
            }
            rc = interp_direct_multi_f32(s->ycur, s->ynext, s->k, cm, s->fit_dt, &xp[0][0], outs, nt, s->n_elem, st);
            if (rc) return rc;
            s->evals_in_step += nt;
            i += nt;
        }
    }
    return NDCN_OK;
}

int solver_stats(const ndcn_solver *s, double h[6]) {
    NDCN_CHECK_ARG(s && h, "null argument");
    h[0] = (double)s->n_attempt; h[1] = (double)s->n_accept; h[2] = (double)s->n_rhs;
    h[3] = s->t1; h[4] = s->dt; h[5] = s->last_ratio;
    return NDCN_OK;
}

int64_t solver_steplog(const ndcn_solver *s, double *rows, int64_t cap) {
    if (!s) return 0;
    const int64_t n = (int64_t)s->log.size() / 5;
    if (rows) {
        const int64_t m = n < cap ? n : cap;
        for (int64_t i = 0; i < m * 5; ++i) rows[i] = s->log[i];
    }
    return n;
}

}  // namespace ndcn
