// extern "C" boundary of libndcn_hip.so: argument validation + forwarding.  See include/ndcn_hip.h.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace ndcn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int stream_grid_full(int64_t n_items, int block) {
    static const bool full = [] { const char *e = getenv("NDCN_STREAM_FULL"); return !(e && e[0] == '0'); }();
    if (!full) return stream_grid(n_items, block);
    int64_t g = (n_items + block - 1) / block;
    if (g > (1ll << 24)) g = 1ll << 24;             // (grid-stride loops take the rest)
    return g < 1 ? 1 : (int)g;
}

static int check_csr(const ndcn_csr *A, const char *who) {
    if (!A) { set_error("%s: null operator", who); return NDCN_EINVAL; }
    if (A->n_rows < 0 || A->n_cols < 0 || A->nnz < 0 || A->n_rows >= (1ll << 31) - 1 || A->nnz >= (1ll << 31) - 1) {
        set_error("%s: operator dimensions out of range", who);
        return NDCN_EINVAL;
    }
    if (!A->rowptr || (A->nnz > 0 && (!A->colidx || !A->val))) { set_error("%s: null CSR array", who); return NDCN_EINVAL; }
    return NDCN_OK;
}

}  // namespace ndcn

using namespace ndcn;
#define ST(s) static_cast<hipStream_t>(s)

extern "C" {

int ndcn_abi_version(void) { return NDCN_ABI_VERSION; }
int ndcn_debug_last_rhs_path(void) { return g_last_rhs_path; }
int ndcn_set_range_guard(int on) { return set_range_guard(on); }
const char *ndcn_last_error(void) { return g_err; }

int64_t ndcn_adjoint_rhs_work_bytes(int64_t n_rows, int H, uint32_t flags) { return adjoint_rhs_work_bytes(n_rows, H, flags); }

int ndcn_adjoint_rhs_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *y, const float *a, const float *W, const float *b,
                         float *K, float *vjp_y, float *vjp_W, float *vjp_b, void *work, int H, uint32_t flags, void *stream) {
    NDCN_CHECK_ARG(A && y && a && K && vjp_y && work && H > 0, "null argument");
    const bool graph = !(flags & NDCN_F_NO_GRAPH), ctl = !(flags & NDCN_F_NO_CONTROL);
    if (graph) {
        int rc = check_csr(A, __func__);
        if (rc) return rc;
        if ((rc = check_csr(A_t, __func__))) return rc;
        NDCN_CHECK_ARG(A->n_rows == A->n_cols && A_t->n_rows == A->n_cols && A_t->n_cols == A->n_rows && A_t->nnz == A->nnz,
                       "the adjoint needs a square operator and its transpose");
    }
    NDCN_CHECK_ARG(!ctl || (W && vjp_W && vjp_b), "weight / parameter-gradient buffers missing");
    NDCN_CHECK_ARG(aligned16(y) && aligned16(a) && aligned16(K) && aligned16(vjp_y) && (reinterpret_cast<uintptr_t>(work) & 255) == 0,
                   "panels must be 16-byte aligned, the scratch 256-byte aligned");
    return adjoint_rhs_f32(A, A_t, y, a, W, b, K, vjp_y, vjp_W, vjp_b, work, H, flags, ST(stream));
}

int64_t ndcn_gcn_work_bytes(int64_t n_cols, int H_out) { return n_cols < 0 || H_out <= 0 ? 0 : (((n_cols * (int64_t)H_out * 4) + 255) & ~(int64_t)255); }

int ndcn_gcn_f32(const ndcn_csr *A, const float *X, const float *W, const float *b, float *Y, float *work, int H_in, int H_out,
                 uint32_t flags, void *stream) {
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    NDCN_CHECK_ARG(X && W && Y && work && H_in > 0 && H_out > 0, "null argument");
    NDCN_CHECK_ARG(aligned16(work) && aligned16(Y), "panels must be 16-byte aligned");
    // support = X W^T + b over the operator's COLUMN nodes (no activation), then the sparse product over it
    if ((rc = linear_f32(X, W, b, work, A->n_cols, H_in, H_out, 0, ST(stream)))) return rc;
    return spmm_f32(A, work, nullptr, A->n_cols, Y, H_out, 1.0f, flags & NDCN_F_RELU, ST(stream));
}

int ndcn_solve_small_supported(const ndcn_csr *A, int H, uint32_t flags, int method, int backward) {
    if (!A) return 0;
    return backward ? solve_small_bwd_supported(A, H, flags, method) : solve_small_supported(A, H, flags, method);
}

int ndcn_solve_small_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method, const float *y0,
                         const float *h_dt, int64_t n_ticks, float *out, void *stream) {
    NDCN_CHECK_ARG(A && y0 && (n_ticks == 0 || (h_dt && out)) && n_ticks >= 0, "null argument");
    NDCN_CHECK_ARG((flags & NDCN_F_NO_CONTROL) || W, "weight missing");
    if (!(flags & NDCN_F_NO_GRAPH)) { int rc = check_csr(A, __func__); if (rc) return rc; }
    return solve_small_f32(A, W, b, H, flags, method, y0, h_dt, n_ticks, out, ST(stream));
}

int ndcn_solve_small_keep_supported(const ndcn_csr *A, int H, uint32_t flags) { return A ? solve_small_keep_supported(A, H, flags) : 0; }

int ndcn_solve_small_keep_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, const float *y0, const float *h_dt,
                              int64_t n_ticks, float *out, float *keep, void *stream) {
    NDCN_CHECK_ARG(A && y0 && W && h_dt && out && keep && n_ticks >= 1, "null argument");
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    return solve_small_f32(A, W, b, H, flags, NDCN_M_EULER, y0, h_dt, n_ticks, out, ST(stream), keep);
}

int ndcn_solve_small_bwd_keep_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *W, const float *b, int H, uint32_t flags,
                                  const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, const float *keep, float *g_y0,
                                  float *g_W, float *g_b, void *stream) {
    NDCN_CHECK_ARG(A && traj && g_out && g_y0 && h_dt && keep && W && g_W && g_b && n_ticks >= 1, "null argument");
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    if ((rc = check_csr(A_t, __func__))) return rc;
    return solve_small_bwd_f32(A, A_t, W, b, H, flags, NDCN_M_EULER, traj, g_out, h_dt, n_ticks, g_y0, g_W, g_b, ST(stream), keep);
}

int ndcn_solve_small_bwd_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *W, const float *b, int H, uint32_t flags, int method,
                             const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, float *g_y0, float *g_W,
                             float *g_b, void *stream) {
    NDCN_CHECK_ARG(A && traj && g_out && g_y0 && h_dt && n_ticks >= 1, "null argument");
    NDCN_CHECK_ARG((flags & NDCN_F_NO_CONTROL) || (W && g_W && g_b), "weight / gradient buffers missing");
    if (!(flags & NDCN_F_NO_GRAPH)) {
        int rc = check_csr(A, __func__);
        if (rc) return rc;
        if ((rc = check_csr(A_t, __func__))) return rc;
    }
    return solve_small_bwd_f32(A, A_t, W, b, H, flags, method, traj, g_out, h_dt, n_ticks, g_y0, g_W, g_b, ST(stream));
}

int ndcn_device_info(int64_t h_out[6]) {
    NDCN_CHECK_ARG(h_out, "null output");
    int dev = 0;
    NDCN_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    NDCN_HIP(hipGetDeviceProperties(&p, dev));
    h_out[0] = p.multiProcessorCount;
    h_out[1] = kXcds;
    h_out[2] = p.warpSize;
    h_out[3] = p.clockRate;
    h_out[4] = (int64_t)(p.totalGlobalMem >> 20);
    h_out[5] = (int64_t)(p.l2CacheSize >> 10);
    return NDCN_OK;
}

int ndcn_spmm_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own, float *Y, int H, float alpha,
                  uint32_t flags, void *stream) {
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    NDCN_CHECK_ARG(H > 0, "H must be positive");
    NDCN_CHECK_ARG(A->n_rows == 0 || (X && Y), "null panel");
    NDCN_CHECK_ARG(X != Y, "Y must not alias X");
    NDCN_CHECK_ARG(X_halo || n_own >= A->n_cols, "columns beyond n_own need a halo panel");
    return spmm_f32(A, X, X_halo, n_own, Y, H, alpha, flags, ST(stream));
}

int ndcn_linear_f32(const float *S, const float *W, const float *b, float *Y, int64_t n, int H_in, int H_out,
                    uint32_t flags, void *stream) {
    NDCN_CHECK_ARG(n >= 0 && H_in > 0 && H_out > 0, "bad shape");
    NDCN_CHECK_ARG(n == 0 || (S && W && Y), "null pointer");
    NDCN_CHECK_ARG(S != Y, "Y must not alias S");
    return linear_f32(S, W, b, Y, n, H_in, H_out, flags, ST(stream));
}

int64_t ndcn_rhs_work_bytes(int64_t n_rows, int H, uint32_t flags) { return rhs_work_bytes(n_rows, H, flags); }

int64_t ndcn_linear_bwd_work_bytes(int64_t n, int H_in, int H_out) { return linear_bwd_work_bytes(n, H_in, H_out); }

int ndcn_linear_bwd_f32(const float *g, const float *Y, const float *S, const float *W, float *gS, float *gW, float *gb,
                        void *work, int64_t n, int H_in, int H_out, uint32_t flags, void *stream) {
    NDCN_CHECK_ARG(n >= 0 && H_in > 0 && H_out > 0, "bad shape");
    NDCN_CHECK_ARG(n == 0 || g, "null gradient");
    NDCN_CHECK_ARG(!gS || W, "gS needs the weight");
    NDCN_CHECK_ARG(gS != g, "gS must not alias g");
    return linear_bwd_f32(g, Y, S, W, gS, gW, gb, work, n, H_in, H_out, ST(stream), flags);
}

int64_t ndcn_rk_bwd_ws_bytes(void) { return rk_bwd_ws_bytes(); }

int ndcn_rk_dot_diff_f32(const float *g, const float *a, const float *b, double *d_dots, void *d_ws, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && g && a && d_dots && d_ws, "bad argument");
    return rk_dot_diff_f32(g, a, b, d_dots, d_ws, n_elem, ST(stream));
}

int ndcn_rk_combine_bwd_f32(const float *g, const float *const *h_k, const float *h_c, int n_k, float *const *h_gk,
                            const float *const *h_acc, float *gy0, const float *acc_y0, double *d_dots, void *d_ws, int64_t n_elem,
                            void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && g && h_k && h_c && d_dots && d_ws, "bad argument");
    return rk_combine_bwd_f32(g, h_k, h_c, n_k, h_gk, h_acc, gy0, acc_y0, d_dots, d_ws, n_elem, ST(stream));
}

int ndcn_rk_error_bwd_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol,
                          float atol, float g_r, double inv_n, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0,
                          const float *acc_y1, const float *const *h_acc, double *d_dots, void *d_ws, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && y0 && y1 && h_k && h_c && d_dots && d_ws, "bad argument");
    return rk_error_bwd_f32(y0, y1, h_k, h_c, n_k, rtol, atol, g_r, inv_n, gy0, gy1, h_gk, acc_y0, acc_y1, h_acc, d_dots, d_ws, n_elem,
                            ST(stream));
}

int ndcn_rk_rms_bwd_f32(const float *a, const float *b, const float *y, float rtol, float atol, float coef, float *ga, float *gb,
                        float *gy, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && a && y, "bad argument");
    return rk_rms_bwd_f32(a, b, y, rtol, atol, coef, ga, gb, gy, n_elem, ST(stream));
}

int ndcn_dopri5_interp_bwd_f32(const float *g, const float *y0, const float *y1, const float *const *h_k, float dt, float x,
                               float *gy0, float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1,
                               const float *const *h_acc, double *d_dots, void *d_ws, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && g && y0 && y1 && h_k && d_dots && d_ws, "bad argument");
    return rk_dense_bwd_f32(g, y0, y1, h_k, dt, x, gy0, gy1, h_gk, acc_y0, acc_y1, h_acc, d_dots, d_ws, n_elem, ST(stream));
}

int ndcn_dopri5_interp_direct_multi_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                                        const float *h_xpow, float *const *h_out, int n_t, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && y0 && y1 && h_k && h_cmid && h_xpow && h_out, "bad argument");
    return interp_direct_multi_f32(y0, y1, h_k, h_cmid, dt, h_xpow, h_out, n_t, n_elem, ST(stream));
}

int ndcn_dopri5_interp_bwd_multi_f32(const float *const *h_g, int n_t, const float *y0, const float *y1, const float *const *h_k, float dt,
                                     const float *h_x, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0,
                                     const float *acc_y1, const float *const *h_acc, double *d_dots, void *d_ws, int64_t n_elem,
                                     void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_g && y0 && y1 && h_k && h_x && d_dots && d_ws, "bad argument");
    return rk_dense_bwd_multi_f32(h_g, n_t, y0, y1, h_k, dt, h_x, gy0, gy1, h_gk, acc_y0, acc_y1, h_acc, d_dots, d_ws, n_elem, ST(stream));
}

int ndcn_relu_bwd_f32(float *out, const float *g, const float *y, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && (n_elem == 0 || (out && g && y)), "bad argument");
    return relu_bwd_f32(out, g, y, n_elem, ST(stream));
}

int ndcn_scale_f32(float *out, const float *x, float w, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && (n_elem == 0 || (out && x)), "bad argument");
    return scale_f32(out, x, w, n_elem, ST(stream));
}

int ndcn_rhs_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own, const float *W, const float *b,
                 float *Y, float *work, int H, uint32_t flags, void *stream) {
    NDCN_CHECK_ARG(A, "null operator descriptor (n_rows is read from it even under NO_GRAPH)");
    NDCN_CHECK_ARG(H > 0, "H must be positive");
    if (!(flags & NDCN_F_NO_GRAPH)) {
        int rc = check_csr(A, __func__);
        if (rc) return rc;
        NDCN_CHECK_ARG(X_halo || n_own >= A->n_cols, "columns beyond n_own need a halo panel");
    }
    NDCN_CHECK_ARG(A->n_rows == 0 || (X && Y), "null panel");
    NDCN_CHECK_ARG(X != Y, "Y must not alias X");
    NDCN_CHECK_ARG((flags & NDCN_F_NO_CONTROL) || W, "weight missing");
    return rhs_f32(A, X, X_halo, n_own, W, b, Y, work, H, flags, ST(stream));
}

int ndcn_copy_f32(float *dst, const float *src, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && (n_elem == 0 || (dst && src)), "bad argument");
    return copy_f32(dst, src, n_elem, ST(stream));
}

int ndcn_row_l1_normalize_bwd_f32(const float *G, const float *X, float *GX, int64_t n_rows, int H, void *stream) {
    NDCN_CHECK_ARG(n_rows >= 0 && H >= 0, "negative size");
    NDCN_CHECK_ARG(n_rows == 0 || H == 0 || (G && X && GX), "null panel");
    return row_l1_normalize_bwd_f32(G, X, GX, n_rows, H, ST(stream));
}

static int rhs_rk_entry(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own, const float *W, const float *b,
                        float *K, float *work, int H, uint32_t flags, int rk_mode, const float *y0,
                        const float *const *h_kprev, const float *h_c, int n_prev, float *y_next, const float *y1, float *y_aux,
                        const float *h_c_aux, float rtol, float atol, double *d_out, void *d_ws, const float *x_add, float x_add_c,
                        void *stream, const float *x_mask = nullptr, float *s_out = nullptr) {
    NDCN_CHECK_ARG(A, "null operator descriptor");
    NDCN_CHECK_ARG(H > 0, "H must be positive");
    NDCN_CHECK_ARG(rk_mode >= 0 && rk_mode <= 3, "rk_mode must be 0, NDCN_RK_COMBINE, NDCN_RK_ERROR or NDCN_RK_RK4");
    if (!(flags & NDCN_F_NO_GRAPH)) {
        int rc = check_csr(A, __func__);
        if (rc) return rc;
        NDCN_CHECK_ARG(X_halo || n_own >= A->n_cols, "columns beyond n_own need a halo panel");
    }
    NDCN_CHECK_ARG(A->n_rows == 0 || (X && K), "null panel");
    NDCN_CHECK_ARG(X != K, "K must not alias X");
    NDCN_CHECK_ARG((flags & NDCN_F_NO_CONTROL) || W, "weight missing");
    if (rk_mode != 0) {
        NDCN_CHECK_ARG(y0 && h_c && (n_prev == 0 || h_kprev), "rk arguments missing");
        NDCN_CHECK_ARG(rk_mode != NDCN_RK_COMBINE || (y_next && y_next != X && y_next != K), "y_next missing or aliased");
        NDCN_CHECK_ARG(rk_mode != NDCN_RK_ERROR || (d_out && d_ws), "error record / scratch missing");
        NDCN_CHECK_ARG(rk_mode != NDCN_RK_RK4 || (y_next && y_next != X && y_next != K && n_prev >= 0 && n_prev <= 3),
                       "rk4 stage: y_next missing / aliased or stage index outside 0..3");
    }
    NDCN_CHECK_ARG(!y_aux || (rk_mode == NDCN_RK_COMBINE && h_c_aux && y_aux != y_next && y_aux != K && y_aux != X),
                   "y_aux: NDCN_RK_COMBINE only, with h_c_aux, not aliasing X / K / y_next");
    if (x_add) {
        NDCN_CHECK_ARG(!X_halo && x_add != K && x_add != y_next && rk_mode == NDCN_RK_COMBINE && rhs_xadd_supported(A, H, flags, rk_mode, n_prev),
                       "x_add: not supported for this operator / mode (ndcn_rhs_xadd_supported), or aliased");
    }
    if (x_mask || s_out) {
        NDCN_CHECK_ARG(!X_halo && !x_add && !(x_mask && s_out) && (!x_mask || (x_mask != K && x_mask != y_next)) &&
                           (!s_out || (s_out != K && s_out != y_next && s_out != X)) && rhs_adj_supported(A, H, flags, rk_mode, n_prev),
                       "x_mask / s_out: not supported for this operator / mode (ndcn_rhs_adj_supported), both given, or aliased");
    }
    const RkOpt opt = {y1, (flags & NDCN_F_ACCUM) ? 1 : 0, y_aux, h_c_aux, x_add, x_add_c, x_mask, s_out};
    return rhs_rk_f32(A, X, X_halo, n_own, W, b, K, work, H, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol,
                      d_out, d_ws, ST(stream), &opt);
}

int ndcn_rhs_rk_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own, const float *W, const float *b,
                    float *K, float *work, int H, uint32_t flags, int rk_mode, const float *y0,
                    const float *const *h_kprev, const float *h_c, int n_prev, float *y_next, const float *y1, float *y_aux,
                    const float *h_c_aux, float rtol, float atol, double *d_out, void *d_ws, void *stream) {
    return rhs_rk_entry(A, X, X_halo, n_own, W, b, K, work, H, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, y1, y_aux, h_c_aux,
                        rtol, atol, d_out, d_ws, nullptr, 0.f, stream);
}

int ndcn_rhs_adj_supported(const ndcn_csr *A, int H, uint32_t flags, int rk_mode, int n_prev) {
    return A ? rhs_adj_supported(A, H, flags, rk_mode, n_prev) : 0;
}

int ndcn_rhs_rk_adj_f32(const ndcn_csr *A, const float *X, const float *x_mask, float *s_out, const float *W, const float *b, float *K,
                        float *work, int H, uint32_t flags, int rk_mode, const float *y0, const float *const *h_kprev, const float *h_c,
                        int n_prev, float *y_next, const float *y1, float rtol, float atol, double *d_out, void *d_ws, void *stream) {
    NDCN_CHECK_ARG(x_mask || s_out, "one of x_mask / s_out expected (ndcn_rhs_rk_f32 otherwise)");
    return rhs_rk_entry(A, X, nullptr, A ? A->n_cols : 0, W, b, K, work, H, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, y1, nullptr,
                        nullptr, rtol, atol, d_out, d_ws, nullptr, 0.f, stream, x_mask, s_out);
}

int ndcn_rhs_xadd_supported(const ndcn_csr *A, int H, uint32_t flags, int rk_mode, int n_prev) {
    return (A && rk_mode == NDCN_RK_COMBINE) ? rhs_xadd_supported(A, H, flags, rk_mode, n_prev) : 0;
}

int ndcn_rhs_rk_xadd_f32(const ndcn_csr *A, const float *X, const float *x_add, float x_add_c, const float *W, const float *b,
                         float *K, float *work, int H, uint32_t flags, const float *y0, const float *k_prev, const float *h_c,
                         float *y_next, void *stream) {
    NDCN_CHECK_ARG(x_add && k_prev, "null panel");
    const float *kp[1] = {k_prev};
    return rhs_rk_entry(A, X, nullptr, A ? A->n_cols : 0, W, b, K, work, H, flags, NDCN_RK_COMBINE, y0, kp, h_c, 1, y_next, nullptr,
                        nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, x_add, x_add_c, stream);
}

int ndcn_gather_rows_f32(const float *X, const int32_t *idx, int64_t n_idx, int H, float *out, void *stream) {
    NDCN_CHECK_ARG(n_idx >= 0 && H > 0, "bad shape");
    NDCN_CHECK_ARG(n_idx == 0 || (X && idx && out), "null pointer");
    return gather_rows_f32(X, idx, n_idx, H, out, ST(stream));
}

int ndcn_rk_combine_f32(float *out, const float *y0, const float *const *h_k, const float *h_c, int n_k, int64_t n_elem,
                        void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_k && h_c, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || out, "null panel");       // y0 == NULL: the plain linear combination sum_j c_j k_j
    return rk_combine_f32(out, y0, h_k, h_c, n_k, n_elem, ST(stream));
}

int ndcn_rk_error_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol,
                      float atol, int64_t n_elem, double *d_out, void *d_ws, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_k && h_c && d_out && d_ws, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || (y0 && y1), "null panel");
    return rk_error_f32(y0, y1, h_k, h_c, n_k, rtol, atol, n_elem, d_out, d_ws, ST(stream));
}

int ndcn_scaled_sumsq_f32(const float *a, const float *b, const float *y, float rtol, float atol, int64_t n_elem,
                          double *d_out, void *d_ws, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && d_out && d_ws, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || (a && y), "null panel");
    return scaled_sumsq_f32(a, b, y, rtol, atol, n_elem, d_out, d_ws, ST(stream));
}

int64_t ndcn_reduce_ws_bytes(void) { return reduce_ws_bytes(); }

int64_t ndcn_set_aten_norm_max(int64_t n_elem) { return set_aten_order_max_elems(n_elem); }

int ndcn_dopri5_interp_fit_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                               float *a, float *b, float *c, float *d, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_k && h_cmid, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || (y0 && y1 && a && b && c && d), "null panel");
    return interp_fit_f32(y0, y1, h_k, h_cmid, dt, a, b, c, d, n_elem, ST(stream));
}

int ndcn_dopri5_interp_direct_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                                  const float h_xpow[5], float *out, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_k && h_cmid && h_xpow, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || (y0 && y1 && out), "null panel");
    return interp_direct_f32(y0, y1, h_k, h_cmid, dt, h_xpow, out, n_elem, ST(stream));
}

int ndcn_interp_eval_f32(const float *a, const float *b, const float *c, const float *d, const float *e,
                         const float h_xpow[5], float *out, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0 && h_xpow, "bad argument");
    NDCN_CHECK_ARG(n_elem == 0 || (a && b && c && d && e && out), "null panel");
    return interp_eval_f32(a, b, c, d, e, h_xpow, out, n_elem, ST(stream));
}

int ndcn_fixed_stage_f32(int op, float *out, const float *y, const float *k1, const float *k2, const float *k3,
                         const float *k4, float dt, int64_t n_elem, void *stream) {
    NDCN_CHECK_ARG(n_elem >= 0, "bad size");
    NDCN_CHECK_ARG(n_elem == 0 || (out && y), "null panel");
    return fixed_stage_f32(op, out, y, k1, k2, k3, k4, dt, n_elem, ST(stream));
}

int ndcn_row_l1_normalize_f32(const float *X, float *Y, int64_t n_rows, int H, void *stream) {
    NDCN_CHECK_ARG(n_rows >= 0 && H >= 0, "negative size");
    NDCN_CHECK_ARG(n_rows == 0 || H == 0 || (X && Y), "null panel");
    return row_l1_normalize_f32(X, Y, n_rows, H, ST(stream));
}

int ndcn_gene_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float f, float h, void *stream) {
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    NDCN_CHECK_ARG(A->n_rows == 0 || (x && out), "null vector");
    NDCN_CHECK_ARG(x != out, "out must not alias x");
    return gene_rhs_f32(A, x, out, b, f, h, ST(stream));
}

int ndcn_mutual_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float k, float c, float d, float e,
                        float h, void *stream) {
    int rc = check_csr(A, __func__);
    if (rc) return rc;
    NDCN_CHECK_ARG(A->n_rows == 0 || (x && out), "null vector");
    NDCN_CHECK_ARG(x != out, "out must not alias x");
    return mutual_rhs_f32(A, x, out, b, k, c, d, e, h, ST(stream));
}

int64_t ndcn_solver_workspace_bytes(const ndcn_solver_desc *desc) { return solver_workspace_bytes(desc); }
int ndcn_solver_create(const ndcn_solver_desc *desc, void *workspace, int64_t workspace_bytes, ndcn_solver **out) {
    return solver_create(desc, workspace, workspace_bytes, out);
}
int ndcn_solver_destroy(ndcn_solver *s) { return solver_destroy(s); }
int ndcn_solver_begin(ndcn_solver *s, const float *y0, double t0, void *stream) { return solver_begin(s, y0, t0, ST(stream)); }
int ndcn_solver_begin_borrowed(ndcn_solver *s, const float *y0, double t0, void *stream) { return solver_begin(s, y0, t0, ST(stream), true); }
int ndcn_solver_advance(ndcn_solver *s, double next_t, float *out, int64_t step_budget, void *stream) {
    return solver_advance(s, next_t, out, step_budget, ST(stream));
}
int ndcn_solver_advance_many(ndcn_solver *s, const double *h_ticks, int64_t n_ticks, float *out, void *stream) {
    return solver_advance_many(s, h_ticks, n_ticks, out, ST(stream));
}
int ndcn_solver_stats(const ndcn_solver *s, double h_stats[6]) { return solver_stats(s, h_stats); }
int64_t ndcn_solver_steplog(const ndcn_solver *s, double *h_rows, int64_t cap) { return solver_steplog(s, h_rows, cap); }

}  // extern "C"
