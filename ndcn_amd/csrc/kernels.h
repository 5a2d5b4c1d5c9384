// Internal (C++) entry points of the kernels; the extern "C" layer in api.hip validates arguments and
// forwards here.
#pragma once
#include "common.h"

namespace ndcn {

// Options of an RK-epilogue launch that only the sharded callers use (ndcn_rhs_rk_f32: y1 / NDCN_F_ACCUM):
//   y1     ERROR mode - the state whose error record is formed, by ROW of this launch (NULL: X, the evaluation's input).
//          A launch over a row block of a shard, or the second phase of a two-phase evaluation (whose X is the partial sum
//          A_own X, not y1), passes it explicitly.
//   accum  ERROR mode - add this launch's {sum, bad} to d_out instead of overwriting it (an evaluation split into
//          several launches on one stream; the order of the launches fixes the order of the sum)
//   y_aux  COMBINE mode - a SECOND linear combination of the same stages, without y0:
//            y_aux = sum_{m<n_prev} c_aux[m] kprev[m] + c_aux[n_prev] K          (left to right, products rounded on their own)
//          dopri5 uses it to form E = dt sum_{j<=6} c_err[j] k_j in the launch that produces k6 - where k1, k3, k4, k5
//          are in registers anyway - so that the error launch reads {y0, E, y1} instead of {y0, k1, k3, k4, k5, k6, y1}:
//          3 P less traffic per step for 1 P written, the same sum in the same order (E holds the partial sum exactly).
//   xadd / xadd_c (plain and COMBINE with one earlier stage, operators on the rhs_fused3 path without a halo panel -
//          rhs_xadd_supported()): the evaluation's input is X + xadd_c * xadd, formed on the rows the kernel stages instead of by a
//          combine launch in front (3 panels); same product-then-sum rounding per element.
//   no_k   (a hint; COMBINE / RK4 modes) the caller reads only y_next: a kernel that can skips the store of K - an Euler step's K, the
//          fourth stage of an RK4 step.  Honoured by rhs_fused3 (one panel of HBM writes less); K must still point at a panel.
struct RkOpt { const float *y1; int accum; float *y_aux; const float *c_aux; const float *xadd; float xadd_c;
               const float *xmask; float *s_out; int no_k; };      // (xmask, s_out: rhs_adj_supported)

int spmm_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, int H, float alpha,
             uint32_t flags, hipStream_t st);
int gather_rows_f32(const float *X, const int32_t *idx, int64_t n_idx, int H, float *out, hipStream_t st);
int linear_f32(const float *S, const float *W, const float *b, float *Y, int64_t n, int Hi, int Ho, uint32_t flags,
               hipStream_t st);
int rhs_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *Y,
            float *work, int H, uint32_t flags, hipStream_t st);
int64_t rhs_work_bytes(int64_t n_rows, int H, uint32_t flags);
int linear_bwd_f32(const float *g, const float *Y, const float *S, const float *W, float *gS, float *gW, float *gb, void *work,
                   int64_t n, int Hi, int Ho, hipStream_t st, uint32_t flags = 0, float acc_scale = 1.f, bool accumulate = false);
int64_t linear_bwd_work_bytes(int64_t n, int Hi, int Ho);
int scale_f32(float *out, const float *x, float w, int64_t n, hipStream_t st);
int relu_bwd_f32(float *out, const float *g, const float *y, int64_t n, hipStream_t st);
int copy_f32(float *dst, const float *src, int64_t n, hipStream_t st);
int rhs_rk_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *K,
               float *work, int H, uint32_t flags, int rk_mode, const float *y0, const float *const *h_kprev,
               const float *h_c, int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws,
               hipStream_t st, const RkOpt *opt = nullptr);
int rhs_fused_supported(int H, uint32_t flags);
int pack_weight_256(const float *W, float *Wp, hipStream_t st);
// the image at Wp was packed from weights whose in-row range exceeds the split product's guarantee (split16.h: kS16GuardBits): the
// launches that read it take the fp32 matrix-core route (rhs.hip).  False when NDCN_RANGE_GUARD=0.
bool weights_wide_range(const void *Wp);
int set_range_guard(int on);                  // ndcn_set_range_guard
int pack_weight_256_t16(const float *W, void *Wq, hipStream_t st);   // planes of W^T + per-row unscale factors: kS16Bytes + kS16TailBytes
int rhs_fused2_supported(const ndcn_csr *A, int H, uint32_t flags);
int rhs_fused2_variant(int mode, int n_prev);
int64_t rhs_fused2_partials_bytes();
// rhs_fused2_exact.hip: the same contract with the Linear on the fp32 matrix cores (Wp: the fp32 image at the head of the packed scratch)
int rhs_fused2_exact_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b,
                         float *K, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c,
                         int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st,
                         const RkOpt *opt = nullptr);
// mode 0: K only; 1: also y_next = y0 + sum c_m kprev_m + c_new K; 2: also the dopri5 error record into d_out
int rhs_fused2_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b,
                   float *K, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c,
                   int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st,
                   const RkOpt *opt = nullptr);
// fixed-order sum of n {sum, bad} fp64 pairs into d_out[0..1] (one workgroup: deterministic accept / reject)
int partials_finish(const double *partials, int n, double *d_out, hipStream_t st, int accum = 0);
// rhs_fused3.hip: the same contract as rhs_fused2_f32 for operators that carry the 16-row group-record plan (Wq: the
// split weights of pack_weight_256, i.e. Wp + 256 * 256 floats)
int rhs_fused3_supported(const ndcn_csr *A);
int rhs_xadd_supported(const ndcn_csr *A, int H, uint32_t flags, int mode, int n_prev);   // RkOpt::xadd can be honoured
int rhs_adj_supported(const ndcn_csr *A, int H, uint32_t flags, int mode, int n_prev);    // RkOpt::xmask / s_out can be honoured
int rhs_fused3_variant(int mode, int n_prev);
int rhs_fused3_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const void *Wq, const float *b, float *K,
                   uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev,
                   float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const RkOpt *opt = nullptr);
// spmm_sweep.hip: Y = A X through the column-sweep plan (H = 256, no halo panel, alpha = 1, no activation)
int spmm_sweep_supported(const ndcn_csr *A, int H);
int spmm_sweep_f32(const ndcn_csr *A, const float *X, float *Y, hipStream_t st);
int spmm_rec_supported(const ndcn_csr *A, int H);
int spmm_rec_variant(int mode, int n_prev);
int64_t spmm_rec_partials_bytes();
// mode 0: Y = alpha (A X) [relu]; modes 1-3 (NDCN_RK_*): K = relu(A X) plus the RK algebra, as rhs_fused2_f32
int spmm_rec_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, float alpha, uint32_t flags,
                 int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev, float *y_next,
                 float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev = nullptr,
                 const RkOpt *opt = nullptr);
// out[i] = fl(dt[0] * beta[i]), i < n: the effective coefficients of a replayed adaptive step (device-resident step size)
int scale_coef_f32(float *out, const float *beta, const float *dt, int n, hipStream_t st);
int spmm_wide_rk_supported(const ndcn_csr *A, int H);
int spmm_wide_rk_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *K, uint32_t flags, int mode,
                     const float *y0, const float *const *h_kprev, const float *h_c, int n_prev, float *y_next, float rtol,
                     float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev = nullptr,
                     const RkOpt *opt = nullptr);
// rhs_small.hip: the whole ODEFunc (+ RK epilogue, modes as rhs_fused2_f32) in one launch for H <= 128
int rhs_small_supported(const ndcn_csr *A, int H, uint32_t flags);
int rhs_small_wanted(int64_t n_rows, int H, uint32_t flags);      // the same decision from the sizes alone (rhs_work_bytes)
int rhs_small_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *K,
                  int H, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev,
                  float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev = nullptr,
                  const RkOpt *opt = nullptr);
// solve_small.hip: a whole fixed-grid solve (all ticks) in ONE launch for states that fit one CU; h_dt: the n_ticks step sizes in
// the state dtype; out: n_ticks panels.  Euler also has the reverse sweep (traj / g_out: n_ticks + 1 panels, y_0 first).
int solve_small_supported(const ndcn_csr *A, int H, uint32_t flags, int method);
int solve_small_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method, const float *y0,
                    const float *h_dt, int64_t n_ticks, float *out, hipStream_t st, float *keep = nullptr);
int solve_small_keep_supported(const ndcn_csr *A, int H, uint32_t flags);
int solve_small_bwd_supported(const ndcn_csr *A, int H, uint32_t flags, int method);
int solve_small_bwd_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags, int method,
                        const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, float *g_y0, float *g_W,
                        float *g_b, hipStream_t st, const float *keep = nullptr);
// adjoint.hip: func_eval and the three vector-Jacobian products of the adjoint system's right-hand side for ODEFunc
int64_t adjoint_rhs_work_bytes(int64_t n_rows, int H, uint32_t flags);
int adjoint_rhs_f32(const ndcn_csr *A, const ndcn_csr *At, const float *y, const float *a, const float *W, const float *b, float *K,
                    float *vjp_y, float *vjp_W, float *vjp_b, void *work, int H, uint32_t flags, hipStream_t st);
int rhs_fused_packed_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp,
                         const float *b, float *Y, uint32_t flags, hipStream_t st);

// dt_dev (nullable, device): the step size is read from device memory and h_c holds the bare tableau entries - the
// kernel forms fl(dt * c) itself (hipGraph replay of an adaptive step: one captured graph serves every step size)
int rk_combine_f32(float *out, const float *y0, const float *const *h_k, const float *h_c, int n_k, int64_t n,
                   hipStream_t st, const float *dt_dev = nullptr);
int rk_error_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol,
                 float atol, int64_t n, double *d_out, void *d_ws, hipStream_t st, const float *dt_dev = nullptr, int accum = 0);
// VJPs of the dopri5 panel operations (rk_bwd.hip); d_dots receives 8 doubles, d_ws: rk_bwd_ws_bytes() bytes
int64_t rk_bwd_ws_bytes();
int rk_pull_f32(float *out, const float *base, const float *const *h_p, const float *h_c, int n_p, const float *mask, const float *ua,
                const float *ub, double *d_dots, void *d_ws, int64_t n, hipStream_t st);      // tape.hip's reverse pass (not exported)
int rk_dot_diff_f32(const float *g, const float *a, const float *b, double *d_dots, void *d_ws, int64_t n, hipStream_t st);
int rk_combine_bwd_f32(const float *g, const float *const *h_k, const float *h_c, int n_k, float *const *h_gk, const float *const *h_acc,
                       float *gy0, const float *acc_y0, double *d_dots, void *d_ws, int64_t n, hipStream_t st);
int rk_error_bwd_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k, float rtol, float atol,
                     float g_r, double inv_n, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1,
                     const float *const *h_acc, double *d_dots, void *d_ws, int64_t n, hipStream_t st);
int rk_rms_bwd_f32(const float *a, const float *b, const float *y, float rtol, float atol, float coef, float *ga, float *gb,
                   float *gy, int64_t n, hipStream_t st);
int rk_dense_bwd_f32(const float *g, const float *y0, const float *y1, const float *const *h_k, float dt, float x, float *gy0,
                     float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1, const float *const *h_acc,
                     double *d_dots, void *d_ws, int64_t n, hipStream_t st);
int rk_dense_bwd_multi_f32(const float *const *h_g, int nt, const float *y0, const float *y1, const float *const *h_k, float dt,
                           const float *h_x, float *gy0, float *gy1, float *const *h_gk, const float *acc_y0, const float *acc_y1,
                           const float *const *h_acc, double *d_dots, void *d_ws, int64_t n, hipStream_t st);
bool prof_pause(bool on);     // returns the previous state
extern thread_local int g_last_rhs_path;     // ndcn_debug_last_rhs_path     // no launch timing while a stream is being captured
int scaled_sumsq_f32(const float *a, const float *b, const float *y, float rtol, float atol, int64_t n, double *d_out,
                     void *d_ws, hipStream_t st);
int64_t reduce_ws_bytes();
int scaled_sumsq_pair_f32(const float *f, const float *y, float rtol, float atol, int64_t n, double *d_out4, void *d_ws,
                          void *d_ws2, hipStream_t st);
int64_t set_aten_order_max_elems(int64_t v);   // run-time override of the bound below (< 0: back to the environment's); returns the previous override
int64_t aten_order_max_elems();   // rk_error_f32 / scaled_sumsq_f32 reduce panels up to this size in ATen's float32 order (0: disabled)
int interp_fit_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt, float *a,
                   float *b, float *c, float *d, int64_t n, hipStream_t st);
int interp_direct_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                      const float xp[5], float *out, int64_t n, hipStream_t st);
// fit + evaluate nt <= 8 ticks of ONE accepted step in one pass (h_xp: nt x {x^4, x^3, x^2, x, 1}; h_out: nt panels)
int interp_direct_multi_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_cmid, float dt,
                            const float *h_xp, float *const *h_out, int nt, int64_t n, hipStream_t st);
int interp_eval_f32(const float *a, const float *b, const float *c, const float *d, const float *e, const float xp[5],
                    float *out, int64_t n, hipStream_t st);
int fixed_stage_f32(int op, float *out, const float *y, const float *k1, const float *k2, const float *k3,
                    const float *k4, float dt, int64_t n, hipStream_t st, const float *dt_dev = nullptr);

int row_l1_normalize_f32(const float *X, float *Y, int64_t n_rows, int H, hipStream_t st);
int row_l1_normalize_bwd_f32(const float *G, const float *X, float *GX, int64_t n_rows, int H, hipStream_t st);
int gene_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float f, float h, hipStream_t st);
int mutual_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float k, float c, float d, float e, float h,
                   hipStream_t st);

// comm.hip
int comm_world(const ndcn_comm *c);
int comm_allreduce_sum_f64(ndcn_comm *c, double *d_buf, int n, hipStream_t st);
int64_t halo_plan_n_halo(const ndcn_halo_plan *p);
int64_t halo_plan_n_send(const ndcn_halo_plan *p);
const int32_t *halo_plan_send_idx(const ndcn_halo_plan *p);        // device array of n_send own-row indices
int halo_exchange_f32(ndcn_halo_plan *p, const float *X, int H, float *d_pack, float *X_halo, hipStream_t st);

int64_t solver_workspace_bytes(const ndcn_solver_desc *desc);
int solver_create(const ndcn_solver_desc *desc, void *workspace, int64_t ws_bytes, ndcn_solver **out);
int solver_destroy(ndcn_solver *s);
int solver_begin(ndcn_solver *s, const float *y0, double t0, hipStream_t st, bool borrow = false);
int solver_advance(ndcn_solver *s, double next_t, float *out, int64_t budget, hipStream_t st);
int solver_advance_many(ndcn_solver *s, const double *h_ticks, int64_t n_ticks, float *out, hipStream_t st);
int solver_stats(const ndcn_solver *s, double h[6]);
int64_t solver_steplog(const ndcn_solver *s, double *rows, int64_t cap);

}  // namespace ndcn
