// ODEFunc.forward as one entry point: Y = relu(W (A X) + b)   (neural_dynamics.py:20-39, dropout p = 0).
// H = 256 runs the fused kernel (rhs_fused.hip); other widths compose the SpMM and the MFMA Linear through a
// scratch panel.
#include <stdio.h>

#include <atomic>

#include "kernels.h"
#include "split16.h"

namespace ndcn {

int rhs_fused_supported(int H, uint32_t flags);
int64_t rhs_fused_work_bytes(int H);
int rhs_fused_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b,
                  float *Y, float *work, int H, uint32_t flags, hipStream_t st);
int rhs_fused_packed_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp, const float *b, float *Y,
                         uint32_t flags, hipStream_t st);

__global__ __launch_bounds__(256) void relu_copy_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n,
                                                        int relu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = relu ? relu_nan(v) : v;
    }
}

// y = relu(y) in place, 16 bytes per lane (n4 float4 items; the panels of the H = 256 kernels are 16-byte aligned multiples of 4)
__global__ __launch_bounds__(256) void relu_inplace4_kernel(float4 *y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = y[i];
    v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
    y[i] = v;
}

int64_t rhs_work_bytes(int64_t n_rows, int H, uint32_t flags) {
    const bool graph = !(flags & NDCN_F_NO_GRAPH), ctl = !(flags & NDCN_F_NO_CONTROL);
    if (!(graph && ctl)) return 0;
    if (rhs_fused_supported(H, flags)) return rhs_fused_work_bytes(H);       // packed weights
    if (rhs_small_wanted(n_rows, H, flags)) return 16;                          // rhs_small.hip needs none (a token size keeps callers' pointers non-null)
    return n_rows * (int64_t)H * (int64_t)sizeof(float);                        // S = A X between the two kernels
}

int rhs_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *Y,
            float *work, int H, uint32_t flags, hipStream_t st) {
    const bool graph = !(flags & NDCN_F_NO_GRAPH), ctl = !(flags & NDCN_F_NO_CONTROL);
    const uint32_t act = flags & NDCN_F_RELU;
    const int64_t n = A->n_rows;
    if (graph && ctl) {
        if (!work) { set_error("rhs: scratch of ndcn_rhs_work_bytes() bytes required for H=%d", H); return NDCN_EINVAL; }
        if (rhs_fused_supported(H, flags)) {
            if (rhs_fused2_supported(A, H, flags)) {          // operator carries a row-group union plan
                if (!(aligned16(X) && aligned16(Y) && aligned16(W) && aligned16(work) && (!Xh || aligned16(Xh)))) {
                    set_error("rhs: panels must be 16-byte aligned");
                    return NDCN_EINVAL;
                }
                int rc = (flags & NDCN_F_PACKED) ? NDCN_OK : pack_weight_256(W, work, st);
                if (rc) return rc;
                if (weights_wide_range(work))
                    // range guard (split16.h): weights outside the split product's guarantee take the fp32 matrix cores - the same
                    // kernel with the v_mfma_f32_32x32x2_f32 consumer over the fp32 image of W in the same scratch (rhs_fused2_exact.hip)
                    return rhs_fused2_exact_f32(A, X, Xh, n_own, work, b, Y, flags, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f, 0.f,
                                                nullptr, nullptr, st);
                return rhs_fused2_f32(A, X, Xh, n_own, work, b, Y, flags, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f,
                                      0.f, nullptr, nullptr, st);
            }
            return rhs_fused_f32(A, X, Xh, n_own, W, b, Y, work, H, flags, st);
        }
        if (rhs_small_supported(A, H, flags))                     // narrow panels: SpMM -> Linear -> ReLU in one launch
            return rhs_small_f32(A, X, Xh, n_own, W, b, Y, H, flags, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f, 0.f, nullptr,
                                 nullptr, st);
        int rc = spmm_f32(A, X, Xh, n_own, work, H, 1.f, 0, st);
        if (rc) return rc;
        return linear_f32(work, W, b, Y, n, H, H, act, st);
    }
    if (graph) {                                                                // no_control: relu(A x)
        // operators with a column-sweep plan: S = A X by the sweep straight into Y, the ReLU as a streaming pass in place
        if (act && !Xh && H == 256 && spmm_sweep_supported(A, H) && aligned16(X) && aligned16(Y)) {
            int rc = spmm_sweep_f32(A, X, Y, st);
            if (rc) return rc;
            const int64_t total = n * (int64_t)H;
            hipLaunchKernelGGL(relu_inplace4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<float4 *>(Y), total / 4);
            NDCN_LAUNCH_CHECK();
            return NDCN_OK;
        }
        return spmm_f32(A, X, Xh, n_own, Y, H, 1.f, act, st);
    }
    if (ctl) return linear_f32(X, W, b, Y, n, H, H, act, st);                   // no_graph: relu(W x + b)
    const int64_t total = n * (int64_t)H;                                       // neither: relu(x)
    if (total == 0) return NDCN_OK;
    hipLaunchKernelGGL(relu_copy_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, X, Y, total, act ? 1 : 0);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rhs_rk_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *K,
               float *work, int H, uint32_t flags, int rk_mode, const float *y0, const float *const *h_kprev,
               const float *h_c, int n_prev, float *y_next, float rtol, float atol, double *d_out, void *d_ws,
               hipStream_t st, const RkOpt *opt) {
    if (rk_mode == 0) return rhs_f32(A, X, Xh, n_own, W, b, K, work, H, flags, st);
    if (n_prev < 0 || n_prev > 5) { set_error("rhs_rk: n_prev must be 0..5"); return NDCN_EINVAL; }
    const bool both = !(flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL));
    if (both && rhs_fused2_supported(A, H, flags) && rhs_fused2_variant(rk_mode, n_prev)) {
        if (!work) { set_error("rhs_rk: scratch of ndcn_rhs_work_bytes() bytes required"); return NDCN_EINVAL; }
        int rc = (flags & NDCN_F_PACKED) ? NDCN_OK : pack_weight_256(W, work, st);
        if (rc) return rc;
        const bool wide = weights_wide_range(work);
        // options that exist only inside the fused launches (input formed on the staged rows, S written on the side) have no composed
        // form: such a launch stays on the split product and says so (NDCN_PATH_RANGE; the solvers switch the options off instead)
        const bool fused_only = opt && (opt->xadd || opt->xmask || opt->s_out);
        if (!wide || fused_only) {
            rc = rhs_fused2_f32(A, X, Xh, n_own, work, b, K, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol,
                                d_out, d_ws, st, opt);
            if (wide) {
                g_last_rhs_path |= NDCN_PATH_RANGE;
                static std::atomic<bool> warned{false};
                if (!warned.exchange(true))
                    fprintf(stderr, "[ndcn] weights with an in-row range beyond 2^%d on a fused launch that has no fp32 form (x_add / x_mask / "
                                    "s_out): the split fp16 product's guarantee (csrc/split16.h) does not cover them\n", kS16GuardBits);
            }
            return rc;
        }
        // the same launch - gather, Linear, RK epilogue - with the fp32 matrix cores as its consumer (rhs_fused2_exact.hip)
        return rhs_fused2_exact_f32(A, X, Xh, n_own, work, b, K, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol, d_out, d_ws,
                                    st, opt);
    }
    else if (both && rhs_small_supported(A, H, flags)) {
        g_last_rhs_path = NDCN_PATH_SMALL;
        return rhs_small_f32(A, X, Xh, n_own, W, b, K, H, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol, d_out, d_ws,
                             st, nullptr, opt);
    }
    // no_control: relu(A X) with the stage algebra in the epilogue of the group-record SpMM (spmm_rec.hip)
    const bool graph_only = !(flags & NDCN_F_NO_GRAPH) && (flags & NDCN_F_NO_CONTROL);
    if (graph_only && spmm_rec_supported(A, H) && spmm_rec_variant(rk_mode, n_prev) && A->n_rows * (int64_t)1024 < (1ll << 32) &&
        aligned16(X) && aligned16(K) && (!Xh || aligned16(Xh))) {
        g_last_rhs_path = NDCN_PATH_REC | ((Xh && A->n_cols > n_own) ? NDCN_PATH_HALO : 0);
        return spmm_rec_f32(A, X, Xh, n_own, K, 1.f, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol, d_out, d_ws, st,
                            nullptr, opt);
    }
    const bool swept = graph_only && !Xh && H == 256 && spmm_sweep_supported(A, H) && aligned16(X) && aligned16(K);
    if (!swept && graph_only && spmm_wide_rk_supported(A, H) && aligned16(X) && aligned16(K) && (!Xh || aligned16(Xh))) {   // any other graph
        g_last_rhs_path = NDCN_PATH_WIDE | ((Xh && A->n_cols > n_own) ? NDCN_PATH_HALO : 0);
        return spmm_wide_rk_f32(A, X, Xh, n_own, K, flags, rk_mode, y0, h_kprev, h_c, n_prev, y_next, rtol, atol, d_out, d_ws, st,
                                nullptr, opt);
    }
    g_last_rhs_path = 0;
    // composition with the same term order: K first, then the algebra over {kprev..., K}
    int rc = rhs_f32(A, X, Xh, n_own, W, b, K, work, H, flags, st);
    if (rc) return rc;
    const float *kk[6];
    for (int m = 0; m < n_prev; ++m) kk[m] = h_kprev[m];
    kk[n_prev] = K;
    const int64_t n = A->n_rows * (int64_t)H;
    if (rk_mode == 3)
        return fixed_stage_f32(2 + n_prev, y_next, y0, kk[0], n_prev > 0 ? kk[1] : nullptr, n_prev > 1 ? kk[2] : nullptr,
                               n_prev > 2 ? kk[3] : nullptr, h_c[0], n, st, nullptr);
    if (rk_mode == 1) {
        rc = rk_combine_f32(y_next, y0, kk, h_c, n_prev + 1, n, st);
        if (!rc && opt && opt->y_aux && opt->c_aux) rc = rk_combine_f32(opt->y_aux, nullptr, kk, opt->c_aux, n_prev + 1, n, st);
        return rc;
    }
    return rk_error_f32(y0, (opt && opt->y1) ? opt->y1 : X, kk, h_c, n_prev + 1, rtol, atol, n, d_out, d_ws, st, nullptr,
                        (opt && opt->accum) ? 1 : 0);
}

}  // namespace ndcn
