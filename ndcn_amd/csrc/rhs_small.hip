// The whole ODEFunc  K = relu(W (A X) + b)  (neural_dynamics.py:27-36) plus the Runge-Kutta algebra that consumes K
// (rk_common.py:45-60,72-78; misc.py:22-25,146-157) for NARROW panels, H <= 128 - every README dynamics command of the
// reference runs H = 20 (heat_dynamics.py:33), the dgnn defaults H = 16 (dgnn.py:44).  One launch instead of SpMM ->
// scratch panel -> Linear (-> separate stage kernel): at these widths a step is launch- and latency-bound, and the
// weights (H x H fp32 <= 64 KiB) fit the LDS of every CU.
//
//   workgroup = 4 waves, persistent over rows; a wave owns one row at a time:
//     gather   lane h (and h + 64 for H > 64) accumulates S[h] = sum_j val_j X[col_j][h] with one fma per entry in stored
//              order - the rounding sequence of spmm_csr_kernel; the (col, val) pairs of the row are loaded one per lane
//              and broadcast with v_readlane
//     linear   lane o forms K[o] = relu(b[o] + sum_h W[o][h] S[h]) as an fma chain over h in ascending order - the
//              sequence of linear_f32 (its fp32 MFMA is bitwise such a chain) - reading W^T from LDS (lanes read
//              consecutive words) and S[h] as an LDS broadcast
//     epilogue row-local panels at column o: the stage algebra in the reference's operator order (separate mul / add)
//   => results are bit-identical to the composed path  ndcn_spmm_f32 -> ndcn_linear_f32 -> rk kernel.
#include <stdlib.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace ndcn {

constexpr int kSmMaxPrev = 5;
constexpr int kSmMaxH = 128;
enum { SM_PLAIN = 0, SM_COMBINE = 1, SM_ERROR = 2, SM_RK4 = 3 };

struct SmallArgs {
    const int *rowptr, *colidx;
    const float *val;
    const float *X, *Xh;
    int n_own, n_rows, H, relu;
    const float *W, *bias;
    float *K;
    float *S_out;                     // nullable: S = A X is written too (RkOpt::s_out - kept for the weight gradient by the training tape)
};
struct SmallEpi {
    const float *y0;
    const float *kprev[kSmMaxPrev];
    float *y_next;
    double *partials;                 // ERROR: [gridDim.x * 4][2]
    float c[kSmMaxPrev + 1];
    int n_prev;
    float rtol, atol;
    const float *c_dev;               // nullable: coefficients in device memory (hipGraph replay)
    const float *y1;                  // ERROR: the state of the error record, by row of this launch
    float *y_aux;                     // COMBINE, nullable: second linear combination (no y0)
    float c2[kSmMaxPrev + 1];
};

// NH = 1: H <= 64 (one column per lane), NH = 2: H <= 128
template <int NH, bool HALO, int MODE>
__global__ __launch_bounds__(256) void rhs_small_kernel(SmallArgs a, SmallEpi e) {
    extern __shared__ float lds[];
    const int H = a.H;
    const int ldw = H + 1;                                  // W^T rows padded: lane o reads word h * ldw + o
    float *s_wt = lds;                                      // [H][ldw]   s_wt[h * ldw + o] = W[o][h]
    float *s_row = lds + H * ldw;                           // [4 waves][H]  the wave's S row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < H * H; i += 256) {
        const int o = i / H, h = i - o * H;
        s_wt[h * ldw + o] = a.W[i];
    }
    __syncthreads();
    float *srow = s_row + wave * H;
    float bias[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u) bias[u] = (a.bias && lane + 64 * u < H) ? a.bias[lane + 64 * u] : 0.f;
    auto coef = [&](int m) { return (MODE != SM_PLAIN && e.c_dev) ? e.c_dev[m] : e.c[m]; };
    double err_sum = 0.0, err_bad = 0.0;
    const int np = e.n_prev;
    const int n_waves = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + wave; r < a.n_rows; r += n_waves) {
        // ---- gather: S = (A X)[r, :]
        float s[NH];
#pragma unroll
        for (int u = 0; u < NH; ++u) s[u] = 0.f;
        const int j0 = a.rowptr[r], j1 = a.rowptr[r + 1];
        for (int jb = j0; jb < j1; jb += 64) {
            const int cnt = min(64, j1 - jb);
            int c = 0;
            float v = 0.f;
            if (lane < cnt) { c = a.colidx[jb + lane]; v = a.val[jb + lane]; }
            for (int j = 0; j < cnt; ++j) {
                int cj = __builtin_amdgcn_readlane(c, j);
                const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
                const float *xr = a.X;
                if (HALO && cj >= a.n_own) { xr = a.Xh; cj -= a.n_own; }
                xr += (size_t)cj * H;
#pragma unroll
                for (int u = 0; u < NH; ++u)
                    if (lane + 64 * u < H) s[u] = fmaf(vj, xr[lane + 64 * u], s[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (lane + 64 * u < H) srow[lane + 64 * u] = s[u];
        if (a.S_out) {
#pragma unroll
            for (int u = 0; u < NH; ++u)
                if (lane + 64 * u < H) a.S_out[(size_t)r * H + lane + 64 * u] = s[u];
        }
        __builtin_amdgcn_wave_barrier();                     // (one wave: LDS accesses of a wave are performed in order)
        // ---- linear + activation: K[o] = relu(sum_h S[h] W[o][h] + b[o])
        float k[NH];
#pragma unroll
        for (int u = 0; u < NH; ++u) k[u] = 0.f;
        for (int h = 0; h < H; ++h) {
            const float sh = srow[h];
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int o = lane + 64 * u;
                if (o < H) k[u] = fmaf(sh, s_wt[h * ldw + o], k[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int o = lane + 64 * u;
            if (o >= H) continue;
            float kn = k[u] + bias[u];
            if (a.relu) kn = relu_nan(kn);
            const size_t idx = (size_t)r * H + o;
            a.K[idx] = kn;
            if (MODE == SM_PLAIN) continue;
            const float y0 = e.y0[idx];
            float km[kSmMaxPrev];
#pragma unroll
            for (int m = 0; m < kSmMaxPrev; ++m) km[m] = m < np ? e.kprev[m][idx] : 0.f;
            if (MODE == SM_RK4) {
                // rk4_alt_step_func (rk_common.py:72-78), same operator order as fixed_stage_kernel ops 2-5
                const float dt = coef(0);
                float sd;
                if (np == 0) sd = (kn * dt) / 3.f;
                else if (np == 1) sd = (km[0] / -3.f + kn) * dt;
                else if (np == 2) sd = ((km[0] - km[1]) + kn) * dt;
                else sd = (((km[0] + km[1] * 3.f) + km[2] * 3.f) + kn) * (dt / 8.f);
                e.y_next[idx] = y0 + sd;
                continue;
            }
            // sum of the stages left to right, the new one last (misc.py:22-25), each product rounded on its own
            float sm = kn * coef(np);
            if (np > 0) {
                float uu = km[0] * coef(0);
#pragma unroll
                for (int m = 1; m < kSmMaxPrev; ++m)
                    if (m < np) uu = uu + km[m] * coef(m);
                sm = uu + sm;
            }
            if (MODE == SM_COMBINE) {
                e.y_next[idx] = y0 + sm;
                if (e.y_aux) {
                    float w2 = kn * e.c2[np];
                    if (np > 0) {
                        float u2 = km[0] * e.c2[0];
#pragma unroll
                        for (int m = 1; m < kSmMaxPrev; ++m)
                            if (m < np) u2 = u2 + km[m] * e.c2[m];
                        w2 = u2 + w2;
                    }
                    e.y_aux[idx] = w2;
                }
            } else {
                const float y1 = e.y1[idx];
                const float tol = e.atol + e.rtol * max_nan(fabsf(y0), fabsf(y1));
                const float z = sm / tol;
                err_sum += (double)(z * z);
                err_bad += (double)(int)(!(fabsf(y1) <= 3.402823466e38f));
            }
        }
        __builtin_amdgcn_wave_barrier();                     // the row buffer is reused by this wave's next row
    }
    if (MODE == SM_ERROR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            err_sum += __shfl_down(err_sum, off, 64);
            err_bad += __shfl_down(err_bad, off, 64);
        }
        if (lane == 0) {
            e.partials[2 * (blockIdx.x * 4 + wave)] = err_sum;
            e.partials[2 * (blockIdx.x * 4 + wave) + 1] = err_bad;
        }
    }
}

// The one-launch form pays where a step is launch- and latency-bound; a wave's Linear is a serial fma chain per row (that is
// what makes it bit-identical to the composed kernels), so beyond a few thousand rows the composed path - row SpMM into a
// scratch panel, MFMA Linear, stage kernel - is faster (10^5-node grid, H = 128: 0.72 ms here, 0.08 ms composed; measured
// crossovers, tools/micro/rhs_small_time.py: H = 128 between 2 000 and 5 000 rows, H = 64 at 10 000, H = 20 between 10 000 and
// 32 000: n H <= 2^18).  Both give the same bits, so the switch is invisible; rhs_work_bytes() asks the same question to size
// the scratch.
int rhs_small_wanted(int64_t n_rows, int H, uint32_t flags) {
    static const bool enabled = [] { const char *e = getenv("NDCN_RHS_SMALL"); return !(e && e[0] == '0'); }();
    static const int64_t max_elems = [] { const char *e = getenv("NDCN_RHS_SMALL_MAX"); return (e && *e) ? atoll(e) : (int64_t)1 << 18; }();
    if (!enabled || H < 1 || H > kSmMaxH) return 0;
    if (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) return 0;
    return n_rows * (int64_t)H <= max_elems ? 1 : 0;
}

int rhs_small_supported(const ndcn_csr *A, int H, uint32_t flags) { return A ? rhs_small_wanted(A->n_rows, H, flags) : 0; }

int rhs_small_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b, float *K,
                  int H, uint32_t flags, int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev,
                  float *y_next, float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev,
                  const RkOpt *opt) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    if (n_prev < 0 || n_prev > kSmMaxPrev || (mode == SM_RK4 && n_prev > 3)) { set_error("rhs_small: bad stage count"); return NDCN_EINVAL; }
    SmallArgs a;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val; a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.n_rows = n_rows;
    a.H = H; a.relu = (flags & NDCN_F_RELU) ? 1 : 0; a.W = W; a.bias = b; a.K = K;
    a.S_out = (opt && opt->s_out) ? opt->s_out : nullptr;
    SmallEpi e = {};
    e.y0 = y0; e.y_next = y_next; e.n_prev = n_prev; e.rtol = rtol; e.atol = atol; e.partials = static_cast<double *>(d_ws);
    e.c_dev = c_dev;
    e.y1 = (opt && opt->y1) ? opt->y1 : X;
    e.y_aux = (mode == SM_COMBINE && !c_dev && opt && opt->y_aux && opt->c_aux) ? opt->y_aux : nullptr;
    for (int m = 0; m <= kSmMaxPrev; ++m) e.c2[m] = (e.y_aux && m <= n_prev) ? opt->c_aux[m] : 0.f;
    for (int m = 0; m < kSmMaxPrev; ++m) e.kprev[m] = (m < n_prev && h_kprev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kSmMaxPrev; ++m) e.c[m] = (mode != SM_PLAIN && mode != SM_RK4 && m <= n_prev) ? h_c[m] : 0.f;
    if (mode == SM_RK4) e.c[0] = h_c[0];
    int grid = (n_rows + 3) / 4;
    if (grid > kCus * 4) grid = kCus * 4;                 // ERROR partials: one slot per wave, reduce_ws_bytes() holds kCus * 16
    const size_t lds = ((size_t)H * (H + 1) + 4 * (size_t)H) * sizeof(float);
    const double P = 4.0 * H * (double)n_rows;
    double bytes = 8.0 * A->nnz + 4.0 * (n_rows + 1) + 4.0 * H * (double)(A->n_rows + A->n_cols) + 4.0 * H * H;
    if (mode != SM_PLAIN) bytes += P * (n_prev + 2 + (e.y_aux ? 1 : 0));
    ProfScope prof(PROF_RHS_FUSED, st, bytes, 2.0 * A->nnz * H + 2.0 * (double)n_rows * H * H);
#define NDCN_SM(NH_, HALO_, MODE_)                                                                                    \
    do {                                                                                                              \
        auto kern = rhs_small_kernel<NH_, HALO_, MODE_>;                                                              \
        static std::atomic<unsigned long long> attr_seen{0};                                                                                 \
        if (once_per_device(attr_seen)) {                                                                                              \
            NDCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,              \
                                         (int)(((size_t)kSmMaxH * (kSmMaxH + 1) + 4 * kSmMaxH) * sizeof(float))));     \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, e);                                               \
    } while (0)
#define NDCN_SM_MODE(NH_, HALO_)                                   \
    do {                                                           \
        if (mode == SM_PLAIN) NDCN_SM(NH_, HALO_, SM_PLAIN);       \
        else if (mode == SM_COMBINE) NDCN_SM(NH_, HALO_, SM_COMBINE); \
        else if (mode == SM_ERROR) NDCN_SM(NH_, HALO_, SM_ERROR);  \
        else NDCN_SM(NH_, HALO_, SM_RK4);                          \
    } while (0)
    if (H <= 64) { if (Xh) NDCN_SM_MODE(1, true); else NDCN_SM_MODE(1, false); }
    else { if (Xh) NDCN_SM_MODE(2, true); else NDCN_SM_MODE(2, false); }
#undef NDCN_SM_MODE
#undef NDCN_SM
    NDCN_LAUNCH_CHECK();
    if (mode == SM_ERROR) return partials_finish(e.partials, grid * 4, d_out, st, (opt && opt->accum) ? 1 : 0);
    return NDCN_OK;
}

}  // namespace ndcn
