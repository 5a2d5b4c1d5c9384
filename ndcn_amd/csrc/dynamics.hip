// Ground-truth dynamics of the three drivers on an N x 1 state, edge-wise in O(nnz)  (SURVEY A11).
// The reference evaluates these with torch.sparse.mm / torch.mm on N x 1 (heat, gene) and, for the
// mutualistic model, by materialising a dense N x N interaction matrix (mutualistic_dynamics.py:206-216),
// which cannot exist at 10^6 nodes.  Here 8 lanes share a row, each lane walks every 8th stored edge,
// and the partial sums meet in a 3-step shuffle.
#include "common.h"

namespace ndcn {

constexpr int kLanesPerRow = 8;

__device__ __forceinline__ float ipow(float x, float p) {
    // torch's x ** 1 and x ** 2 are exact (x, x*x); anything else goes through powf
    if (p == 1.f) return x;
    if (p == 2.f) return x * x;
    return powf(x, p);
}

struct GeneOp {
    float b, f, h;
    __device__ __forceinline__ float edge(float a, float xi, float xj) const {
        const float p = ipow(xj, h);
        return a * (p / (p + 1.f));                                   // gene_dynamics.py:202
    }
    __device__ __forceinline__ float self(float xi) const { return -b * ipow(xi, f); }
};

struct MutualOp {
    float b, k, c, d, e, h;
    __device__ __forceinline__ float edge(float a, float xi, float xj) const {
        return a * (xj * xi / (d + e * xj + h * xi));                 // mutualistic_dynamics.py:209-211 as executed
    }
    __device__ __forceinline__ float self(float xi) const { return b + xi * (1.f - xi / k) * (xi / c - 1.f); }
};

template <class Op>
__global__ __launch_bounds__(256) void edge_dynamics_kernel(const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                            const float *__restrict__ val, const float *__restrict__ x,
                                                            float *__restrict__ out, int n_rows, Op op) {
    const int rows_per_block = 256 / kLanesPerRow;
    const int li = threadIdx.x % kLanesPerRow;
    for (int r = blockIdx.x * rows_per_block + threadIdx.x / kLanesPerRow; r < n_rows; r += gridDim.x * rows_per_block) {
        const float xi = x[r];
        float acc = 0.f;
        for (int j = rowptr[r] + li; j < rowptr[r + 1]; j += kLanesPerRow) acc += op.edge(val[j], xi, x[colidx[j]]);
#pragma unroll
        for (int off = kLanesPerRow / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, kLanesPerRow);
        if (li == 0) out[r] = op.self(xi) + acc;
    }
}

template <class Op>
static int launch_edge(const ndcn_csr *A, const float *x, float *out, Op op, hipStream_t st) {
    const int n = (int)A->n_rows;
    if (n == 0) return NDCN_OK;
    ProfScope prof(PROF_DYN, st, 8.0 * A->nnz + 4.0 * (n + 1) + 8.0 * n, 8.0 * A->nnz);
    const int rows_per_block = 256 / kLanesPerRow;
    int g = (n + rows_per_block - 1) / rows_per_block;
    if (g > kCus * 16) g = kCus * 16;
    hipLaunchKernelGGL((edge_dynamics_kernel<Op>), dim3(g), dim3(256), 0, st, A->rowptr, A->colidx, A->val, x, out, n, op);
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int gene_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float f, float h, hipStream_t st) {
    return launch_edge(A, x, out, GeneOp{b, f, h}, st);
}

int mutual_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float k, float c, float d, float e, float h,
                   hipStream_t st) {
    return launch_edge(A, x, out, MutualOp{b, k, c, d, e, h}, st);
}

}  // namespace ndcn
