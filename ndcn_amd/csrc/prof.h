// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's `roofline` leg).
// Disabled by default: a disabled scope costs one load and a branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndcn {

enum ProfKind {
    PROF_SPMM = 0, PROF_LINEAR, PROF_RHS_FUSED, PROF_COMBINE, PROF_ERROR, PROF_SUMSQ, PROF_FIT, PROF_EVAL,
    PROF_STAGE, PROF_GATHER, PROF_DYN,
    // the vector-Jacobian products of the training path (round 5: a roofline per backward kernel, tools/bench_train.py)
    PROF_COMBINE_BWD, PROF_ERROR_BWD, PROF_SUMSQ_BWD, PROF_DENSE_BWD, PROF_LINEAR_GS, PROF_LINEAR_WGRAD, PROF_RELU_BWD,
    PROF_RHS_ADJ_FWD, PROF_RHS_ADJ_T,      // the two halves of odeint_adjoint's right-hand side (ndcn_rhs_rk_adj_f32: s_out / x_mask)
    PROF_NKINDS
};

struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kind, hipStream_t st, double bytes, double flops);
    ~ProfScope();
};

}  // namespace ndcn
