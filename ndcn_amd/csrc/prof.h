// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's `roofline` leg).
// Disabled by default: a disabled scope costs one load and a branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndcn {

enum ProfKind {
    PROF_SPMM = 0, PROF_LINEAR, PROF_RHS_FUSED, PROF_COMBINE, PROF_ERROR, PROF_SUMSQ, PROF_FIT, PROF_EVAL,
    PROF_STAGE, PROF_GATHER, PROF_DYN, PROF_NKINDS
};

struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kind, hipStream_t st, double bytes, double flops);
    ~ProfScope();
};

}  // namespace ndcn
