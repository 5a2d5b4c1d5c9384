// The range guard's route (rhs.hip: NDCN_PATH_EXACT32): rhs_fused2.hip once more with the fp32 matrix cores as its consumer -
// v_mfma_f32_32x32x2_f32 over the fp32 image of W (exact fp32 fma chain over k: the reference's nn.Linear, neural_dynamics.py:33) -
// 12 gather waves + 4 MFMA waves, every launch mode and the halo / long-row / column-sweep prologues of the default build.
// Entry point: rhs_fused2_exact_f32 (kernels.h); kernels: rhs_fused2_exact_kernel<HALO, MODE, NP>.
#define NDCN_F2_EXACT 1
#include "rhs_fused2.hip"
