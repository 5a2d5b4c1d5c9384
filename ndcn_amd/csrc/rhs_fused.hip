// Fused ODEFunc right-hand side (SpMM -> LDS -> fp32 MFMA Linear -> bias -> ReLU).  Placeholder gate:
// the fused kernel lands in a later commit; until then every shape takes the two-kernel path in rhs.hip.
#include "kernels.h"

namespace ndcn {

int rhs_fused_supported(int H, uint32_t flags) {
    (void)H; (void)flags;
    return 0;
}

int rhs_fused_f32(const ndcn_csr *, const float *, const float *, int64_t, const float *, const float *, float *, int,
                  uint32_t, hipStream_t) {
    set_error("rhs_fused: not built for this shape");
    return NDCN_EINVAL;
}

}  // namespace ndcn
