// coarse_bf16.hip — the bf16 instantiations of the coarse (MFMA GEMM + fused top-k') kernels; see kernels_coarse.h.
#include "coarse_launch_2byte.h"

namespace cgv {
int coarse_attrs_bf16() { return coarse_attrs_2byte<DT_BF16>(); }
int launch_coarse_bf16(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#ifdef CGV_ABLATE_BUILD   // `make ABLATE=1`: + the timing-only ablations and the A/B reference instantiations
    return launch_coarse_2byte<DT_BF16, true>(mode, a, W, s, ev0, ev1);
#else
    return launch_coarse_2byte<DT_BF16, false>(mode, a, W, s, ev0, ev1);
#endif
}
}  // namespace cgv
