// coarse_fp16.hip — the fp16 instantiations of the coarse (MFMA GEMM + fused top-k') kernels; see kernels_coarse.h.
#include "coarse_launch_2byte.h"

namespace cgv {
int coarse_attrs_fp16() { return coarse_attrs_2byte<DT_FP16>(); }
int launch_coarse_fp16(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    return launch_coarse_2byte<DT_FP16, false>(mode, a, W, s, ev0, ev1);
}
}  // namespace cgv
