// coarse_fp8.hip — the fp8 (e4m3, block-scaled K=64 MFMA) coarse kernels: kernels_coarse_fp8.h (8 waves) and
// kernels_coarse_fp8_w4.h (one wave per SIMD, the default for even K/64 >= 4).
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/cgvec.h"
#include "coarse_launch.h"
#include "kernels_coarse_fp8.h"
#include "kernels_coarse_fp8_w4.h"

extern "C" int cgv_set_error_(int code, const char* msg);

namespace cgv {
namespace {
int status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return CGV_OK;
    return cgv_set_error_(CGV_ERR_HIP, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
}
int set_lds(const void* kern) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
    if (e == hipSuccess) return CGV_OK;
    return cgv_set_error_(CGV_ERR_HIP, (std::string("hipFuncSetAttribute(fp8 coarse kernel): ") + hipGetErrorString(e)).c_str());
}
}  // namespace

int coarse_attrs_fp8() {
    int rc;
    if ((rc = set_lds((const void*)coarse_fp8s_w4_kernel<0>))) return rc;
    if ((rc = set_lds((const void*)coarse_fp8s_w4_kernel<0, 2>))) return rc;
    if ((rc = set_lds((const void*)coarse_fp8s_kernel<COARSE_EMIT>))) return rc;
    if ((rc = set_lds((const void*)coarse_fp8s_kernel<COARSE_DUMP>))) return rc;
    if ((rc = set_lds((const void*)coarse_fp8s_kernel<COARSE_SAMPLE>))) return rc;
    if ((rc = set_lds((const void*)coarse_fp8s_kernel<COARSE_TOP2>))) return rc;
    return CGV_OK;
}

int launch_coarse_fp8(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr size_t lds = COARSE_LDS_BYTES;
    if (mode == COARSE_DUMP) {
        CGV_KLAUNCH(coarse_fp8s_kernel<COARSE_DUMP>, dim3(W), dim3(512), lds, s, a);
        return status("coarse_fp8s_kernel (dump)");
    }
    if (mode == COARSE_SAMPLE) {
        CGV_KLAUNCH(coarse_fp8s_kernel<COARSE_SAMPLE>, dim3(W), dim3(512), lds, s, a);
        return status("coarse_fp8s_kernel (sample)");
    }
    if (mode == COARSE_TOP2) {   // small batches (kernels_coarse.h: Top2): the 8-wave kernel, whose epilogue is tile_epilogue
        if (a.nqt != 1 || a.nq > 64 || !a.floor_ord)
            return cgv_set_error_(CGV_ERR_INTERNAL, "COARSE_TOP2 launched on a shape it does not serve");
        CGV_KLAUNCH(coarse_fp8s_kernel<COARSE_TOP2>, dim3(W), dim3(512), lds, s, a);
        return status("coarse_fp8s_kernel (top-2 cells)");
    }
    if (mode != COARSE_EMIT) return cgv_set_error_(CGV_ERR_INTERNAL, "fp8 coarse kernels: unknown launch mode");
    // one wave per SIMD (even kc >= 4) or the 8-wave kernel (CGV_COARSE=w8, other kc)
#ifdef CGV_ABLATE_BUILD
    static const bool w8 = getenv("CGV_COARSE") && !strcmp(getenv("CGV_COARSE"), "w8");
#else
    constexpr bool w8 = false;
#endif
    if (!w8 && a.kc >= 4 && (a.kc & 1u) == 0) {
#ifdef CGV_ABLATE_BUILD
        static const int abl4 = getenv("CGV_ABLATE_W4") ? atoi(getenv("CGV_ABLATE_W4")) : 0;  // timing only
        if (abl4) {
#define CGV_ABLK4(N)                                                           \
    case N: {                                                                  \
        auto k2 = coarse_fp8s_w4_kernel<N>;                                    \
        if (int rc = set_lds((const void*)k2)) return rc;                      \
        CGV_KLAUNCH(k2, dim3(W), dim3(256), lds, s, a);                 \
        break;                                                                 \
    }
            switch (abl4) {
                CGV_ABLK4(1) CGV_ABLK4(3) CGV_ABLK4(9)
                default: return cgv_set_error_(CGV_ERR_INVALID_ARG, "CGV_ABLATE_W4: unknown mask");
            }
#undef CGV_ABLK4
            return status("coarse_fp8s_w4_kernel (ablation)");
        }
#endif
#ifdef CGV_ABLATE_BUILD
        if (a.kc % 4 == 0 && (a.epi & 512u) != 0) {   // A/B: the epilogue spread over two k-steps (EPI2; knob epi = 513)
            auto k2 = coarse_fp8s_w4_kernel<0, 2, true>;
            if (int rc = set_lds((const void*)k2)) return rc;
            CGV_KLAUNCH(k2, dim3(W), dim3(256), lds, s, a);
            return status("coarse_fp8s_w4_kernel (si, epi2)");
        }
#endif
        if (a.kc % 4 == 0 && (a.epi & 8u) == 0) {   // static issue side, ring-unrolled (epi bit 3 = the dynamic form, for A/B)
            CGV_KLAUNCH((coarse_fp8s_w4_kernel<0, 2>), dim3(W), dim3(256), lds, s, a);
            return status("coarse_fp8s_w4_kernel (si)");
        }
        CGV_KLAUNCH(coarse_fp8s_w4_kernel<0>, dim3(W), dim3(256), lds, s, a);
        return status("coarse_fp8s_w4_kernel");
    }
    CGV_KLAUNCH(coarse_fp8s_kernel<COARSE_EMIT>, dim3(W), dim3(512), lds, s, a);
    return status("coarse_fp8s_kernel");
}

}  // namespace cgv
