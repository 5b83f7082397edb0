// coarse_launch_2byte.h — launchers of the bf16 / fp16 coarse kernels (included by coarse_bf16.hip and
// coarse_fp16.hip only: one translation unit per dtype).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/cgvec.h"
#include "coarse_launch.h"
#ifdef CGV_ABLATE_BUILD
#include "experiments/kernels_coarse_w4.h"   // one wave per SIMD, 128 x 128 per wave (A/B against the 8-wave kernel; epi bit 7)
#include "experiments/kernels_coarse_wg2.h"  // two workgroups of 4 waves per CU, 128 x 256 tiles (A/B; epi bit 10)
#endif

extern "C" int cgv_set_error_(int code, const char* msg);

namespace cgv {

inline int coarse_hip_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return CGV_OK;
    return cgv_set_error_(CGV_ERR_HIP, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
}
inline int coarse_set_lds(const void* kern) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
    if (e == hipSuccess) return CGV_OK;
    return cgv_set_error_(CGV_ERR_HIP, (std::string("hipFuncSetAttribute(coarse kernel): ") + hipGetErrorString(e)).c_str());
}

template <int DT>
int coarse_attrs_2byte() {
    int rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_DUMP>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_SAMPLE>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_TOP2, 0, 1, true, 2>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_TOP2, 0, 1, true, 0>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, true>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 1>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 1>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 2>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 2>))) return rc;
#ifdef CGV_ABLATE_BUILD
    // the threshold ladder (kernels_coarse.h): a measured negative result (profiles/r05_tau_ladder_ab.txt), measurement flavour only
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 2, true>))) return rc;
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 2, true>))) return rc;   // the fused sample + emit launch: a measured negative result, kept for A/B in the measurement flavour only
    if ((rc = coarse_set_lds((const void*)coarse_kernel<DT, COARSE_EMIT_BOOT, 0, 1, false, 2>))) return rc;
#endif
    return CGV_OK;
}

// ABLATE: the timing-only ablation instantiations (bf16 only; scripts/gpu_ablate.sh, gpu_clock.sh)
template <int DT, bool ABLATE>
int launch_coarse_2byte(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr size_t lds = COARSE_LDS_BYTES;  // attribute set per device by ensure_kernel_attrs()
    if (mode == COARSE_DUMP) {
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_DUMP>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (dump)");
    }
    if (mode == COARSE_SAMPLE) {
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_SAMPLE>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (sample)");
    }
    if (mode == COARSE_TOP2) {  // small batches: one query tile, every corpus tile read once -> non-temporal corpus stream
        if (a.nqt != 1 || a.nq > 64 || !a.floor_ord)
            return cgv_set_error_(CGV_ERR_INTERNAL, "COARSE_TOP2 launched on a shape it does not serve");
        if (a.kc >= 4 && a.kc % 4 == 0)
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_TOP2, 0, 1, true, 2>), dim3(W), dim3(512), lds, s, a);
        else
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_TOP2, 0, 1, true, 0>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (top-2 cells)");
    }
    if (mode == COARSE_EMIT_BOOT) {  // the fused sample + emit launch: the ring-unrolled, shared-tile form only (search.hip: can_fuse)
#ifdef CGV_ABLATE_BUILD
        if (!(a.kc >= 4 && a.kc % 4 == 0 && a.nqt > 1 && a.boot_sync && a.tau_out && a.dump && a.cnt >= 2 * a.nsplit))
            return cgv_set_error_(CGV_ERR_INTERNAL, "COARSE_EMIT_BOOT launched on a shape it does not serve");
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT_BOOT, 0, 1, false, 2>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (emit + boot)");
#else
        return cgv_set_error_(CGV_ERR_INTERNAL, "COARSE_EMIT_BOOT exists in the measurement flavour only (make ABLATE=1)");
#endif
    }
    if constexpr (ABLATE) {
        // results are wrong when set: only the launch time means anything
        static const int abl = getenv("CGV_ABLATE") ? atoi(getenv("CGV_ABLATE")) : 0;
        if (abl && a.kc >= 4 && a.kc % 4 == 0 && (a.epi & 24u) == 0) {   // ... of the default (ring-unrolled) form
#define CGV_ABLK2(N)                                                           \
    case N: {                                                                  \
        auto k2 = coarse_kernel<DT, COARSE_EMIT, N, 1, false, 2>;              \
        if (int rc = coarse_set_lds((const void*)k2)) return rc;               \
        CGV_KLAUNCH(k2, dim3(W), dim3(512), lds, s, a);                 \
        break;                                                                 \
    }
            switch (abl) {
                CGV_ABLK2(1) CGV_ABLK2(4) CGV_ABLK2(5) CGV_ABLK2(65) CGV_ABLK2(197) CGV_ABLK2(256) CGV_ABLK2(512) CGV_ABLK2(768)
                CGV_ABLK2(1024) CGV_ABLK2(1025) CGV_ABLK2(2048) CGV_ABLK2(4096)
                default: return cgv_set_error_(CGV_ERR_INVALID_ARG, "CGV_ABLATE: no such mask for the ring-unrolled kernel (1, 4, 5, 65, 197, 256, 512, 768, 1024, 1025, 2048, 4096)");
            }
#undef CGV_ABLK2
            return coarse_hip_status("coarse_kernel (ablation, ring-unrolled)");
        }
        if (abl) {
#define CGV_ABLK(N)                                                            \
    case N: {                                                                  \
        auto k2 = coarse_kernel<DT, COARSE_EMIT, N>;                           \
        if (int rc = coarse_set_lds((const void*)k2)) return rc;               \
        CGV_KLAUNCH(k2, dim3(W), dim3(512), lds, s, a);                 \
        break;                                                                 \
    }
            switch (abl) {
                CGV_ABLK(1) CGV_ABLK(2) CGV_ABLK(4) CGV_ABLK(8) CGV_ABLK(10) CGV_ABLK(15) CGV_ABLK(16) CGV_ABLK(32)
                CGV_ABLK(65) CGV_ABLK(197)
                default: return cgv_set_error_(CGV_ERR_INVALID_ARG, "CGV_ABLATE: unknown mask");
            }
#undef CGV_ABLK
            return coarse_hip_status("coarse_kernel (ablation)");
        }
    }
#ifdef CGV_ABLATE_BUILD
    if ((a.epi & 1024u) != 0 && a.kc >= 3 && a.kc % 3 == 0 && a.nqt >= 2 && !a.lad) {  // A/B: two workgroups per CU (kernels_coarse_wg2.h)
        auto kw = coarse_wg2_kernel<DT, false>;
        const hipError_t e = hipFuncSetAttribute((const void*)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_WG2_LDS_BYTES);
        if (e != hipSuccess) return cgv_set_error_(CGV_ERR_HIP, (std::string("hipFuncSetAttribute(coarse_wg2_kernel): ") + hipGetErrorString(e)).c_str());
        static const int per_cu = [&] {   // (measurement flavour: say once what the runtime will co-schedule)
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kw, 256, COARSE_WG2_LDS_BYTES) != hipSuccess) nb = -1;
            fprintf(stderr, "coarse_wg2_kernel: %d workgroups of 4 waves per CU (%zu B of LDS each)\n", nb, (size_t)COARSE_WG2_LDS_BYTES);
            return nb;
        }();
        (void)per_cu;
        CGV_KLAUNCH(kw, dim3(W), dim3(256), COARSE_WG2_LDS_BYTES, s, a);
        return coarse_hip_status("coarse_wg2_kernel");
    }
    if ((a.epi & 128u) != 0 && a.kc >= 4) {  // A/B: one wave per SIMD with the folded epilogue (bit 8 = its boundary-block form)
        if ((a.epi & 256u) != 0) {
            auto kw = coarse_w4_kernel<DT, false>;
            if (int rc = coarse_set_lds((const void*)kw)) return rc;
            CGV_KLAUNCH(kw, dim3(W), dim3(256), lds, s, a);
        } else {
            auto kw = coarse_w4_kernel<DT, true>;
            if (int rc = coarse_set_lds((const void*)kw)) return rc;
            CGV_KLAUNCH(kw, dim3(W), dim3(256), lds, s, a);
        }
        return coarse_hip_status("coarse_w4_kernel");
    }
#endif
    if constexpr (ABLATE) {  // A/B reference: the round-2 epilogue (bf16 build only)
        if ((a.epi & 1u) == 0) {
            auto k0 = coarse_kernel<DT, COARSE_EMIT, 0, 0>;
            if (int rc = coarse_set_lds((const void*)k0)) return rc;
            CGV_KLAUNCH(k0, dim3(W), dim3(512), lds, s, a);
            return coarse_hip_status("coarse_kernel (epi 0)");
        }
    }
    // every corpus tile read by exactly ONE workgroup (a single query tile per XCD group): stream it non-temporally;
    // static issue side when a tile has >= 4 K chunks (epi bit 3 = the dynamic form, for A/B); with a multiple of 4 chunks
    // the stage loop unrolled by the ring size, LDS addresses as immediates (epi bit 4 = the rolled form, for A/B)
    const bool nt = a.nqt == 1 && (a.epi & 2u) == 0, si = a.kc >= 4 && (a.epi & 8u) == 0;
    const bool u4 = si && a.kc % 4 == 0 && (a.epi & 16u) == 0;
    if constexpr (ABLATE) {   // A/B reference (bf16 build only): the ring-unrolled loop with the fragment reads in the barrier's gap
        if (u4 && (a.epi & 32u) != 0 && !nt) {
            auto kp = coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 3>;
            if (int rc = coarse_set_lds((const void*)kp)) return rc;
            CGV_KLAUNCH(kp, dim3(W), dim3(512), lds, s, a);
            return coarse_hip_status("coarse_kernel (reads in the barrier gap)");
        }
    }
#ifdef CGV_ABLATE_BUILD
    if (u4 && a.lad && a.ladc) {   // the threshold ladder: the ring-unrolled form only (every headline shape); knob `ladder`
        if (nt)
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 2, true>), dim3(W), dim3(512), lds, s, a);
        else
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 2, true>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (si, ring-unrolled, ladder)");
    }
#endif
    if (u4) {
        if (nt)
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 2>), dim3(W), dim3(512), lds, s, a);
        else
            CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 2>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (si, ring-unrolled)");
    }
    if (nt && si) {
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, true, 1>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (nt, si)");
    }
    if (si) {
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, false, 1>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (si)");
    }
    if (nt) {
        CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT, 0, 1, true>), dim3(W), dim3(512), lds, s, a);
        return coarse_hip_status("coarse_kernel (nt)");
    }
    CGV_KLAUNCH((coarse_kernel<DT, COARSE_EMIT>), dim3(W), dim3(512), lds, s, a);
    return coarse_hip_status("coarse_kernel");
}

}  // namespace cgv
