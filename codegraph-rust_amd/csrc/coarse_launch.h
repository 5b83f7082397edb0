// coarse_launch.h — host-side entry points of the coarse (MFMA) kernels. Each storage dtype's
// kernels live in their own translation unit (coarse_bf16.hip, coarse_fp16.hip, coarse_fp8.hip): they are
// the expensive instantiations of the library, and separate objects build in parallel.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include "kernels_coarse.h"

namespace cgv {

// what one launch does with the scores of a tile (tile_epilogue, kernels_coarse.h)
constexpr int COARSE_EMIT = 0;    // candidates above the per-query threshold -> (workgroup, query) lists
constexpr int COARSE_DUMP = 1;    // dense [nq][n] coarse scores (debug / guarantee tests)
constexpr int COARSE_SAMPLE = 2;  // per-lane maxima of every 32 x 32 block -> [nq][16 per tile] (first threshold)
constexpr int COARSE_EMIT_BOOT = 3;  // EMIT whose first tile per workgroup is also the sample: rendezvous, thresholds computed in
                                     // the launch, then emission (bf16 / fp16, K a multiple of 4 chunks, > 1 query tile)
constexpr int COARSE_TOP2 = 4;    // small batches (nq <= 64, one query tile): no threshold - every cell (workgroup, M-half, lane
                                  // half) keeps its two best rows per query + the best score it left out (Top2, kernels_coarse.h)

// 4 stages of (256 + 256) rows x 64 B, per-query counters, 8-deep ring of per-tile inverse norms + bounds
// (+ for fp8 an 8-deep ring of the tiles' 256 scale exponents)
// (+ the threshold ladder of the bf16 / fp16 emitting kernel: the queries' {tau0, delta} and a 2-deep ring of counter words)
constexpr size_t COARSE_LDS_BYTES = 4 * (size_t)(256 + 256) * 64 + (size_t)256 * 4 + 8 * 256 * 4 + 8 * 16 * 4 + 8 * 256 +
                                    (size_t)256 * 16 + 2 * (size_t)256 * 8;

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: called once per device by ensure_kernel_attrs()
int coarse_attrs_bf16();
int coarse_attrs_fp16();
int coarse_attrs_fp8();

// W workgroups on stream s; status = CGV_OK or a CGV_ERR_* with the thread's error message set.
// ev0 / ev1 (optional): HIP events that take the START and the END of this very dispatch (hipExtLaunchKernelGGL: the timestamps of
// the kernel's own completion signal) instead of hipEventRecord in front of and behind the launch (two marker packets on the
// stream). In-process A/B, profiles/r06_launch_events_ab.txt: a 125 k-row shard's batch 0.3263 (markers) -> 0.3234 ms (0.3190 with
// no events at all), C2 1.3312 -> 1.3304 (1.3226): a timed dispatch costs a few microseconds either way, the markers ~3 more.
int launch_coarse_bf16(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int launch_coarse_fp16(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int launch_coarse_fp8(int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// (inside the launch functions: ev0 / ev1 are their parameters; null events = a plain launch)
#define CGV_KLAUNCH(kern, grid, block, lds, s, a) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), s, ev0, ev1, 0u, a)

}  // namespace cgv
