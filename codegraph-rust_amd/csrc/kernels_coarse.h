// kernels_coarse.h — the dominant kernel: batched query x corpus similarity as an
// MFMA GEMM (corpus rows = M, queries = N, dim = K, both operands K-contiguous) with
// a fused threshold top-k' epilogue. The B x N score matrix is never written.
//
// Replaces the scoring loop + full sort of ParallelVectorOps::parallel_top_k_search
// (crates/codegraph-vector/src/simd_ops.rs:361-383) for a whole batch of queries:
// it produces, per query, a superset of the top-k' rows by APPROXIMATE score (fp32
// MFMA accumulation order differs from the reference's AVX2 lane order); the exact
// re-score kernel (kernels_exact.h) then reproduces the reference arithmetic
// bit-for-bit on those candidates, the same two-stage shape as
// SemanticSearch::search_by_embedding (search.rs:113-137).
//
// Structure (gfx950):
//   * workgroup = BM x BN output tile (256 x 256, 8 waves as 2(M) x 4(N), each wave
//     128 x 64 = 4 x 2 blocks of v_mfma_f32_32x32x16_{bf16,f16}); K streamed in 64-element
//     chunks, HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     stages, one barrier per chunk;
//   * LDS rows are 128 B; the 16-B chunk c of row r is stored at slot c ^ (r & 7)
//     (XOR swizzle applied on the per-lane SOURCE address, LDS destination stays
//     lane-linear as the DMA requires) so ds_read_b128 fragment reads are <=2-way;
//   * a workgroup is persistent over a list of corpus tiles for ONE query tile: the
//     (tile, k-chunk) sequence is one flat software pipeline, so the next tile's first
//     chunk is in flight while the epilogue of the current tile runs;
//   * MFMA C layout (32x32): lane holds column (= query) lane&31 and 16 rows
//     (= corpus rows) (r&3)+8*(r>>2)+4*(lane>>5): every lane owns ONE query per
//     N-block, so the per-query threshold lives in a register and the epilogue is
//     a v_max3 tree + one compare per 16 scores; only scores above the query's
//     threshold tau (a lower bound of its final k'-th best, from earlier stages)
//     are appended to that (workgroup, query) candidate list;
//   * blockIdx -> (query tile, corpus split) is XCD-aware: the workgroups that stream
//     the same corpus split for different query tiles sit on the same XCD (same L2).
#pragma once
#include "common.h"

namespace cgv {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int DT>
struct Mfma;
template <>
struct Mfma<DT_BF16> {
    typedef bf16x8_t frag;
    static __device__ inline f32x16_t mma(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mfma<DT_FP16> {
    typedef f16x8_t frag;
    static __device__ inline f32x16_t mma(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr uint32_t CAND_CAPS = 256;  // entries per (workgroup, query) candidate list

struct CoarseArgs {
    const char* rows;       // [n][ld] storage dtype (2-byte elements), zero padded
    const char* qrows;      // [nq][ld]
    const float* invn_c;    // [n]
    const float* invn_q;    // [nq]
    const float* blk_min;   // [ceil(n/32)] min row norm per 32-row block
    const float* blk_max;   // [ceil(n/32)]
    const float* tau;       // [nq] emission threshold (coarse score units)
    uint2* cand;            // [W][BN][CAND_CAPS] (score bits, row)
    uint32_t* cand_cnt;     // [W][BN]
    uint32_t* overflow;     // [nq]
    float* dump;            // DUMP mode: dense [nq][n] coarse scores
    uint32_t n, nq, ld, kc;
    uint32_t stage, T1, stride, cnt, nsplit, nqt, metric;
};

// Stage partition of the corpus tiles (see api: staged thresholds).
//   stage 1: tiles [0, T1)
//   stage 2: tiles T1 + j*stride                       (strided sample)
//   stage 3: every remaining tile (stride == 0: all tiles >= T1)
__host__ __device__ inline uint32_t stage_tile(uint32_t stage, uint32_t T1, uint32_t stride,
                                               uint32_t j) {
    if (stage == 1) return j;
    if (stage == 2) return T1 + j * stride;
    if (stride == 0) return T1 + j;
    return T1 + (j / (stride - 1)) * stride + (j % (stride - 1)) + 1;
}

__device__ inline void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int DT, int BM, int BN, int WM, int WN, bool DUMP>
__global__ __launch_bounds__(WM* WN * 64) void coarse_kernel(const CoarseArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN, MB = WTM / 32, NB = WTN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int LA = BM / (8 * NW), LB = BN / (8 * NW);
    static_assert(LA >= 1 && LB >= 1 && MB >= 1 && NB >= 1, "tile/wave shape");
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + 2 * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware logical workgroup id: physical block b runs on XCD b % 8; give each XCD a
    // contiguous range of logical ids so the nqt workgroups of one corpus split share an L2.
    const uint32_t W = gridDim.x;
    uint32_t g = blockIdx.x;
    if ((W & 7u) == 0) g = (blockIdx.x & 7u) * (W >> 3) + (blockIdx.x >> 3);
    const uint32_t qt = g % a.nqt, split = g / a.nqt;

    for (int i = tid; i < BN; i += NT) cntq[i] = 0;

    float tauv[NB], tq[NB], invq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = qt * BN + wn * WTN + nb * 32 + (lane & 31);
        const bool valid = q < a.nq;
        const float tau = valid ? a.tau[q] : INFINITY;
        const float iq = (a.metric == METRIC_DOT) ? 1.0f : (valid ? a.invn_q[q] : 0.0f);
        tauv[nb] = tau;
        invq[nb] = iq;
        tq[nb] = (tau == -INFINITY) ? -INFINITY : (iq == 0.0f ? INFINITY : tau / iq);
    }

    const uint32_t jlo = (uint32_t)(((uint64_t)split * a.cnt) / a.nsplit);
    const uint32_t jhi = (uint32_t)(((uint64_t)(split + 1) * a.cnt) / a.nsplit);
    const uint32_t KC = a.kc;
    const uint32_t total = (jhi - jlo) * KC;

    // per-lane source swizzle: LDS slot p = lane&7 of row (lane>>3) holds chunk p ^ row&7
    const uint32_t cx = (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
    const uint64_t ldb = (uint64_t)a.ld * 2;  // row pitch in bytes
    const char* bsrc[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        uint32_t qr = qt * BN + wave * (BN / NW) + j * 8 + (lane >> 3);
        qr = qr < a.nq ? qr : a.nq - 1;
        bsrc[j] = a.qrows + (uint64_t)qr * ldb + cx;
    }

    auto issue = [&](uint32_t tile, uint32_t kc, int buf) {
        const uint64_t row0 = (uint64_t)tile * BM + wave * (BM / NW) + (lane >> 3);
        char* dA = smem + buf * STAGE + wave * (BM / NW) * 128;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            uint64_t row = row0 + j * 8;
            row = row < a.n ? row : (uint64_t)a.n - 1;
            glds16(a.rows + row * ldb + (uint64_t)kc * 128 + cx, dA + j * 1024);
        }
        char* dB = smem + buf * STAGE + A_BYTES + wave * (BN / NW) * 128;
#pragma unroll
        for (int j = 0; j < LB; ++j) glds16(bsrc[j] + (uint64_t)kc * 128, dB + j * 1024);
    };

    // fragment read offsets (bytes): row r = base + (lane&31), chunk c = 2*kk + (lane>>5)
    uint32_t xo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xo[kk] = (uint32_t)((((2 * kk + (lane >> 5)) ^ (lane & 7))) << 4);
    const uint32_t aoff = (uint32_t)(wm * WTM + (lane & 31)) * 128;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 128;

    f32x16_t acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    if (total > 0) issue(stage_tile(a.stage, a.T1, a.stride, jlo), 0, 0);
    __syncthreads();  // cntq zeroed; stage 0 landed (vmcnt(0) precedes the barrier)

    uint32_t cj = 0, ckc = 0;
    for (uint32_t it = 0; it < total; ++it) {
        const int cur = (int)(it & 1u);
        uint32_t nkc = ckc + 1, nj = cj;
        if (nkc == KC) {
            nkc = 0;
            nj = cj + 1;
        }
        if (it + 1 < total) issue(stage_tile(a.stage, a.T1, a.stride, jlo + nj), nkc, cur ^ 1);

        const char* sbase = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            frag af[MB], bf[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                af[mb] = *(const frag*)(sbase + aoff + mb * 32 * 128 + xo[kk]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[nb] = *(const frag*)(sbase + boff + nb * 32 * 128 + xo[kk]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = Mfma<DT>::mma(af[mb], bf[nb], acc[mb][nb]);
        }

        if (ckc == KC - 1) {
            // ---- fused top-k' epilogue for this corpus tile ----
            const uint32_t tile = stage_tile(a.stage, a.T1, a.stride, jlo + cj);
            const uint64_t trow0 = (uint64_t)tile * BM + wm * WTM;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const uint64_t brow = trow0 + mb * 32;
                float mn = 0.0f, mx = 0.0f;
                const bool blk_valid = brow < a.n;
                if (blk_valid && a.metric != METRIC_DOT) {
                    mn = a.blk_min[brow >> 5];
                    mx = a.blk_max[brow >> 5];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x16_t v = acc[mb][nb];
                    if (blk_valid) {
                        const uint32_t ql = wn * WTN + nb * 32 + (lane & 31);
                        if (DUMP) {
                            const uint32_t q = qt * BN + ql;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const uint64_t row = brow + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                if (row < a.n && q < a.nq) {
                                    float s = (a.metric == METRIC_DOT)
                                                  ? v[r]
                                                  : v[r] * a.invn_c[row] * invq[nb];
                                    a.dump[(uint64_t)q * a.n + row] = s;
                                }
                            }
                        } else {
                            // conservative per-block threshold in raw-accumulator units
                            float t = tq[nb];
                            if (a.metric != METRIC_DOT && fabsf(t) < INFINITY)
                                t = (t >= 0.0f) ? t * mn * (1.0f - 3.8147e-6f)
                                                : t * mx * (1.0f + 3.8147e-6f);
                            float m0 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                            float m1 = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
                            float m2 = fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11]));
                            float m3 = fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]));
                            const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
                            if (m > t) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float av = v[r];
                                    if (av > t) {
                                        const uint64_t row =
                                            brow + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                        if (row < a.n) {
                                            const float s = (a.metric == METRIC_DOT)
                                                                ? av
                                                                : av * a.invn_c[row] * invq[nb];
                                            if (s > tauv[nb]) {
                                                const uint32_t p = atomicAdd(&cntq[ql], 1u);
                                                if (p < CAND_CAPS)
                                                    a.cand[((uint64_t)g * BN + ql) * CAND_CAPS + p] =
                                                        make_uint2(__float_as_uint(s), (uint32_t)row);
                                                else
                                                    a.overflow[qt * BN + ql] = 1u;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
                }
            }
        }
        cj = nj;
        ckc = nkc;
        __syncthreads();
    }

    __syncthreads();
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

}  // namespace cgv
