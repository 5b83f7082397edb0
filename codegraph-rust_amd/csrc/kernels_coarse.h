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
// Structure (gfx950) — see DESIGN.md §5.1 for the measurements behind each choice:
//   * workgroup = 256 x 256 output tile, 8 waves as 2(M) x 4(N), each wave 128 x 64 =
//     4 x 2 blocks of v_mfma_f32_32x32x16_{bf16,f16} (128 accumulator VGPRs); K streamed in
//     64-element chunks through two 64-KiB LDS stages by global_load_lds_dwordx4;
//   * both operands live in HBM in the BLOCKED layout B64 (common.h): a (tile, chunk)
//     block is 32 KiB contiguous and byte-identical to its LDS image (XOR-swizzled so that
//     ds_read_b128 fragment reads are bank-conflict-free), so every DMA instruction is a
//     linear 1-KiB copy. The kernel is bound by the L2->LDS path (64 KB per chunk per CU):
//     contiguous blocks move at ~60 GB/s/CU, 128-B row pieces at a 1536-B pitch at ~33;
//   * hand-pinned software pipeline: fragments double-buffered in registers (the 6
//     ds_reads of k-step kk+1 are issued right after the first MFMA of k-step kk), the
//     per-chunk "DMA landed" wait + raw s_barrier sits before the LAST k-step so barrier
//     skew, the next DMA issue and the first LDS latency of the next chunk hide under MFMAs;
//   * a workgroup is persistent over a list of corpus tiles for ONE query tile: the
//     (tile, k-chunk) sequence is one flat pipeline across tile boundaries;
//   * MFMA C layout (32x32): lane holds column (= query) lane&31 and 16 rows
//     (= corpus rows) (r&3)+8*(r>>2)+4*(lane>>5): every lane owns ONE query per N-block, so
//     the per-query threshold lives in a register and the epilogue is a v_max tree + one
//     compare per 16 scores; only scores above the query's threshold tau (a lower bound of
//     its final k'-th best, from earlier stages) are appended to that (workgroup, query)
//     candidate list;
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

// fp8 (e4m3): one 16-byte LDS piece holds 16 elements = the lane's share of TWO K=16 MFMAs
// (low / high 8 bytes), so "mma" issues v_mfma_f32_32x32x16_fp8_fp8 twice and the K chunk of
// a 128-byte row is 128 elements. Any K permutation is fine as long as A and B use the same one.
typedef long fp8x16_t __attribute__((ext_vector_type(2)));
template <>
struct Mfma<DT_FP8> {
    typedef fp8x16_t frag;
    static __device__ inline f32x16_t mma(frag a, frag b, f32x16_t c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[0], b[0], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[1], b[1], c, 0, 0, 0);
    }
};

constexpr uint32_t CAND_CAPS = 256;  // entries per (workgroup, query) candidate list

struct CoarseArgs {
    const char* rows;       // corpus, blocked layout B64 (128-byte row chunks), zero padded
    const char* qrows;      // queries, blocked layout B64
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
    uint32_t T1, R, P, j0, cnt, nsplit, nqt, metric;
};

// Order in which the corpus tiles beyond the boot tiles are visited: tile j of
// the sequence is T1 + (j * P) mod R with P ~ 0.618 R coprime to R (a golden-ratio stride), so
// EVERY prefix of the sequence is spread evenly over the corpus: each threshold stage is a
// representative sample of the rows still to come, whatever the insertion order of the corpus.
__host__ __device__ inline uint32_t stage_tile(uint32_t T1, uint32_t R, uint32_t P, uint32_t j) {
    return T1 + (uint32_t)(((uint64_t)j * P) % R);
}

// LDS counter increment (returns the old value). Inline asm on purpose: for a builtin LDS
// atomic hipcc inserts s_waitcnt vmcnt(0) first (it may alias an in-flight LDS-DMA write as
// far as the compiler knows), which made every emitting epilogue wait for the next chunk's DMA.
// cntq is never a DMA destination, so no such wait is needed.
__device__ inline uint32_t lds_inc_rtn(uint32_t* p) {
    uint32_t old;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p;
    const uint32_t one = 1u;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(addr), "v"(one) : "memory");
    return old;
}

__device__ inline void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}


// Fused top-k' epilogue of one corpus tile (shared by the coarse kernel variants).
// MFMA C layout: lane owns query column lane&31 of each N-block and 16 corpus rows
// (r&3)+8*(r>>2)+4*(lane>>5) of each M-block. The accumulators are NOT cleared here: the
// first k-step of the next tile starts from a zero C operand (free in the MFMA encoding).
template <int BM, int BN, int WTM, int WTN, int MB, int NB, bool DUMP>
__device__ __forceinline__ void tile_epilogue(const CoarseArgs& a, f32x16_t (&acc)[MB][NB], uint32_t tile, int wm,
                                              int wn, int lane, uint32_t g, uint32_t qt, const float (&tq)[NB],
                                              const float (&tauv)[NB], const float (&invq)[NB], uint32_t* cntq,
                                              const float* invn_s /* LDS: inverse norms of this tile's 256 rows */,
                                              const float* stat_s /* LDS: 8 block-min + 8 block-max norms */) {
    const uint64_t trow0 = (uint64_t)tile * BM + wm * WTM;
    // Opaque copy of the lane id: stops LICM from hoisting the 16 x MB per-register row
    // offsets out of the K loop (it cost ~50 VGPRs in the first build).
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    lane = lane_o;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const uint64_t brow = trow0 + mb * 32;
        float mn = 0.0f, mx = 0.0f;
        const bool blk_valid = brow < a.n;
        if (blk_valid && a.metric != METRIC_DOT) {  // from LDS: no vector/scalar global load in the epilogue
            mn = stat_s[(wm * WTM) / 32 + mb];
            mx = stat_s[8 + (wm * WTM) / 32 + mb];
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
                            float s = (a.metric == METRIC_DOT) ? v[r] : v[r] * invn_s[row - (uint64_t)tile * BM] * invq[nb];
                            a.dump[(uint64_t)q * a.n + row] = s;
                        }
                    }
                } else {
                    // conservative per-block threshold in raw-accumulator units
                    float t = tq[nb];
                    if (a.metric != METRIC_DOT && fabsf(t) < INFINITY)
                        t = (t >= 0.0f) ? t * mn * (1.0f - 3.8147e-6f) : t * mx * (1.0f + 3.8147e-6f);
                    // max tree written as nested 3-input maxima (v_max3_f32): 16 values in 8 ops;
                    // the four group maxima are reused to skip whole groups in the slow path
                    const float m0 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const float m1 = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
                    const float m2 = fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11]));
                    const float m3 = fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]));
                    const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
                    if (m > t) {
                        const float gm[4] = {m0, m1, m2, m3};
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            if (gm[gq] > t) {
#pragma unroll
                                for (int r4 = 0; r4 < 4; ++r4) {
                                    const int r = gq * 4 + r4;
                                    const float av = v[r];
                                    if (av > t) {
                                        const uint32_t rl =
                                            (uint32_t)(wm * WTM + mb * 32 + (r & 3) + 8 * (r >> 2)) + 4u * (uint32_t)(lane >> 5);
                                        const uint64_t row = (uint64_t)tile * BM + rl;
                                        if (row < a.n) {
                                            // inverse norm from LDS: a vector global load here would need
                                            // s_waitcnt vmcnt(0), i.e. wait for the next chunk's DMA
                                            const float s =
                                                (a.metric == METRIC_DOT) ? av : av * invn_s[rl] * invq[nb];
                                            if (s > tauv[nb]) {
                                                const uint32_t p = lds_inc_rtn(&cntq[ql]);
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
                }
            }
        }
    }
}

template <int DT, bool DUMP>
__global__ __launch_bounds__(512) void coarse_kernel(const CoarseArgs a) {
    constexpr int BM = 256, BN = 256, WN = 4, NW = 8, NT = 512;
    constexpr int WTM = 128, WTN = 64, MB = 4, NB = 2;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int LA = BM / (8 * NW), LB = BN / (8 * NW);  // 4 + 4 DMA instructions per wave per chunk
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + 2 * STAGE);
    float* invn_s = (float*)(smem + 2 * STAGE + BN * 4);  // [2][256], by tile parity
    float* stat_s = invn_s + 2 * 256;                      // [2][16]: 8 block-min + 8 block-max norms

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

    // DMA sources (blocked layout): block (tile, kc) is 32 KiB contiguous; this wave copies its
    // 4 KiB slab (rows 32w..32w+31) of the A block and of the B block, 1 KiB per instruction.
    const uint32_t slab = (uint32_t)wave * 4096u + (uint32_t)lane * 16u;
    const char* bq = a.qrows + (uint64_t)qt * KC * BLOCK_BYTES + slab;  // + kc * 32 KiB
    const char* atile = a.rows + slab;                                  // + (tile*KC + kc) * 32 KiB
    uint32_t lj = 0, lkc = 0, issued = 0, ltile = 0;
    const char* acur = atile;
    auto set_tile = [&](uint32_t j) {
        ltile = stage_tile(a.T1, a.R, a.P, a.j0 + jlo + j);
        acur = atile + (uint64_t)ltile * KC * BLOCK_BYTES;
    };
    if (total > 0) set_tile(0);
    // DMA of chunk `issued`, split in two halves so the caller can interleave them with MFMAs.
    auto issue_half = [&](int half) {
        const int buf = (int)(issued & 1u);
        const uint64_t koff = (uint64_t)lkc * BLOCK_BYTES;
        if (half == 0) {
            char* dA = smem + buf * STAGE + wave * 4096;
#pragma unroll
            for (int jj = 0; jj < LA; ++jj) glds16(acur + koff + jj * 1024, dA + jj * 1024);
        } else {
            char* dB = smem + buf * STAGE + A_BYTES + wave * 4096;
#pragma unroll
            for (int jj = 0; jj < LB; ++jj) glds16(bq + koff + jj * 1024, dB + jj * 1024);
            if (lkc == 0 && wave == 0)  // the tile's 256 inverse norms ride along with its first chunk
                glds16((const char*)a.invn_c + (uint64_t)ltile * 1024 + lane * 16, (char*)(invn_s + (lj & 1u) * 256));
            if (lkc == 0 && wave == 1 && lane < 4) {  // and its 8 + 8 per-32-row-block norm bounds (4 lanes x 16 B)
                const float* sp = ((lane & 2) ? a.blk_max : a.blk_min) + (uint64_t)ltile * 8 + (lane & 1) * 4;
                glds16((const char*)sp, (char*)(stat_s + (lj & 1u) * 16));
            }
            ++issued;
            if (++lkc == KC) {
                lkc = 0;
                ++lj;
                if (issued < total) set_tile(lj);
            }
        }
    };
    auto issue_next = [&]() {
        issue_half(0);
        issue_half(1);
    };

    // fragment read offsets (bytes): row r = base32 + (lane&31), chunk c = 2*kk + (lane>>5),
    // stored at slot c ^ ((r>>1)&7)
    uint32_t xo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xo[kk] = (uint32_t)((((2 * kk + (lane >> 5)) ^ ((lane >> 1) & 7))) << 4);
    const uint32_t aoff = (uint32_t)(wm * WTM + (lane & 31)) * 128;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 128;

    f32x16_t acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    frag fa0[MB], fb0[NB], fa1[MB], fb1[NB];
#define CGV_LOAD_FRAGS(FA, FB, BASE, KK)                                                         \
    {                                                                                            \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) FA[mb] =                               \
            *(const frag*)((BASE) + aoff + mb * 32 * 128 + xo[KK]);                              \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) FB[nb] =                               \
            *(const frag*)((BASE) + boff + nb * 32 * 128 + xo[KK]);                              \
    }
#define CGV_MMA_FIRST(FA, FB) \
    { acc[0][0] = Mfma<DT>::mma(FA[0], FB[0], acc[0][0]); }
#define CGV_MMA_REST(FA, FB)                                                                     \
    {                                                                                            \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) \
            if (mb + nb > 0) acc[mb][nb] = Mfma<DT>::mma(FA[mb], FB[nb], acc[mb][nb]);           \
    }
// first k-step of a tile: C operand = 0 (an inline constant in the MFMA encoding) instead of
// 128 v_mov per wave per tile to clear the accumulators
#define CGV_MMA_FIRST_Z(FA, FB) \
    { acc[0][0] = Mfma<DT>::mma(FA[0], FB[0], zero16); }
#define CGV_MMA_REST_Z(FA, FB)                                                                   \
    {                                                                                            \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) \
            if (mb + nb > 0) acc[mb][nb] = Mfma<DT>::mma(FA[mb], FB[nb], zero16);                \
    }
// One k-step: hipcc's waitcnt pass emits lgkmcnt(0) (not a counted wait) in front of the first
// MFMA that consumes fragments, so the next step's ds_reads are issued right AFTER that first
// MFMA: they then have the remaining 7 MFMAs (224 matrix-pipe cycles) to land.
#define CGV_STEP(FA, FB, NEXT_LOADS)              \
    {                                             \
        CGV_MMA_FIRST(FA, FB);                    \
        __builtin_amdgcn_sched_barrier(0);        \
        NEXT_LOADS;                               \
        __builtin_amdgcn_sched_barrier(0);        \
        CGV_MMA_REST(FA, FB);                     \
        __builtin_amdgcn_sched_barrier(0);        \
    }

    if (total > 0) issue_next();
    if (total > 1) issue_next();
    if (issued == 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // chunk 0 landed, chunk 1 (8 DMA instr) in flight
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my cntq zero-stores done
    __builtin_amdgcn_s_barrier();
    if (total > 0) CGV_LOAD_FRAGS(fa0, fb0, smem, 0);

    uint32_t cj = 0, ckc = 0;
    for (uint32_t it = 0; it < total; ++it) {
        const char* sb = smem + (it & 1u) * STAGE;
        const char* sn = smem + ((it + 1) & 1u) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        if (ckc == 0) {  // kk = 0 of a new tile: accumulate from zero
            CGV_MMA_FIRST_Z(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            CGV_LOAD_FRAGS(fa1, fb1, sb, 1);
            __builtin_amdgcn_sched_barrier(0);
            CGV_MMA_REST_Z(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            CGV_STEP(fa0, fb0, CGV_LOAD_FRAGS(fa1, fb1, sb, 1));  // kk = 0
        }
        CGV_STEP(fa1, fb1, CGV_LOAD_FRAGS(fa0, fb0, sb, 2));  // kk = 1
        CGV_STEP(fa0, fb0, CGV_LOAD_FRAGS(fa1, fb1, sb, 3));  // kk = 2
        // chunk it+1 landed (this wave's DMA) + all my reads of chunk it done -> barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CGV_MMA_FIRST(fa1, fb1);  // kk = 3
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < total) CGV_LOAD_FRAGS(fa0, fb0, sn, 0);  // first fragments of chunk it+1
        __builtin_amdgcn_sched_barrier(0);
        // chunk it+2 -> the buffer chunk it used; its 8 DMA instructions are spread over the
        // shadows of this step's MFMAs instead of stalling the matrix pipe in one burst
        const bool more = issued < total;
        acc[0][1] = Mfma<DT>::mma(fa1[0], fb1[1], acc[0][1]);
        acc[1][0] = Mfma<DT>::mma(fa1[1], fb1[0], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_half(0);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = Mfma<DT>::mma(fa1[1], fb1[1], acc[1][1]);
        acc[2][0] = Mfma<DT>::mma(fa1[2], fb1[0], acc[2][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_half(1);
        __builtin_amdgcn_sched_barrier(0);
        acc[2][1] = Mfma<DT>::mma(fa1[2], fb1[1], acc[2][1]);
        acc[3][0] = Mfma<DT>::mma(fa1[3], fb1[0], acc[3][0]);
        acc[3][1] = Mfma<DT>::mma(fa1[3], fb1[1], acc[3][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ckc == KC - 1)
            tile_epilogue<BM, BN, WTM, WTN, MB, NB, DUMP>(a, acc, stage_tile(a.T1, a.R, a.P, a.j0 + jlo + cj), wm,
                                                          wn, lane, g, qt, tq, tauv, invq, cntq,
                                                          invn_s + (cj & 1u) * 256, stat_s + (cj & 1u) * 16);
        if (++ckc == KC) {
            ckc = 0;
            ++cj;
        }
    }
#undef CGV_STEP
#undef CGV_MMA_FIRST
#undef CGV_MMA_REST
#undef CGV_MMA_FIRST_Z
#undef CGV_MMA_REST_Z
#undef CGV_LOAD_FRAGS

    __syncthreads();
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

// ---------------------------------------------------------------------------------
// boot_kernel: dense coarse scores of the FIRST n_boot corpus rows against all queries,
// straight from global memory (no LDS, no atomics): one wave = one 64 x 64 output block
// (2 x 2 MFMA blocks), K loop unrolled so many 16-B loads are in flight. It bootstraps
// the per-query threshold tau (top-k' of n_boot scores) in ~30 us instead of running the
// big-tile kernel on 4 workgroups. dense: [nq][n_boot].
// ---------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void boot_kernel(const char* __restrict__ rows,
                                                  const char* __restrict__ qrows,
                                                  const float* __restrict__ invn_c,
                                                  const float* __restrict__ invn_q, uint32_t n_boot,
                                                  uint32_t nq, uint32_t ld, int metric,
                                                  float* __restrict__ dense) {
    typedef typename Mfma<DT>::frag frag;
    const int lane = threadIdx.x & 63;
    const uint32_t nrb = (n_boot + 63) / 64;
    const uint32_t rb = blockIdx.x % nrb, qb = blockIdx.x / nrb;
    const int hi = lane >> 5;
    const char* ap[2];
    const char* bp[2];
    uint32_t akey[2], bkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint32_t r = rb * 64 + i * 32 + (lane & 31);
        r = r < n_boot ? r : n_boot - 1;
        ap[i] = rows + blocked_row_base(r, ld, kchunk_of(DT));
        akey[i] = blocked_row_key(r);
        uint32_t q = qb * 64 + i * 32 + (lane & 31);
        q = q < nq ? q : nq - 1;
        bp[i] = qrows + blocked_row_base(q, ld, kchunk_of(DT));
        bkey[i] = blocked_row_key(q);
    }
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const uint32_t ks = ld * Elem<DT>::bytes / 32;  // pairs of 16-byte pieces per row
#pragma unroll 4
    for (uint32_t k = 0; k < ks; ++k) {
        // this lane's 16-byte piece 2k + hi = slot (2k + hi) & 7 of 128-byte row chunk k >> 2
        const uint64_t blk = (uint64_t)(k >> 2) * BLOCK_BYTES;
        const uint32_t c = (2 * k + hi) & 7u;
        frag a0 = *(const frag*)(ap[0] + blk + ((c ^ akey[0]) << 4)), a1 = *(const frag*)(ap[1] + blk + ((c ^ akey[1]) << 4));
        frag b0 = *(const frag*)(bp[0] + blk + ((c ^ bkey[0]) << 4)), b1 = *(const frag*)(bp[1] + blk + ((c ^ bkey[1]) << 4));
        acc[0][0] = Mfma<DT>::mma(a0, b0, acc[0][0]);
        acc[0][1] = Mfma<DT>::mma(a0, b1, acc[0][1]);
        acc[1][0] = Mfma<DT>::mma(a1, b0, acc[1][0]);
        acc[1][1] = Mfma<DT>::mma(a1, b1, acc[1][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t q = qb * 64 + j * 32 + (lane & 31);
        if (q >= nq) continue;
        const float iq = (metric == METRIC_DOT) ? 1.0f : invn_q[q];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = rb * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < n_boot) {
                    const float v = acc[i][j][r];
                    dense[(uint64_t)q * n_boot + row] = (metric == METRIC_DOT) ? v : v * invn_c[row] * iq;
                }
            }
    }
}

}  // namespace cgv
