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
// Structure (gfx950) — see DESIGN.md §5.1 / §10 for the measurements behind each choice:
//   * workgroup = 256 x 256 output tile, 8 waves as 2(M) x 4(N), each wave 128 x 64 =
//     4 x 2 blocks of v_mfma_f32_32x32x16_{bf16,f16} (128 accumulator VGPRs);
//   * both operands live in HBM in the BLOCKED layout B32 (common.h): a (tile, 64-byte K chunk)
//     block is 16 KiB contiguous and byte-identical to its LDS image (XOR-swizzled so that
//     ds_read_b128 fragment reads are bank-conflict-free), so every DMA instruction is a
//     linear 1-KiB copy (contiguous blocks move at ~60 GB/s/CU, row pieces at a pitch at ~33);
//   * K is streamed through a RING of four 32-KiB LDS stages (A chunk + B chunk = two k-steps)
//     by buffer_load_dwordx4 ... lds, three stages ahead, ONE instruction at a time between MFMA
//     groups, retired by a counted s_waitcnt vmcnt(8); the stream never ends (no tail branches);
//   * hand-pinned software pipeline: fragments double-buffered in registers (the 6 ds_reads of
//     the next k-step are issued right after the first MFMA of the current one); one raw
//     s_barrier per stage, placed after the first MFMA of the stage's last k-step; the loop is
//     tile-structured: the first k-step of a tile runs zero-C MFMAs in its own straight-line block;
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
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int V>
struct IntC {
    static constexpr int value = V;
};

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
// a 64-byte row chunk is 64 elements. Any K permutation is fine as long as A and B use the same one.
// (The production fp8 path is kernels_coarse_fp8.h; this one remains for the boot kernel and A/B timing.)
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
    const char* rows;       // corpus, blocked layout B32 (64-byte row chunks), zero padded
    const char* qrows;      // queries, blocked layout B32
    const float* invn_c;    // [n]
    const float* invn_q;    // [nq]
    const float* blk_min;   // [ceil(n/32)] min row norm per 32-row block
    const float* blk_max;   // [ceil(n/32)]
    const float* tau;       // [nq] emission threshold (coarse score units)
    uint2* cand;            // [W][BN][CAND_CAPS] (score bits, row)
    uint32_t* cand_cnt;     // [W][BN]
    uint32_t* overflow;     // [nq]
    float* dump;            // DUMP mode: dense [nq][n] coarse scores; SAMPLE mode: [nq][sample_ld] block maxima
    uint32_t n, nq, ld, kc;
    uint32_t T1, R, P, j0, cnt, nsplit, nqt, metric;
    uint32_t qgroup;        // query tiles that share one XCD (block_to_work); 0 = all of them
    const int8_t* rexp_c;   // fp8 only: [n] per-row power-of-two scale exponents of the corpus ...
    const int8_t* rexp_q;   //           [nq] ... and of the queries (kernels_coarse_fp8.h)
    uint32_t* pace;         // [W] progress words of the workgroups (Pace below); NULL = no pacing
    uint32_t sample_ld;     // SAMPLE mode: floats per query row of `dump`
    uint32_t sample_vals;   // SAMPLE mode: group maxima per (query, tile): 16, 8 or 4 (tile_epilogue)
    uint32_t epi;           // A/B switches (scripts/ab.py): bit 0 clear = round-2 epilogue (bf16 build only); bit 1 = no NT hint;
                            // bits 3-5: issue-side / loop forms (coarse_launch_2byte.h); bit 6: s_setprio 1 for waves 4-7
    // COARSE_EMIT_BOOT (the fused sample + first emitting launch, BootSync below)
    float* tau_out;         // [nq] the first thresholds, written by the launch itself
    uint32_t* boot_sync;    // [4 * nqt] rendezvous words of the query-tile groups (zero at launch)
    uint32_t kprime;        // the threshold is the k'-th largest group maximum
    // Threshold ladder (COARSE_EMIT, bf16 / fp16 kernel; Ladder below): wait-free in-launch tightening of the thresholds
    unsigned long long* lad;   // [nqt * 256] four 16-bit cumulative counters per query: candidates emitted at or above level j; NULL = off
    const float4* ladc;     // [nqt * 256] {tau0, delta, 1 / delta, -}: level j (1..4) of a query starts at tau0 + j * delta; delta <= 0: no ladder
    // COARSE_TOP2 (small batches, Top2 below)
    uint32_t* floor_ord;    // [nq] f2ord of the largest coarse score any cell left OUT of its top-2 (atomicMax; zero at launch)
    float* cellb;           // [4 * nsplit][64] the same per CELL ((split * 2 + wm) * 2 + lane half; -inf: nothing left out), or NULL:
                            // what top2_repair_kernel (kernels_repair.h) finds the cells to re-scan with when a floor is too high
    // The sample launch as an EMITTING launch (round 6; SAMPLE mode of tile_epilogue): its tiles are not scored again
    uint32_t sample_emit;   // SAMPLE mode: != 0 = every lane also keeps the two best rows of its 64 rows per query column (one cell =
                            // (tile, M-half of the wave grid, lane half)) -> scand[workgroup][query][cell][2], and the best
                            // score it left out -> dump[q][sample_floor + seq * 4 + cell] (tau_kernel folds them into floor_ord[q])
    uint32_t sample_floor;  // SAMPLE mode: offset (floats) of the floor values in a query's row of `dump`
    uint2* scand;           // [W of the sample][256][8] (score bits, row): the cells' two best rows (-inf: none)
    uint32_t append_splits; // EMIT: the workgroups of corpus splits [0, append_splits) first move the sample's candidates of their
                            // list index (same g) that score above the query's threshold - known by now - into their lists
};

// Physical workgroup -> (query tile, corpus split), XCD-aware. Block b runs on XCD b % 8 (observed placement,
// used for speed only) and every XCD has its own 4 MiB L2. Two streams compete for it: the corpus split a
// workgroup walks (shared by the workgroups of the same split, read once from HBM if they sit on one XCD)
// and the query tile of each workgroup (256 x ld x esize bytes, re-read for every corpus tile). With 32 query
// tiles (C5, B = 8192) the old mapping put all 32 query tiles of one split on an XCD: 32 x 192 KiB = 6 MiB of
// query rows thrashed the L2 and every corpus tile re-fetched its query tile through the fabric (PMC, c5mini:
// 66.7 GB per launch against 2 GB algorithmic, L2 hit rate 0.48). So an XCD now serves `qgroup` query tiles
// (host: as many as keep the query rows under ~1.5 MiB) x W/8/qgroup corpus splits; the corpus is then streamed
// nqt / qgroup times from HBM, which these MFMA-bound shapes can afford.
// Returns g = split * nqt + qt, the logical id the candidate lists are addressed by.
__device__ inline uint32_t block_to_work(const CoarseArgs& a, uint32_t& qt, uint32_t& split) {
    const uint32_t W = gridDim.x, b = blockIdx.x;
    uint32_t g = b;
    if ((W & 7u) == 0) {
        const uint32_t per = W >> 3, x = b & 7u, l = b >> 3;  // workgroups per XCD, XCD, index on it
        const uint32_t G = a.qgroup ? a.qgroup : a.nqt;
        const uint32_t NG = a.nqt / G;                         // query-tile groups
        if (G * NG == a.nqt && per % G == 0 && NG <= 8 && (8 % NG) == 0) {
            const uint32_t SP = per / G;                       // corpus splits per XCD
            const uint32_t NB = 8 / NG;                        // XCDs per query-tile group
            qt = (x / NB) * G + l % G;
            split = (x % NB) * SP + l / G;
            return split * a.nqt + qt;
        }
        g = x * per + l;  // contiguous logical ids per XCD
    }
    qt = g % a.nqt;
    split = g / a.nqt;
    return g;
}

// Soft lockstep of the workgroups that stream the SAME corpus split for different query tiles of one XCD group
// (block_to_work: logical ids g - qt % G .. + G - 1). They read the same tiles in the same order and only share
// them through the XCD's L2 while they stay within a few tiles of each other - and nothing keeps them there: a
// leader's misses are hidden by the 3-stage DMA lead, so it is never slowed down, and over the 1000+ tiles of a
// C5 launch the group drifts apart until every member streams its own copy from HBM (PMC, c5mini main launch,
// one-wave fp8 kernel: 43 GB fetched for 1.7 GB of corpus x 4 groups, L2 hit rate 0.69). So at every tile
// boundary wave 0 publishes the workgroup's tile count and waits (bounded: never more than ~20 us, whatever the
// others do) while an ACTIVE member of its group is more than PACE_WINDOW tiles behind; the other waves wait for
// it at the stage barrier. Work is partitioned statically, so holding a leader back costs nothing: the launch
// ends with its slowest workgroup either way. 0 = not started / finished (never waited for): the words are zero
// at the start of every search (memset with the flags) and each workgroup clears its own at exit.
constexpr uint32_t PACE_WINDOW = 2;
struct Pace {
    uint32_t* me;         // NULL: pacing off
    const uint32_t* grp;  // the group's G consecutive words
    uint32_t G;
};
__device__ inline Pace pace_init(const CoarseArgs& a, uint32_t g, uint32_t qt) {
    Pace p;
    p.G = a.qgroup ? a.qgroup : a.nqt;
    const bool on = a.pace != nullptr && p.G > 1 && p.G <= 64 && a.nqt % p.G == 0;
    p.me = on ? a.pace + g : nullptr;
    p.grp = on ? a.pace + (g - qt % p.G) : nullptr;
    return p;
}
// called by ONE wave; tiles = tiles this workgroup has started (>= 1)
__device__ inline void pace_step(const Pace& p, uint32_t tiles, int lane) {
    if (!p.me) return;
    if (lane == 0) __hip_atomic_store(p.me, tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int it = 0; it < 64; ++it) {
        const uint32_t v = (uint32_t)lane < p.G ? __hip_atomic_load(p.grp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (!__any(v != 0u && v + PACE_WINDOW < tiles)) break;
        __builtin_amdgcn_s_sleep(8);  // 512 cycles
    }
}
__device__ inline void pace_done(const Pace& p) {
    if (p.me) __hip_atomic_store(p.me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Order in which the corpus tiles are visited: tile j of the sequence is T1 + (j * P) mod R
// with P ~ 0.618 R coprime to R (a golden-ratio stride), so EVERY prefix of the sequence is spread evenly over
// the corpus: each threshold stage is a representative sample of the rows still to come, whatever the insertion
// order of the corpus. (The host passes T1 = 0, R = all tiles: the boot stage is a sample of its own, boot_row().)
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

// Start of an emitting launch that follows an emitting sample (CoarseArgs::append_splits): the sample kept two rows per cell and
// query without knowing any threshold; the threshold exists now, so only the few that pass it (about k' of the 8 x S per query)
// enter the candidate lists - select / final see ~k' more keys, not 512. cntq must be zero and visible (barrier) before the call.
template <int BN, int NT>
__device__ __forceinline__ void append_sample_candidates(const CoarseArgs& a, uint32_t g, uint32_t qt, uint32_t split, uint32_t* cntq,
                                                         int tid) {
    if (split >= a.append_splits) return;   // uniform
    const uint2* src = a.scand + (uint64_t)g * BN * 8u;
    for (int i = tid; i < BN * 8; i += NT) {
        const uint2 e = src[i];
        const uint32_t ql = (uint32_t)i >> 3, q = qt * (uint32_t)BN + ql;
        if (q < a.nq && __uint_as_float(e.x) > a.tau[q]) {   // block_hits' condition; absent entries carry -inf
            const uint32_t p = lds_inc_rtn(&cntq[ql]);
            if (p < CAND_CAPS) a.cand[((uint64_t)g * BN + ql) * CAND_CAPS + p] = e;
            else a.overflow[q] = 1u;
        }
    }
}

__device__ inline void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}


// 3-input maximum as ONE instruction. fmaxf() on MFMA outputs compiles to a canonicalising v_max_f32 x, x per
// input plus the maxima (18 VALU per 16 scores instead of 10): the scores are never signalling NaNs, and a NaN
// accumulator (non-finite inputs are rejected at ingest) would only miss the fast filter. The asm is invisible to
// hipcc's hazard recogniser, so the caller pads the MFMA-result -> VALU-read distance itself (tile_epilogue).
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float vmin2(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmed3(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// maximum of the 16 scores a lane holds of one 32 x 32 block: 8 VALU (7 three-input maxima + one two-input)
__device__ __forceinline__ float block_max(const f32x16_t& v) {
    const float m0 = vmax3(v[0], v[1], v[2]);
    const float m1 = vmax3(v[3], v[4], v[5]);
    const float m2 = vmax3(v[6], v[7], v[8]);
    const float m3 = vmax3(v[9], v[10], v[11]);
    const float m4 = vmax3(v[12], v[13], v[14]);
    const float m5 = vmax3(m0, m1, v[15]);
    const float m6 = vmax3(m2, m3, m4);
    return vmax2(m5, m6);
}

// ---- Threshold ladder: wait-free tightening of the per-query thresholds INSIDE an emitting launch (round 5) ------------------
// A launch's thresholds come from what was seen BEFORE it (the sample, earlier launches): a launch over N rows with a threshold
// learnt from `seen` rows sends k' * N / seen scores per query down the slow path, which is why round 3 splits the corpus into
// staged launches (sample -> 112 k rows -> the rest at C2) - each with its ramp, a select launch and a hit-heavy start. Round 4
// tried to fuse them with a rendezvous and broke even because every form WAITS. Thresholds are monotone lower bounds, so nobody
// has to wait: every query owns four score levels tau0 + j * delta above its first threshold (tau_kernel spaces them with the
// sample's own spread, scaled to where the final k'-th best is expected) and four 16-bit counters packed into one 64-bit word
// of global memory; a lane that appends a candidate with score s adds 1 to every level at or below s with ONE non-returning
// 64-bit atomic (no return value: the hit path never waits for memory); the word of each of the workgroup's 256 queries
// reaches LDS with the tile's side data (2 x 1 KiB DMA per tile), and at the tile boundary every lane raises its queries'
// thresholds to the highest level whose counter has reached k' - at least k' DISTINCT rows at or above that level are in the
// candidate lists, so it is a valid lower bound of the final k'-th best coarse score, however stale the read. One emitting
// launch straight behind the sample then tightens itself as the staged plan did, without its launches.
constexpr int LADDER_LEVELS = 4;
__device__ __forceinline__ float ladder_level_score(float tau0, float delta, int j) { return tau0 + (float)j * delta; }

// Slow path of one 32 x 32 block (some lane holds a score above its conservative threshold t): lanes with a score
// above t compute the precise coarse score acc * invn_c * invn_q and append (score, row) to their (workgroup,
// query) candidate list through an LDS counter. MFMA C layout: the lane owns query column lane&31 of the N-block
// and 16 corpus rows (r&3)+8*(r>>2)+4*(lane>>5) of the M-block.
template <int BM, int BN>
__device__ __forceinline__ void block_hits(const CoarseArgs& a, const f32x16_t& v, float t, float tau, float iq,
                                           uint32_t rl0 /* first row of the block within the tile */, uint32_t ql,
                                           uint32_t tile, int lane, uint32_t g, uint32_t qt, uint32_t* cntq,
                                           const float* invn_s, const float4* ladc_s = nullptr /* LDS: the queries' {tau0, delta, 1/delta} */) {
    unsigned long long lad_add = 0ull;   // this lane's ladder increments of the block (Ladder above): ONE atomic at the end
    // row numbers are formed HERE, behind the branch, from an opaque base (32-bit: a device index holds < 2^32
    // rows): hoisted, they were 16 64-bit additions per tile on the path every block takes
    uint32_t rbase = rl0 + 4u * (uint32_t)(lane >> 5);
    asm volatile("" : "+v"(rbase));
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        // the group maximum skips four compares at a time
        if (vmax3(v[4 * gq], v[4 * gq + 1], vmax2(v[4 * gq + 2], v[4 * gq + 3])) > t) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = gq * 4 + r4;
                const float av = v[r];
                if (av > t) {
                    const uint32_t rl = rbase + (uint32_t)((r & 3) + 8 * (r >> 2));
                    const uint32_t row = tile * (uint32_t)BM + rl;
                    if (row < a.n) {
                        // inverse norm from LDS: a vector global load here would need
                        // s_waitcnt vmcnt(0), i.e. wait for the next chunk's DMA. (Fetching all 16 of the lane's inverse
                        // norms up front - 4 x 16 bytes at the top of this path - measured -0.4 % on the C2 step at the price
                        // of 16 more VGPRs in the kernel: not kept, profiles/r03f_ab.txt.)
                        const float s = (a.metric == METRIC_DOT) ? av : av * invn_s[rl] * iq;
                        if (s > tau) {
                            const uint32_t p = lds_inc_rtn(&cntq[ql]);
                            if (p < CAND_CAPS) {
                                a.cand[((uint64_t)g * BN + ql) * CAND_CAPS + p] = make_uint2(__float_as_uint(s), row);
                                if (ladc_s) {   // the candidate is in a list: count it on the query's ladder (Ladder above)
                                    const float4 lc = ladc_s[ql];
                                    int lv = (int)fminf((s - lc.x) * lc.z, (float)LADDER_LEVELS);   // (1 / delta = 0: no ladder)
                                    // the reader forms a level's score with exactly this expression: a candidate counts at level
                                    // lv only if it is at or above the score the reader will use as the threshold (the quotient
                                    // can overshoot an integer by a rounding error: one step down then)
                                    lv -= (lv > 0 && !(s >= ladder_level_score(lc.x, lc.y, lv))) ? 1 : 0;
                                    if (lv > 0) lad_add += 0x0001000100010001ull >> (16 * (LADDER_LEVELS - lv));
                                }
                            } else {
                                a.overflow[qt * BN + ql] = 1u;
                            }
                        }
                    }
                }
            }
        }
    }
    if (ladc_s && lad_add != 0ull)   // non-returning: nothing waits for it
        (void)__hip_atomic_fetch_add(a.lad + (uint64_t)qt * BN + ql, lad_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// conservative per-block threshold in raw-accumulator units: tq = tau / invn_q; a row of norm n scores
// acc / n / |q|, so acc > tq * (smallest norm of the block) is necessary when tq >= 0, acc > tq * (largest) when
// tq < 0 (2^-18 head room for the rounding of the products)
__device__ __forceinline__ float block_threshold(const CoarseArgs& a, float tq, float mn, float mx) {
    if (a.metric == METRIC_DOT || !(fabsf(tq) < INFINITY)) return tq;
    return (tq >= 0.0f) ? tq * mn * (1.0f - 3.8147e-6f) : tq * mx * (1.0f + 3.8147e-6f);
}

// ---- COARSE_TOP2: threshold-free candidates for SMALL batches (nq <= 64, one query tile) -----------------------------------
// The reference's trait-level call is ONE query (traits.rs:14 search_similar(&self, &[f32], limit); surreal_store.rs:61-85;
// caller search.rs:114-117), and a batch of a few queries streams the corpus at the HBM rate whatever the kernel does with the
// scores - the staged thresholds of the batched path (sample launch, tau_kernel, 1-2 emitting launches, select) are then pure
// latency: ~45 us of launches that exist only to learn a threshold. Here every lane that owns a query column keeps, over ALL
// the rows it sees in the launch (its CELL: one (workgroup, M-half of the wave grid, lane half) = 1/1024 of the corpus on a
// 256-CU part), the two best coarse scores with their rows and `b`, the best score it left out. At the end of the launch the
// cells' top-2 go to the (workgroup, query) candidate lists and max(b) over the cells to floor_ord[q]. final_kernel takes the
// top-k' of the <= 2048 candidates of a query and checks the usual guarantee against tau = max(k'-th best candidate,
// floor): a row outside the candidates is either in no cell's top-2 (coarse <= its cell's b <= floor) or a candidate below the
// k'-th. Three of a query's top-(k + few) rows in ONE of 1024 cells (probability ~C(11,3) / 1024^2 = 1.6e-4 for k = 10 on
// data without structure; a run of near-duplicates does it on purpose) raise the floor above the k-th exact score: the check
// fails and the query is answered by the exact scan, as always - never a wrong answer. ONE launch, no threshold, no sample.
// Query placement of a COARSE_TOP2 launch (round 6). The MFMA C layout gives wave column wn the query columns wn * 64 .. + 63, and
// the two waves of one wave column (wave = wm * 4 + wn) share a SIMD: with the queries in columns 0 .. 63 all the cell updates
// (~10 VALU per score) ran on ONE of the CU's four SIMDs, beside that SIMD's share of the MFMAs - up to 32 queries the launch still
// streamed the corpus at the HBM rate (418 us on C2's corpus), with 64 it took ~620 us (profiles/r06_coalesce_sweep.txt). So
// query j sits in column (j / 16) * 64 + j % 16: 16 queries in the first N-block of every wave column, every SIMD does a quarter
// of the updates, and a wave never has more than 4 blocks per tile to fold (what 32 queries cost before). The host builds the
// spread copy of the query tile (top2_spread_queries_kernel, kernels_prep.h; batches of <= 16 queries are in place already);
// every per-query array (inverse norms, exponents, candidate lists, floor words) stays indexed by the query.
// (TOP2_QPW = 16 queries per wave column and top2_col_of: common.h)
__device__ __forceinline__ uint32_t top2_query_of(int wn, int nb, int lane) {   // query in this lane's column of N-block nb, or none
    const uint32_t l = (uint32_t)lane & 31u;
    return (nb == 0 && l < TOP2_QPW) ? (uint32_t)wn * TOP2_QPW + l : 0xFFFFFFFFu;
}
// query index of the column a lane owns in N-block nb of wave column wn (one query tile in COARSE_TOP2: qt == 0)
template <int MODE, int BN, int WTN>
__device__ __forceinline__ uint32_t column_query(uint32_t qt, int wn, int nb, int lane) {
    if (MODE == 4) return top2_query_of(wn, nb, lane);
    return qt * (uint32_t)BN + (uint32_t)(wn * WTN + nb * 32 + (lane & 31));
}

struct Top2 {
    float s1, s2, b;      // best, second best, best score left out (coarse / invn_q for cosine: the query's own positive factor
    uint32_t r1, r2;      // is applied when the cell is flushed); rows of s1, s2
};
__device__ __forceinline__ void top2_insert(Top2& t, float v, uint32_t row) {
    const bool g1 = v > t.s1, g2 = v > t.s2;    // (NaN: neither - an unused query column's garbage never enters)
    t.b = fmaxf(t.b, g2 ? t.s2 : v);
    t.r2 = g1 ? t.r1 : (g2 ? row : t.r2);
    t.s2 = g1 ? t.s1 : (g2 ? v : t.s2);
    t.r1 = g1 ? row : t.r1;
    t.s1 = g1 ? v : t.s1;
}
// the cells of this wave -> candidate lists (through the per-query LDS counters, like block_hits) and the floor words
template <int BN, int NB>
__device__ inline void top2_flush(const CoarseArgs& a, const Top2 (&t)[NB], const float (&invq)[NB], int wm, int wn, int lane, uint32_t g,
                                  uint32_t qt, uint32_t* cntq) {
    if ((uint32_t)wn * TOP2_QPW >= a.nq) return;   // (uniform: this wave column holds no query)
    constexpr int nb = 0;                          // (the queries sit in the first N-block of every wave column: top2_col_of)
    const uint32_t q = top2_query_of(wn, nb, lane), ql = q;
    if (q >= a.nq) return;
    const float iq = (a.metric == METRIC_DOT) ? 1.0f : invq[nb];
    const float sv[2] = {t[nb].s1, t[nb].s2};
    const uint32_t rv[2] = {t[nb].r1, t[nb].r2};
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (sv[e] > -INFINITY) {
            const float sc = (a.metric == METRIC_DOT) ? sv[e] : sv[e] * iq;   // = (acc * invn_c) * invn_q, block_hits' order
            const uint32_t p = lds_inc_rtn(&cntq[ql]);
            if (p < CAND_CAPS) a.cand[((uint64_t)g * BN + ql) * CAND_CAPS + p] = make_uint2(__float_as_uint(sc), rv[e]);
            else a.overflow[q] = 1u;
        }
    float fb = -INFINITY;
    if (t[nb].b > -INFINITY) {
        fb = (a.metric == METRIC_DOT) ? t[nb].b : t[nb].b * iq;   // monotone in b: bounds every left-out row
        atomicMax(a.floor_ord + q, f2ord(fb + 0.0f));
    }
    // (g == the corpus split of this workgroup: one query tile. 16 consecutive queries per wave column: 64-byte segments)
    if (a.cellb) a.cellb[(((uint64_t)g * 2u + (uint32_t)wm) * 2u + ((uint32_t)lane >> 5)) * 64u + q] = fb;
    (void)qt;
}

// Fused top-k' epilogue of one corpus tile (shared by the coarse kernel variants).
// Fast filter, BRANCH-FREE over all blocks of the wave tile: the maximum of each block's 16 scores (8 VALU) against a
// conservative raw-accumulator threshold, the verdicts collected in a per-lane bit mask; one branch for the whole
// tile, then the slow path of the flagged blocks only. (Round 1 branched per block: 8 - 16 short basic blocks per
// tile, each a dependent chain with nothing to overlap; measured 43 % of the one-wave-per-SIMD fp8 kernel.)
// The accumulators are NOT cleared here: the first k-step of the next tile starts from a zero C operand (free in
// the MFMA encoding).
template <int BM, int BN, int WTM, int WTN, int MB, int NB, int MODE /* 0 emit, 1 dump, 2 sample, 4 top-2 */, bool ACC_AGPR = false>
__device__ __forceinline__ void tile_epilogue(const CoarseArgs& a, f32x16_t (&acc)[MB][NB], uint32_t tile, int wm,
                                              int wn, int lane, uint32_t g, uint32_t qt, const float (&tq)[NB],
                                              const float (&tauv)[NB], const float (&invq)[NB], uint32_t* cntq,
                                              const float* invn_s /* LDS: inverse norms of this tile's 256 rows */,
                                              const float* stat_s /* LDS: 8 block-min + 8 block-max norms */,
                                              uint32_t seq = 0 /* SAMPLE: position of the tile in the sample */,
                                              Top2* t2 = nullptr /* TOP2: the lane's NB cells */) {
    const uint32_t trow0 = tile * (uint32_t)BM + (uint32_t)(wm * WTM);
    // Opaque copy of the lane id: stops LICM from hoisting per-register row offsets out of the K loop
    // (it cost ~50 VGPRs in the first build).
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    lane = lane_o;
    if (MODE == 4) {
        // TOP2 (small batches): the waves whose columns hold queries (16 per wave column, top2_col_of) fold the tile into their
        // cells. Fast skip per 32 x 32 block: no lane's raw maximum reaches its cell's b (conservative raw-accumulator form of b,
        // block_threshold) - after a few tiles that is almost every block of a batch of few queries (unused columns carry b = +inf).
        if ((uint32_t)wn * TOP2_QPW >= a.nq) return;   // uniform per wave: this wave column holds no query (top2_col_of)
        float mn[MB], mx[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            mn[mb] = stat_s[(wm * WTM) / 32 + mb];
            mx[mb] = stat_s[8 + (wm * WTM) / 32 + mb];
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // vmax3 (asm) reads MFMA results: hipcc pads nothing for asm
        {
            constexpr int nb = 0;   // (the queries sit in the first N-block of every wave column)
            Top2 st = t2[nb];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bool blk_valid = trow0 + (uint32_t)(mb * 32) < a.n;  // uniform
                const float t = block_threshold(a, st.b, mn[mb], mx[mb]);
                if (blk_valid && __ballot(block_max(acc[mb][nb]) > t) != 0ull) {
                    const uint32_t rl0 = (uint32_t)(wm * WTM + mb * 32) + 4u * (uint32_t)(lane >> 5);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        // registers 4 g4 .. 4 g4 + 3 of a block are 4 CONSECUTIVE corpus rows: one 16-byte LDS read of their norms
                        const f32x4_t iv = *(const f32x4_t*)(invn_s + rl0 + 8 * g4);
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const uint32_t row = tile * (uint32_t)BM + rl0 + (uint32_t)(8 * g4 + r4);
                            float v = acc[mb][nb][4 * g4 + r4];
                            if (a.metric != METRIC_DOT) v = v * iv[r4];
                            top2_insert(st, row < a.n ? v : -INFINITY, row);
                        }
                    }
                }
            }
            t2[nb] = st;
        }
        return;
    }
    if (MODE == 2) {
        // SAMPLE: no threshold exists yet. Every lane reduces each of its 32 x 32 blocks to the
        // maximum of its 16 coarse scores (16 distinct corpus rows); the k'-th largest of all the group maxima a query
        // collects is a valid first threshold: >= k' DISTINCT rows score at least that much (tau_kernel,
        // kernels_select.h). a.sample_vals values per (query, tile): 16 = one per lane and block (layout
        // seq * 16 + wm * 8 + (lane >> 5) * 4 + mb: one 16-byte store per lane and N-block); 8 / 4 = the maxima of
        // block pairs / of all four blocks of the lane (groups of 32 / 64 rows), for samples of up to 128 / 256 tiles
        // within tau_kernel's 1024 values per query (one query tile: C4 samples with ALL CUs, 256 tiles).
        // Round 6, a.sample_emit: the sample EMITS. Until then its tiles were scored again by the emitting launches (1.6 % of C2's
        // GEMM, 13 % of a 125 k-row shard's). No threshold exists here, so the lane keeps - per query column - the two best of the
        // 64 rows it holds (its cell: 4 blocks x 16 rows) and the best score it left out, like a COARSE_TOP2 cell: the two go to
        // scand (the first emitting launch moves those above the threshold - which exists by then - into its candidate lists:
        // append_sample_candidates), the left-out score to the floor area of the query's dump row; tau_kernel folds the 4 x S floor values of a query into
        // floor_ord[q] and the final kernel checks its guarantee against max(tau, floor), as it does for COARSE_TOP2: a sampled row
        // that is in no list scores at most its cell's left-out score. (Three of a query's top-(k + 1) rows in one 64-row cell of
        // the sample raise the floor above e_k - C(k + 1, 3) x (S / R)^3 / 256^2 per query: the final kernel re-scores the offending
        // cell itself, sample_floor_repair in kernels_select.h.)
        static_assert(MODE != 2 || MB == 4, "sample layout assumes 4 M-blocks per wave");
        const bool emit = a.sample_emit != 0u;   // uniform
        // The cell's three best values by a max / med3 / min network on floats whose 6 low mantissa bits carry the value's
        // position in the lane's sequence (mb * 16 + r): 6 VALU per score (scale, tag, 4 for the network) instead of the 13 of a
        // compare-and-select insertion with row bookkeeping (the epilogue of a one-tile launch runs with the matrix pipe idle in
        // all 8 waves: 128 scores per lane at 13 VALU cost the sample launch 8.6 us, profiles/r06_sample_emit_first_form.txt).
        // The tag perturbs a score by < 64 ulp = 2^-17 relative: the candidates' coarse scores are off by that much more
        // (coarse_eps_scale prices it), and the floor is raised by it. Rows beyond the corpus (its last tile) are NONE.
        constexpr float NONE = -3.0e38f;
        const bool full = (uint64_t)tile * BM + BM <= (uint64_t)a.n;   // uniform: every row of the tile exists
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const uint32_t q = qt * BN + wn * WTN + nb * 32 + (lane & 31);
            float m[MB];
            float s1 = NONE, s2 = NONE, b3 = NONE;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float mm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t rl = (uint32_t)(wm * WTM + mb * 32 + (r & 3) + 8 * (r >> 2)) + 4u * (uint32_t)(lane >> 5);
                    const float sc = (a.metric == METRIC_DOT) ? acc[mb][nb][r] : acc[mb][nb][r] * invn_s[rl];
                    const bool valid = full || tile * (uint32_t)BM + rl < a.n;
                    if (valid) mm = fmaxf(mm, sc);
                    if (emit) {
                        float v = __uint_as_float((__float_as_uint(sc) & ~63u) | (uint32_t)(mb * 16 + r));
                        if (!full) v = valid ? v : NONE;
                        b3 = vmax2(b3, vmin2(s2, v));   // (one instruction each: fmaxf would canonicalise its inputs first)
                        s2 = vmed3(s1, s2, v);
                        s1 = vmax2(s1, v);
                    }
                }
                m[mb] = (mm == -INFINITY) ? -INFINITY : mm * invq[nb];
            }
            if (q < a.nq) {
                float* dst = a.dump + (uint64_t)q * a.sample_ld + seq * a.sample_vals;
                const uint32_t lg = (uint32_t)(wm * 2 + (lane >> 5));  // the lane's row group within the tile: 0..3
                if (a.sample_vals == 16u) {
                    *(float4*)(dst + lg * 4u) = make_float4(m[0], m[1], m[2], m[3]);
                } else if (a.sample_vals == 8u) {
                    *(float2*)(dst + lg * 2u) = make_float2(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
                } else {
                    dst[lg] = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
                }
                if (emit) {
                    const uint32_t ql = (uint32_t)(wn * WTN + nb * 32 + (lane & 31));
                    const float iq = (a.metric == METRIC_DOT) ? 1.0f : invq[nb];
                    auto row_of = [&](float sv) {   // position tag -> corpus row
                        const uint32_t p = __float_as_uint(sv) & 63u;
                        return tile * (uint32_t)BM + (uint32_t)(wm * WTM) + (p >> 4) * 32u + (p & 3u) + 8u * ((p >> 2) & 3u) + 4u * (uint32_t)(lane >> 5);
                    };
                    // = (acc * invn_c) * invn_q, block_hits' order; one 16-byte store: the cell's two entries are adjacent
                    const float c1 = (s1 > -1.0e38f) ? ((a.metric == METRIC_DOT) ? s1 : s1 * iq) : -INFINITY;
                    const float c2 = (s2 > -1.0e38f) ? ((a.metric == METRIC_DOT) ? s2 : s2 * iq) : -INFINITY;
                    *(uint4*)(a.scand + ((uint64_t)g * BN + ql) * 8u + lg * 2u) =
                        make_uint4(__float_as_uint(c1), row_of(s1), __float_as_uint(c2), row_of(s2));
                    // every score the cell left out is below b3 + 64 ulp (the tags): raise the floor by 2^-17 of its magnitude
                    const float fb = b3 + fabsf(b3) * 7.6294e-6f;
                    a.dump[(uint64_t)q * a.sample_ld + a.sample_floor + seq * 4u + lg] =
                        (b3 > -1.0e38f) ? ((a.metric == METRIC_DOT) ? fb : fb * iq) : -INFINITY;
                }
            }
        }
        return;
    }
    if (MODE == 1) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint32_t q = qt * BN + wn * WTN + nb * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t rl = (uint32_t)(wm * WTM + mb * 32 + (r & 3) + 8 * (r >> 2)) + 4u * (uint32_t)(lane >> 5);
                    const uint32_t row = tile * (uint32_t)BM + rl;
                    if (row < a.n && q < a.nq)
                        a.dump[(uint64_t)q * a.n + row] =
                            (a.metric == METRIC_DOT) ? acc[mb][nb][r] : acc[mb][nb][r] * invn_s[rl] * invq[nb];
                }
            }
        return;
    }
    // the block norm bounds of all M-blocks in one LDS round trip
    float mn[MB], mx[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        mn[mb] = stat_s[(wm * WTM) / 32 + mb];
        mx[mb] = stat_s[8 + (wm * WTM) / 32 + mb];
    }
    // vmax3 (inline asm) reads MFMA results: an 8/16-pass MFMA's result needs up to 18 wait states before a VALU may
    // read it, and hipcc pads nothing for asm
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    // With the accumulators in AGPRs (one wave per SIMD: ACC_AGPR) every score reaches the VALU through a
    // v_accvgpr_read, which hipcc hoists as far up as it can: without the empty re-definitions below it reads all 256
    // accumulators ahead of the filter AND keeps the copies alive for the slow path (512 registers + spills).
    uint32_t hit = 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const bool blk_valid = trow0 + (uint32_t)(mb * 32) < a.n;  // uniform
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float t = block_threshold(a, tq[nb], mn[mb], mx[mb]);
            hit |= (blk_valid && block_max(acc[mb][nb]) > t) ? (1u << (mb * NB + nb)) : 0u;
            if (ACC_AGPR && (nb & 1)) asm volatile("" : "+a"(acc[mb][nb - 1]), "+a"(acc[mb][nb]));
        }
    }
    if (__builtin_expect(hit != 0, 0)) {  // cold: the handlers (tens of KiB of code) stay out of the hot path
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (hit & (1u << (mb * NB + nb))) {
                    if (ACC_AGPR) asm volatile("" : "+a"(acc[mb][nb]));
                    block_hits<BM, BN>(a, acc[mb][nb], block_threshold(a, tq[nb], mn[mb], mx[mb]), tauv[nb], invq[nb],
                                       (uint32_t)(wm * WTM + mb * 32), (uint32_t)(wn * WTN + nb * 32 + (lane & 31)), tile, lane,
                                       g, qt, cntq, invn_s);
                }
    }
}

// Emitting epilogue of the 8-wave bf16 / fp16 kernel with the conservative thresholds thr[mb][nb] PRECOMPUTED (the
// kernel forms them in the MFMA gaps of the tile's last k-step, under matrix-pipe cover): what is left between the
// last MFMA of a tile and the first of the next is, per 32 x 32 block, the maximum of the lane's 16 scores (8 VALU),
// one compare and a scalar OR of the compare mask - ~90 VALU per wave instead of ~200 (r03a clock ablation: the
// epilogue cost 0.14 of the main launch's 1.17 ms while the matrix pipe sat idle in all 8 waves). Blocks are flagged
// wave-wide: the slow path runs block_hits for every lane of a flagged block (lanes without a hit fall through its
// group maxima). Blocks beyond the end of the corpus hold zero accumulators; block_hits drops rows >= n.
template <int BM, int BN, int WTM, int WTN, int MB, int NB>
__device__ __forceinline__ void tile_filter_emit(const CoarseArgs& a, f32x16_t (&acc)[MB][NB], const float (&thr)[MB][NB],
                                                 uint32_t tile, int wm, int wn, int lane, uint32_t g, uint32_t qt,
                                                 const float (&tauv)[NB], const float (&invq)[NB], uint32_t* cntq,
                                                 const float* invn_s, const float4* ladc_s = nullptr) {
    int lane_o = lane;  // opaque copy: no hoisting of per-register row offsets out of the K loop (see tile_epilogue)
    asm volatile("" : "+v"(lane_o));
    lane = lane_o;
    // vmax3 (inline asm) reads MFMA results: up to 18 wait states, and hipcc pads nothing for asm. The maxima are plain
    // (movable) asm: the pad is tied to the accumulators of the k-step's LAST four MFMAs (serpentine order), so no read
    // of those can be scheduled above it; the first four blocks' MFMAs are >= 4 matrix instructions old by then.
    static_assert(MB == 4 && NB == 2, "pad tied to the serpentine order of a 4 x 2 wave tile");
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][1]), "+v"(acc[3][0])::"memory");
    unsigned long long anyhit = 0ull;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) anyhit |= __ballot(block_max(acc[mb][nb]) > thr[mb][nb]);
    if (__builtin_expect(anyhit != 0ull, 0)) {  // cold
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (__ballot(block_max(acc[mb][nb]) > thr[mb][nb]) != 0ull)
                    block_hits<BM, BN>(a, acc[mb][nb], thr[mb][nb], tauv[nb], invq[nb], (uint32_t)(wm * WTM + mb * 32),
                                       (uint32_t)(wn * WTN + nb * 32 + (lane & 31)), tile, lane, g, qt, cntq, invn_s, ladc_s);
    }
}

// ---- COARSE_EMIT_BOOT: sample + first threshold + first emitting pass in ONE launch (round 4) --------------------------
// MEASUREMENT FLAVOUR ONLY (make ABLATE=1; knob `fuse_sample` of scripts/ab.py): a measured negative result, kept so that
// the A/B in profiles/r04_fused_launch_ab.txt can be repeated. HISTORY.md §9.1 has the numbers.
// Round 3 spends three launches on the first threshold: a sample launch (one tile per workgroup, block maxima only), tau_kernel,
// then the first emitting launch, which scores the sample tiles AGAIN - 32 + 7 us + two launch boundaries + a second ramp per
// batch. Here the first tile of every workgroup's walk IS its sample: at the tile's end the workgroup publishes the tile's group
// maxima (what the sample launch wrote), meets the other workgroups of its QUERY TILE (the nsplit <= 64 workgroups whose maxima
// its queries need - not the grid) at rendezvous A, computes the thresholds of its share of the tile's 256 queries (one wave per
// query, kth_largest_wave_top over the nsplit x 16 maxima), meets them again at rendezvous B so that all 256 thresholds are
// visible, and filters the tile's accumulators - which never left their registers - with them; from then on it is an ordinary
// emitting walk. Four forms were built and measured (one MI355X, in-process A/B against the three launches, C2 step / 125 k-row
// shard step): fences around every hand-off +89 / +86 us (16 buffer_wbl2 + 16 buffer_inv per workgroup); fence-free with the
// accumulators stashed in global memory across the wait +19 / +18; arrive at boundary 0, resolve at boundary 1, both tiles
// filtered out of the stash +20 / +22 (64 MB written and read back per stashed tile: more than scoring it again); this form,
// accumulators in registers, a register-light selection, no early fragment reads at the peeled boundary: +1 / -1 us.
// Break-even, because (i) a rendezvous right behind the first tile waits out the launch's own ramp skew (the workgroups of a
// launch start over ~10 us), (ii) two rendezvous round trips + the selection are ~15 us, and (iii) the kernel sits at the
// register limit: with 128 accumulators live across the block the allocator spills a few kernel-long values whose reloads sit
// in the hit paths of the whole walk.
// Hand-offs use NO fences: what other workgroups read (maxima, thresholds) is written with agent-scope write-through (sc1)
// stores, drained with s_waitcnt vmcnt(0) before the arrival, and read with agent-scope loads (MI355X_MICROARCH.md: "sc1
// payload -> vmcnt(0) -> sc1 flag"). Rendezvous = one monotonic arrival counter per query tile and phase; ONE lane polls with
// relaxed loads + s_sleep, the other waves wait at s_barrier.
// It needs the group co-resident, which a plain launch cannot promise (another process' kernel, a second batch in flight on
// the device): every wait is BOUNDED (BOOT_TIMEOUT_TICKS of the 100 MHz wall clock). A workgroup that times out raises the
// group's `degraded` word - every later wait of the group ends on it - marks its 256 queries `overflow` (the final kernel
// sends them through the exact scan: correct, slow) and walks on with an infinite threshold. Nothing hangs, nothing is wrong.
constexpr uint64_t BOOT_TIMEOUT_TICKS = 200000;  // 2 ms
struct BootSync {
    uint32_t* arrive_a;   // workgroups whose maxima are published
    uint32_t* arrive_b;   // workgroups whose thresholds are published
    uint32_t* degraded;   // != 0: somebody gave up waiting
};
__device__ inline BootSync boot_sync_of(const CoarseArgs& a, uint32_t qt) {
    return BootSync{a.boot_sync + 4u * qt, a.boot_sync + 4u * qt + 1u, a.boot_sync + 4u * qt + 2u};
}
// The threshold computation of one workgroup's share of its query tile.
__device__ inline void boot_thresholds(const float* dump, float* tau_out, uint32_t sample_ld, uint32_t kprime,
                                                          uint32_t q0, uint32_t qstep, uint32_t qend, uint32_t nq, int lane) {
    for (uint32_t q = q0; q < qend && q < nq; q += qstep) {
        const float t = kth_largest_wave_top<true, 8>(dump + (uint64_t)q * sample_ld, sample_ld, kprime, lane);
        if (lane == 0) __hip_atomic_store(tau_out + q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ONE lane. Arrive: the caller's write-through stores were drained (s_waitcnt vmcnt(0)) before. Wait: for `need` arrivals;
// false when the group is degraded (a timeout here or elsewhere).
__device__ inline void boot_arrive(uint32_t* counter) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool boot_wait(uint32_t* counter, uint32_t* degraded, uint32_t need) {
    const uint64_t t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if (__hip_atomic_load(degraded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (wall_clock64() - t0 > BOOT_TIMEOUT_TICKS) {
            __hip_atomic_store(degraded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// MODE: COARSE_EMIT / COARSE_DUMP / COARSE_SAMPLE (coarse_launch.h; tile_epilogue).
// ABL: timing-only ablation mask for scripts/gpu_ablate.sh / gpu_clock.sh (results are WRONG for ABL != 0):
// 1 = skip the epilogue, 2 = skip the DMA, 4 = skip the barrier, 8 = skip the fragment reads
// (MFMA on zeros), 16 = skip the vmcnt wait; 32 = global_load...lds instead of buffer_load...lds
// (results stay correct for 32); 64 = DMA for the first ring of stages only (the LDS keeps REAL data);
// 128 = fragments read once, before the loop (MFMA on real data without LDS reads). HISTORY.md §10.2 quotes the numbers.
// 1024 = the query (B) operand for free: no B DMA, no B fragment reads (an upper bound of what feeding B through registers could buy);
// 2048 = only the B DMA goes (the B fragment reads stay, on stale LDS); 4096 = only the B fragment reads go (the DMA stays).
// 256 / 512 / 768 = the stage's counted wait is vmcnt(4) / (6) / (2) instead of (8): results stay correct, the DMA lead shrinks
// by 1 / 0.5 / 1.5 stages (how much of the 3-stage lead does the kernel need? r03i: two stages are enough).
// EPI, the emitting epilogue: 1 (default) = the conservative thresholds are formed in the MFMA gaps of a tile's last
// k-step, and each 32 x 32 block's filter (8 VALU maxima + one compare) sits right in front of the zero-C MFMA of the
// next tile's first k-step that overwrites the block - one wave's filter runs beside its SIMD partner's MFMA instead of
// all 8 waves filtering while the matrix pipe idles. 0 = the round-2 form (everything at the tile boundary,
// tile_epilogue), kept in the bf16 build as the A/B reference (scripts/ab.py knob `epi`). Measured r03c, C2 main
// launch: 1.070 -> 1.049 ms (EPI 0 -> 1); thresholds in the gaps alone, with the filter left at the boundary: 1.070.
// NTA: the corpus (A operand) DMA carries the non-temporal hint - for launches in which every corpus tile is read by
// exactly one workgroup (one query tile per XCD group: C4).
// SI, static issue side (needs KC >= 4): the DMA stream runs exactly three stages ahead of the MFMAs, so WHERE it crosses into
// the next tile is known per loop iteration (the one with kc + 3 == KC) - no per-stage "stages left?" / "chunk == KC?" scalar
// tests, no 64-bit chunk offset: the ring position and the chunk offset are two running scalars. Past its last tile the stream
// simply runs on into the next tile of the visiting order (valid corpus memory) instead of re-reading the last stage. ~8 fewer
// SALU per stage of ~100 instructions per SIMD - the stage loop sits at the issue limit of ~5 fillers per MFMA gap.
// LAD: the threshold ladder (Ladder above) - its own instantiations, so that the kernel without it is exactly what it was.
template <int DT, int MODE, int ABL = 0, int EPI = 1, bool NTA = false, int SI = 0, bool LAD = false>
__global__ __launch_bounds__(512) void coarse_kernel(const CoarseArgs a) {
    constexpr int BM = 256, BN = 256, WN = 4, NT = 512;
    constexpr int WTM = 128, WTN = 64, MB = 4, NB = 2;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;  // one stage = one 64-byte K chunk of A and B
    constexpr bool EMIT = (MODE == 0 || MODE == 3);           // 3 = COARSE_EMIT_BOOT: emits like 0 after its first tile
    constexpr int NSTAGE = 4, NINV = 8;                       // stage ring; per-tile side-data ring
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + NSTAGE * STAGE);
    float* invn_s = (float*)(smem + NSTAGE * STAGE + BN * 4);  // [NINV][256], by tile sequence number
    float* stat_s = invn_s + NINV * 256;                        // [NINV][16]: 8 block-min + 8 block-max norms
    // threshold ladder (behind the fp8 kernels' exponent ring, which this kernel does not use): the queries' {tau0, delta} and
    // a 2-deep ring of their counter words, refreshed with every tile's side data
    float4* ladc_s = (float4*)((char*)(stat_s + NINV * 16) + NINV * 256);   // [256]
    unsigned long long* lad_s = (unsigned long long*)(ladc_s + BN);         // [2][256]
    const bool lad_on = LAD && EMIT && EPI != 0 && a.lad != nullptr;         // uniform

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    uint32_t qt, split;
    const uint32_t g = block_to_work(a, qt, split);
    const Pace pace = pace_init(a, g, qt);

    // A/B probe (epi bit 6; MI355X_MICROARCH.md "Two waves per SIMD", item 4): static priority for the second-dispatched half
    // of the workgroup, the arbitration loser of every SIMD pair. `wave` is a readfirstlane value: a scalar branch around ONE
    // s_setprio (a per-thread condition would be if-converted into an unconditional one).
    if ((a.epi & 64u) != 0u && wave >= 4) __builtin_amdgcn_s_setprio(1);

    for (int i = tid; i < BN; i += NT) cntq[i] = 0;
    if (EMIT && a.append_splits != 0u) {   // uniform: the first launch behind an emitting sample
        __syncthreads();
        append_sample_candidates<BN, NT>(a, g, qt, split, cntq, tid);
    }

    float tauv[NB], tq[NB], invq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = column_query<MODE, BN, WTN>(qt, wn, nb, lane);
        const bool valid = q < a.nq;
        const float tau = (MODE == 3 || MODE == 4) ? INFINITY : (valid ? a.tau[q] : INFINITY);  // MODE 3: no threshold yet (boot_block); 4: none at all
        const float iq = (a.metric == METRIC_DOT) ? 1.0f : (valid ? a.invn_q[q] : 0.0f);
        tauv[nb] = tau;
        invq[nb] = iq;
        tq[nb] = (tau == -INFINITY) ? -INFINITY : (iq == 0.0f ? INFINITY : tau / iq);
    }
    // Conservative per-block threshold in raw-accumulator units (block_threshold): thr(mb, nb) = ta[nb] * (the block's
    // smallest row norm when tq >= 0, its largest when tq < 0; 1 for the dot metric and for infinite thresholds),
    // 2^-18 head room folded into ta. Formed per tile in the MFMA gaps of its last k-step (CGV_THR below).
    float ta[NB], thr[MB][NB];
    bool tneg[NB], tone[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        tone[nb] = (a.metric == METRIC_DOT) || !(fabsf(tq[nb]) < INFINITY);
        tneg[nb] = tq[nb] < 0.0f;
        ta[nb] = tone[nb] ? tq[nb] : (tneg[nb] ? tq[nb] * (1.0f + 3.8147e-6f) : tq[nb] * (1.0f - 3.8147e-6f));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) thr[mb][nb] = INFINITY;
    }

    // COARSE_TOP2: the lane's cells (one per N-block of the wave tile); unused query columns never take part (b = +inf)
    Top2 t2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const bool active = MODE == 4 && top2_query_of(wn, nb, lane) < a.nq;
        t2[nb].s1 = t2[nb].s2 = -INFINITY;
        t2[nb].b = active ? -INFINITY : INFINITY;
        t2[nb].r1 = t2[nb].r2 = 0xFFFFFFFFu;
    }

    // uniform by construction; readfirstlane makes it provable (the 64-bit divisions run on the VALU and would
    // otherwise leave `total` in a VGPR: every "is there a next stage" test became a vector compare)
    const uint32_t jlo = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)split * a.cnt) / a.nsplit));
    const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)(split + 1) * a.cnt) / a.nsplit));
    const uint32_t KC = a.kc;
    const uint32_t total = (jhi - jlo) * KC;  // pipeline stages of this workgroup

    // Tile sequence (stage_tile): t_{j+1} = t_j + P (mod R), kept incrementally on the issue side
    // (lt) and on the consume side (ct) - the 64-bit modulo is paid once per kernel, not per tile.
    const uint32_t t_first = __builtin_amdgcn_readfirstlane((total > 0) ? stage_tile(a.T1, a.R, a.P, a.j0 + jlo) - a.T1 : 0u);
    auto next_tile = [&](uint32_t t) {
        const uint32_t u = t + a.P;  // P < R <= 2^24 tiles: no overflow
        return u >= a.R ? u - a.R : u;
    };

    // DMA sources (blocked layout): block (tile, kc) is 16 KiB contiguous; this wave copies KiB
    // 2w and 2w+1 (rows 32w..32w+31) of the A block and of the B block, 1 KiB per instruction.
    const uint32_t slab = (uint32_t)wave * 2048u + (uint32_t)lane * 16u;
    const char* bq = a.qrows + (uint64_t)qt * KC * BLOCK_BYTES + slab;  // + kc * 16 KiB
    const char* atile = a.rows + slab;                                  // + (tile*KC + kc) * 16 KiB
    uint32_t lkc = 0, issued = 0, lt = t_first;
    const char* acur = atile + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES;
    // One stage = 4 DMA instructions per wave (A0 A1 B0 B1), issued one at a time so the caller
    // can spread them between MFMAs: a steady one-per-few-MFMAs stream instead of a burst that
    // fills the address queue and stalls the wave (and with it the MFMAs behind it).
    // The stream never ends: past the last stage it re-reads the last stage into the (free) ring
    // slot, so the loop body needs no "is there a next stage" branches and the counted
    // s_waitcnt vmcnt(8) below is valid in every iteration (<= 3 x 32 KiB of extra L2 reads).
    // The copies go through buffer_load ... lds (scalar resource + scalar offset + ONE constant per-lane
    // VGPR offset): no per-instruction 64-bit address arithmetic; measured 2.6 % faster on C2 than the
    // global_load ... lds form (ABL & 32 selects that form for A/B timing).
    constexpr bool BUFDMA = (ABL & 32) == 0;
    const char* abase = a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES;  // uniform
    const char* bbase = a.qrows + (uint64_t)qt * KC * BLOCK_BYTES;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, 0x7fffffff, 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u;
    // (the 1-KiB second piece of a block shares M0 and the scalar offset with the first and steps by the
    //  instruction's immediate offset, which applies to the global AND the LDS address)
    uint32_t d_so = 0;
    char* d_dst = smem;
#define CGV_BDMA(RS, DST, IMM) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, voff, d_so, IMM, 0)
#define CGV_BDMA_A(RS, DST, IMM) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, voff, d_so, IMM, NTA ? 2 : 0)
    uint32_t si_slot = 0;                              // SI: ring position in bytes (stage count x STAGE, masked on use)
    uint32_t si_so = (uint32_t)wave * 2048u;           // SI: chunk offset within the tile + this wave's 2 KiB slab
    auto issue_switch_tile = [&]() {                   // SI: the issue side enters the next tile of the visiting order
        lt = next_tile(lt);
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0, 0x7fffffff,
                                                0x00020000);
        si_so = (uint32_t)wave * 2048u;
    };
    auto issue_q = [&](int q) {
        if constexpr (SI != 0) {
            if ((ABL & 2) || ((ABL & 64) && issued >= (uint32_t)NSTAGE)) {  // timing-only ablations
                if (q == 3) ++issued;
                return;
            }
            if ((ABL & 64) && q == 3) ++issued;
            if (q == 0) {
                d_dst = smem + (si_slot & (uint32_t)(NSTAGE * STAGE - 1)) + wave * 2048;
                d_so = si_so;
                CGV_BDMA_A(rsA, d_dst, 0);
            } else if (q == 1) {
                CGV_BDMA_A(rsA, d_dst, 1024);
            } else if (q == 2) {
                if (!(ABL & (1024 | 2048))) CGV_BDMA(rsB, d_dst + A_BYTES, 0);
            } else {
                if (!(ABL & (1024 | 2048))) CGV_BDMA(rsB, d_dst + A_BYTES, 1024);
                if (SI == 1) si_slot += (uint32_t)STAGE;   // SI >= 2: set by the (unrolled) caller, a constant per iteration
                si_so += BLOCK_BYTES;
            }
            return;
        }
        if ((ABL & 2) || ((ABL & 64) && issued >= (uint32_t)NSTAGE)) {
            if (q == 3) ++issued;
            return;
        }
        const uint64_t koff = (uint64_t)lkc * BLOCK_BYTES;
        if (q == 0) {
            d_dst = smem + (issued & (NSTAGE - 1)) * STAGE + wave * 2048;
            d_so = (uint32_t)koff + (uint32_t)wave * 2048u;
            if (BUFDMA) CGV_BDMA_A(rsA, d_dst, 0); else glds16(acur + koff, d_dst);
        } else if (q == 1) {
            if (BUFDMA) CGV_BDMA_A(rsA, d_dst, 1024); else glds16(acur + koff + 1024, d_dst + 1024);
        } else if (q == 2) {
            if (BUFDMA) CGV_BDMA(rsB, d_dst + A_BYTES, 0); else glds16(bq + koff, d_dst + A_BYTES);
        } else {
            if (BUFDMA) CGV_BDMA(rsB, d_dst + A_BYTES, 1024); else glds16(bq + koff + 1024, d_dst + A_BYTES + 1024);
            ++issued;
            if (issued < total && ++lkc == KC) {
                lkc = 0;
                lt = next_tile(lt);
                acur = atile + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES;
                if (BUFDMA)
                    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0,
                                                            0x7fffffff, 0x00020000);
            }
        }
    };
    // Side data of the tile with sequence number seq (absolute tile T1 + tt): its 256 inverse norms (wave 0) and
    // 8 + 8 per-32-row-block norm bounds (4 lanes of wave 1), issued in the straight-line tile-boundary block
    // where the tile STARTS: the stage loop has no branch besides its back edge. The tile's epilogue runs KC
    // stages later; with KC >= 3 the counted vmcnt(8) of its last stage covers these (>= 8 younger DMA
    // instructions behind them), shorter tiles wait explicitly (side_wait).
    auto issue_side = [&](uint32_t tt, uint32_t seq) {
        if (ABL & 2) return;
        if (wave == 0)
            glds16((const char*)a.invn_c + (uint64_t)(a.T1 + tt) * 1024 + lane * 16, (char*)(invn_s + (seq & (NINV - 1)) * 256));
        if (wave == 1 && lane < 4) {
            const float* sp = ((lane & 2) ? a.blk_max : a.blk_min) + (uint64_t)(a.T1 + tt) * 8 + (lane & 1) * 4;
            glds16((const char*)sp, (char*)(stat_s + (seq & (NINV - 1)) * 16));
        }
        if (lad_on && wave == 2) {   // the counter words of the workgroup's 256 queries as they stand now: read one tile later
            const char* src = (const char*)(a.lad + (uint64_t)qt * BN) + lane * 16;
            char* dst = (char*)(lad_s + (seq & 1u) * BN) + lane * 16;
            glds16(src, dst);
            glds16(src + 1024, dst + 1024);
        }
    };
    // every lane: its queries' thresholds up to the highest ladder level whose counter has reached k' (words of ring slot `slot`)
    auto ladder_raise = [&](uint32_t slot) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const uint32_t ql = (uint32_t)(wn * WTN + nb * 32 + (lane & 31));
            const unsigned long long w = lad_s[slot * BN + ql];
            const float4 lc = ladc_s[ql];
            const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32), kp = a.kprime;
            const int j = (hi >> 16) >= kp ? 4 : ((hi & 0xffffu) >= kp ? 3 : ((lo >> 16) >= kp ? 2 : ((lo & 0xffffu) >= kp ? 1 : 0)));
            const float nt = ladder_level_score(lc.x, lc.y, j);
            if (j > 0 && lc.y > 0.0f && nt > tauv[nb]) {   // (never lowers a threshold; padding queries keep +inf)
                const float iq = invq[nb];
                tauv[nb] = nt;
                tq[nb] = (iq == 0.0f) ? INFINITY : nt / iq;
                tone[nb] = (a.metric == METRIC_DOT) || !(fabsf(tq[nb]) < INFINITY);
                tneg[nb] = tq[nb] < 0.0f;
                ta[nb] = tone[nb] ? tq[nb] : (tneg[nb] ? tq[nb] * (1.0f + 3.8147e-6f) : tq[nb] * (1.0f - 3.8147e-6f));
            }
        }
    };
    auto side_wait = [&]() {  // uniform: short tiles only (D <= 64 elements per 64-byte chunk x 2)
        if (KC < 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };

    // fragment read offsets (bytes): row r = base32 + (lane&31); the lane's piece of k-step kk
    // (0/1 within the stage) is c = 2*kk + (lane>>5), stored at slot c ^ ((r>>2)&3)
    uint32_t xo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) xo[kk] = (uint32_t)((((2 * kk + (lane >> 5)) ^ ((lane >> 2) & 3))) << 4);
    const uint32_t aoff = (uint32_t)(wm * WTM + (lane & 31)) * 64;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 64;

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
    if (ABL & 8) {
        for (int i = 0; i < MB; ++i) fa0[i] = fa1[i] = (frag)0;
        for (int i = 0; i < NB; ++i) fb0[i] = fb1[i] = (frag)0;
    }
#define CGV_LDA(FA, I, BASE, KK) if (!(ABL & (8 | 128))) FA[I] = *(const frag*)((BASE) + aoff + (I) * 2048 + xo[KK]);
#define CGV_LDB(FB, I, BASE, KK) if (!(ABL & (8 | 128 | 1024 | 4096))) FB[I] = *(const frag*)((BASE) + boff + (I) * 2048 + xo[KK]);
#define CGV_LOAD_FRAGS(FA, FB, BASE, KK)                                                         \
    {                                                                                            \
        CGV_LDA(FA, 0, BASE, KK) CGV_LDA(FA, 1, BASE, KK) CGV_LDA(FA, 2, BASE, KK) CGV_LDA(FA, 3, BASE, KK) \
        CGV_LDB(FB, 0, BASE, KK) CGV_LDB(FB, 1, BASE, KK)                                        \
    }
#define CGV_LOAD_FRAGS_A(FA, BASE, KK) \
    { CGV_LDA(FA, 0, BASE, KK) CGV_LDA(FA, 1, BASE, KK) CGV_LDA(FA, 2, BASE, KK) CGV_LDA(FA, 3, BASE, KK) }
#define CGV_LOAD_FRAGS_B(FB, BASE, KK) \
    { CGV_LDB(FB, 0, BASE, KK) CGV_LDB(FB, 1, BASE, KK) }
    // Where the fragment reads of a NEW stage go relative to the stage barrier (B phase). LG = 0: right behind it, in the gap
    // of the barrier (rounds 1-3). LG = 12 (every form but the A/B reference SI == 3): nothing but the barrier in gap 0 -
    // both waves of a SIMD wake up together, and the first thing each does is put its next MFMA into the idle matrix pipe;
    // the 4 A reads follow in gap 1, the 2 B reads in gap 2 (5 MFMAs to land). Measured (r03j, C2 main launch, one process):
    // 1.0001 -> 0.9866 ms; reads one / two gaps later in one piece 0.9889 / 0.9881; C3 shard 4.731 -> 4.661. SI == 3 = the
    // ring-unrolled loop with LG = 0 (A/B reference).
    constexpr int LG = (SI == 3) ? 0 : 12;
#define CGV_MMA(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], acc[MBI][NBI]);
    // first k-step of a tile: C operand = 0 (an inline constant in the MFMA encoding) instead of clearing 128
    // accumulator registers per tile. These run in their own straight-line block at every tile boundary (never as
    // a branch inside the stage loop: that made the register allocator copy the accumulators around phis).
#define CGV_MMAZ(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], zero16);
    // Program-order pins. An MFMA is a pure register operation: instruction selection places it anywhere its
    // operands allow, sched_barrier or not (one build of this kernel had the phase's first MFMA sunk below the
    // fragment reads, so its lgkmcnt(0) waited for the reads just issued). An empty asm that consumes a result
    // (the MFMA writing it comes before this point) or redefines an operand (the MFMAs reading it come after
    // this point) ties them to the chain of side-effecting instructions (LDS reads, DMA, barrier, waits:
    // "memory"), which keeps its order. One gap = what is issued between MFMA (MBI, NBI) and the next one.
#define CGV_GAP(MBI, NBI, NEXT_OPERAND, ACTION)                        \
    asm volatile("" : "+v"(acc[MBI][NBI])::"memory");                  \
    ACTION;                                                            \
    asm volatile("" : "+v"(NEXT_OPERAND)::"memory");
#define CGV_NOP_ACTION
    // One k-step: 8 MFMAs on fragments FA/FB; NA/NB (the other buffer) are filled for the next k-step from LDS
    // stage NBASE, k-step NKK; Q0, Q0+1 = the DMA pieces issued here; FIRST = what follows the first MFMA (the
    // stage's counted wait + barrier in a B phase: by then every wave has all its reads of the previous stage
    // back, so its slot may be overwritten by the DMA of stage s+3, and has waited for its own share of stage s).
    // The 6 fragment reads go right behind the first MFMA (hipcc waits lgkmcnt(0), so they get 7 MFMAs to land;
    // the partner wave of the SIMD covers their issue), a DMA piece behind MFMAs 4 and 6. (One read per gap
    // measured the same: 1.003 vs 1.004 ms on the C2 main launch; giving the two waves of a SIMD different DMA
    // gaps - waves 0-3 early, 4-7 late in the k-step - measured 4-8 % SLOWER: HISTORY.md §10.2.)
#define CGV_KSTEP(MMA, FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                                   \
    {                                                                                                            \
        /* serpentine block order: consecutive MFMAs share one operand block (0.5 % on the C2 main launch) */    \
        MMA(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], FIRST; if (!(LG != 0 && (NKK) == 0)) CGV_LOAD_FRAGS(NA, NB_, NBASE, NKK)) \
        MMA(0, 1, FA, FB) CGV_GAP(0, 1, FA[1], if (LG == 1 && (NKK) == 0) CGV_LOAD_FRAGS(NA, NB_, NBASE, NKK);   \
                                               if (LG == 12 && (NKK) == 0) CGV_LOAD_FRAGS_A(NA, NBASE, NKK))     \
        MMA(1, 1, FA, FB) CGV_GAP(1, 1, FB[0], if (LG == 2 && (NKK) == 0) CGV_LOAD_FRAGS(NA, NB_, NBASE, NKK);   \
                                               if (LG == 12 && (NKK) == 0) CGV_LOAD_FRAGS_B(NB_, NBASE, NKK))    \
        MMA(1, 0, FA, FB) CGV_GAP(1, 0, FA[2], issue_q(Q0))                                                      \
        MMA(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], CGV_NOP_ACTION)                                                   \
        MMA(2, 1, FA, FB) CGV_GAP(2, 1, FA[3], issue_q(Q0 + 1))                                                  \
        MMA(3, 1, FA, FB) CGV_GAP(3, 1, FB[0], CGV_NOP_ACTION)                                                   \
        MMA(3, 0, FA, FB) CGV_GAP(3, 0, NA[0], CGV_NOP_ACTION)                                                   \
    }
    // The same k-step with every gap's action spelled out (the tile's LAST k-step carries the threshold set-up).
#define CGV_KSTEP_X(MMA, FA, FB, NA, G0, G1, G2, G3, G4, G5, G6, G7)                     \
    {                                                                                    \
        MMA(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], G0)                                       \
        MMA(0, 1, FA, FB) CGV_GAP(0, 1, FA[1], G1)                                       \
        MMA(1, 1, FA, FB) CGV_GAP(1, 1, FB[0], G2)                                       \
        MMA(1, 0, FA, FB) CGV_GAP(1, 0, FA[2], G3)                                       \
        MMA(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], G4)                                       \
        MMA(2, 1, FA, FB) CGV_GAP(2, 1, FA[3], G5)                                       \
        MMA(3, 1, FA, FB) CGV_GAP(3, 1, FB[0], G6)                                       \
        MMA(3, 0, FA, FB) CGV_GAP(3, 0, NA[0], G7)                                       \
    }
    // thresholds of the tile that ends here: block norm bounds from the side-data ring (safe to read once the
    // stage's counted wait + barrier in gap 0 have passed: KC >= 3, see issue_side), then 3 VALU per block
    f32x4_t mn4, mx4;
#define CGV_THR_LOAD(SEQ)                                                                             \
    {                                                                                                 \
        const float* st_ = stat_s + ((SEQ) & (NINV - 1)) * 16 + wm * MB;                              \
        mn4 = *(const f32x4_t*)st_;                                                                   \
        mx4 = *(const f32x4_t*)(st_ + 8);                                                             \
    }
#define CGV_THR(MBI, NBI, MN, MX)                                                                     \
    {                                                                                                 \
        float mul_ = tone[NBI] ? 1.0f : (tneg[NBI] ? (MX) : (MN));                                    \
        asm volatile("" : "+v"(mul_));                                                                \
        thr[MBI][NBI] = ta[NBI] * mul_;                                                               \
        asm volatile("" : "+v"(thr[MBI][NBI]));                                                       \
    }
#define CGV_STAGE_SYNC                                                                          \
    if (!(ABL & 16)) {                                                                          \
        if ((ABL & 768) == 256) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  /* lead - 1 stage (timing probe) */ \
        else if ((ABL & 768) == 512) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           \
        else if ((ABL & 768) == 768) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");           \
        else if (ABL & (1024 | 2048)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  /* 2 DMA per stage: the same 2-stage lead */ \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                   \
    }                                                                                           \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier()
    // Stage s holds k-steps (s,0) [fragments fa0/fb0] and (s,1) [fa1/fb1]. Iteration s runs the B phase =
    // k-step (s-1,1) with the stage barrier behind its first MFMA, then the A phase = k-step (s,0). DMA lead:
    // 3 stages (the 8 younger DMA instructions at the counted wait are stages s+1, s+2). No MFMA sits inside a
    // branch of a loop body (no accumulator phis for the register allocator to copy around).
#define CGV_A_PHASE(SB_) CGV_KSTEP(CGV_MMA, fa0, fb0, fa1, fb1, SB_, 1, 2, CGV_NOP_ACTION)
#define CGV_A_PHASE_Z(SB_) CGV_KSTEP(CGV_MMAZ, fa0, fb0, fa1, fb1, SB_, 1, 2, CGV_NOP_ACTION)
#define CGV_B_PHASE(SB_) CGV_KSTEP(CGV_MMA, fa1, fb1, fa0, fb0, SB_, 0, 0, CGV_STAGE_SYNC)
#define CGV_B_PHASE_LAST(SB_, SEQ)                                                                                  \
    CGV_KSTEP_X(CGV_MMA, fa1, fb1, fa0, CGV_STAGE_SYNC; if (LG == 0) CGV_LOAD_FRAGS(fa0, fb0, SB_, 0),              \
                if (LG == 1) CGV_LOAD_FRAGS(fa0, fb0, SB_, 0); if (LG == 12) CGV_LOAD_FRAGS_A(fa0, SB_, 0); CGV_THR_LOAD(SEQ), \
                if (LG == 2) CGV_LOAD_FRAGS(fa0, fb0, SB_, 0); if (LG == 12) CGV_LOAD_FRAGS_B(fb0, SB_, 0);         \
                CGV_THR(0, 0, mn4.x, mx4.x) CGV_THR(0, 1, mn4.x, mx4.x), issue_q(0); CGV_THR(1, 0, mn4.y, mx4.y),   \
                CGV_THR(1, 1, mn4.y, mx4.y) CGV_THR(2, 0, mn4.z, mx4.z), issue_q(1); CGV_THR(2, 1, mn4.z, mx4.z),   \
                CGV_THR(3, 0, mn4.w, mx4.w), CGV_THR(3, 1, mn4.w, mx4.w))
    // filter of block (MBI, NBI) of the tile that just ended (ftile; inverse norms finv), placed in the gap in
    // front of the zero-C MFMA that overwrites the block. The opening pin keeps the (movable) maxima behind everything
    // issued so far: the block's last MFMA is then >= 7 matrix instructions old, no hazard pad needed.
    uint32_t ftile = 0;
    const float* finv = invn_s;
    auto filt_block = [&](auto mb_c, auto nb_c) __attribute__((always_inline)) {
        constexpr int MBI = decltype(mb_c)::value, NBI = decltype(nb_c)::value;
        if (!(ABL & 1)) {
            asm volatile("" : "+v"(acc[MBI][NBI])::"memory");
            if (__builtin_expect(__ballot(block_max(acc[MBI][NBI]) > thr[MBI][NBI]) != 0ull, 0))  // cold, out of line
                block_hits<BM, BN>(a, acc[MBI][NBI], thr[MBI][NBI], tauv[NBI], invq[NBI], (uint32_t)(wm * WTM + MBI * 32),
                                   (uint32_t)(wn * WTN + NBI * 32 + (lane & 31)), ftile, lane, g, qt, cntq, finv,
                                   lad_on ? ladc_s : (const float4*)nullptr);
        }
    };
#define CGV_FILT(MBI, NBI) filt_block(IntC<MBI>{}, IntC<NBI>{});
#define CGV_A_PHASE_ZF(SB_)                                                                                         \
    {                                                                                                               \
        CGV_FILT(0, 0)                                                                                              \
        asm volatile("" : "+v"(fa0[0])::"memory");                                                                  \
        CGV_KSTEP_X(CGV_MMAZ, fa0, fb0, fa1, CGV_LOAD_FRAGS(fa1, fb1, SB_, 1); CGV_FILT(0, 1), CGV_FILT(1, 1),      \
                    CGV_FILT(1, 0), issue_q(2); CGV_FILT(2, 0), CGV_FILT(2, 1), issue_q(3); CGV_FILT(3, 1),         \
                    CGV_FILT(3, 0), CGV_NOP_ACTION)                                                                 \
    }
#define CGV_THR_ALL(SEQ)                                                                                            \
    {                                                                                                               \
        CGV_THR_LOAD(SEQ)                                                                                           \
        CGV_THR(0, 0, mn4.x, mx4.x) CGV_THR(0, 1, mn4.x, mx4.x) CGV_THR(1, 0, mn4.y, mx4.y) CGV_THR(1, 1, mn4.y, mx4.y) \
        CGV_THR(2, 0, mn4.z, mx4.z) CGV_THR(2, 1, mn4.z, mx4.z) CGV_THR(3, 0, mn4.w, mx4.w) CGV_THR(3, 1, mn4.w, mx4.w) \
    }

    // COARSE_EMIT_BOOT, the two peeled tile boundaries (BootSync above). stash_tile: the ended tile's accumulators -> global
    // memory, block by block (32 coalesced 16-byte stores per lane; plain stores: only this workgroup reads them back).
    __shared__ uint32_t boot_ok_s;
    // boundary tile 0 -> 1, first half: publish the tile's group maxima (the accumulators stay in their registers), ARRIVE
    auto boot_publish = [&](uint32_t tile0, const float* inv0) __attribute__((always_inline)) {
        if constexpr (MODE == 3) {
            const BootSync bs = boot_sync_of(a, qt);
            const uint32_t lg = (uint32_t)(wm * 2 + (lane >> 5));  // the lane's row group within the tile: 0..3
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float m[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    float mm = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t rl = (uint32_t)(wm * WTM + mb * 32 + (r & 3) + 8 * (r >> 2)) + 4u * (uint32_t)(lane >> 5);
                        const float sc = (a.metric == METRIC_DOT) ? acc[mb][nb][r] : acc[mb][nb][r] * inv0[rl];
                        if (tile0 * (uint32_t)BM + rl < a.n) mm = fmaxf(mm, sc);
                    }
                    m[mb] = (mm == -INFINITY) ? -INFINITY : mm * invq[nb];
                    asm volatile("" ::: "memory");   // block by block (no 128 inverse-norm reads in flight at once)
                }
                const uint32_t q = qt * (uint32_t)BN + (uint32_t)(wn * WTN + nb * 32 + (lane & 31));
                if (q < a.nq) {
                    float* dst = a.dump + (uint64_t)q * a.sample_ld + split * a.sample_vals;
                    auto put = [&](uint32_t i, float v) { __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                    if (a.sample_vals == 16u) {
                        put(lg * 4u, m[0]);
                        put(lg * 4u + 1u, m[1]);
                        put(lg * 4u + 2u, m[2]);
                        put(lg * 4u + 3u, m[3]);
                    } else if (a.sample_vals == 8u) {
                        put(lg * 2u, fmaxf(m[0], m[1]));
                        put(lg * 2u + 1u, fmaxf(m[2], m[3]));
                    } else {
                        put(lg, fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])));
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left (write-through: they are in memory)
            __builtin_amdgcn_s_barrier();
            if (tid == 0) boot_arrive(bs.arrive_a);
        }
    };
    // boundary tile 0 -> 1, second half: rendezvous A, thresholds, rendezvous B, every lane's thresholds
    auto boot_resolve = [&]() __attribute__((always_inline)) {
        if constexpr (MODE == 3) {
            const BootSync bs = boot_sync_of(a, qt);
            // ONE lane waits; the verdict reaches the other waves through LDS (every wave must act on the SAME verdict)
            if (tid == 0) boot_ok_s = boot_wait(bs.arrive_a, bs.degraded, a.nsplit) ? 1u : 0u;
            __builtin_amdgcn_s_barrier();
            const bool ok_a = boot_ok_s != 0u;
            // thresholds of this workgroup's share of the query tile: queries split + nsplit * (wave + 8 j), one wave each
            if (ok_a)
                boot_thresholds(a.dump, a.tau_out, a.sample_ld, a.kprime, qt * (uint32_t)BN + split + (uint32_t)wave * a.nsplit,
                                8u * a.nsplit, (qt + 1u) * (uint32_t)BN, a.nq, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // thresholds (and the stash) have left
            __builtin_amdgcn_s_barrier();   // (also: every wave has read boot_ok_s before it is rewritten)
            // rendezvous B counts PUBLISHED shares only: a workgroup that did not compute its thresholds never arrives, so a full
            // count means all 256 thresholds of the tile exist; the others leave through `degraded` (set before anyone skips)
            if (tid == 0) {
                bool ok = ok_a;
                if (ok) {
                    boot_arrive(bs.arrive_b);
                    ok = boot_wait(bs.arrive_b, bs.degraded, a.nsplit);
                }
                boot_ok_s = ok ? 1u : 0u;
            }
            __builtin_amdgcn_s_barrier();
            const bool good = boot_ok_s != 0u;
            // every lane's thresholds (its NB queries), or - degraded - no emission and the exact scan for the tile's queries
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint32_t q = qt * (uint32_t)BN + (uint32_t)(wn * WTN + nb * 32 + (lane & 31));
                const bool valid = q < a.nq;
                float tau = INFINITY;
                if (valid && good) tau = __hip_atomic_load(a.tau_out + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (valid && !good && wm == 0 && lane < 32) a.overflow[q] = 1u;
                const float iq = invq[nb];
                tauv[nb] = tau;
                tq[nb] = (tau == -INFINITY) ? -INFINITY : (iq == 0.0f ? INFINITY : tau / iq);
                tone[nb] = (a.metric == METRIC_DOT) || !(fabsf(tq[nb]) < INFINITY);
                tneg[nb] = tq[nb] < 0.0f;
                ta[nb] = tone[nb] ? tq[nb] : (tneg[nb] ? tq[nb] * (1.0f + 3.8147e-6f) : tq[nb] * (1.0f - 3.8147e-6f));
            }
        }
    };

    // ---- prologue: three stages in flight --------------------------------------------
    if (total == 0) {  // uniform: nothing to stream for this workgroup
        for (int i = tid; i < BN; i += NT) a.cand_cnt[(uint64_t)g * BN + i] = 0;
        if (MODE == 3 && tid == 0) {  // (the host never launches such a workgroup in this mode; the group must not wait for it)
            const BootSync bs = boot_sync_of(a, qt);
            __hip_atomic_fetch_add(bs.arrive_a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bs.arrive_b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (lad_on && wave == 3) {   // the queries' {tau0, delta, 1 / delta}: constant for the launch, landed with the first stage
        const char* src = (const char*)(a.ladc + (uint64_t)qt * BN) + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(src + i * 1024, (char*)ladc_s + i * 1024 + lane * 16);
    }
    issue_side(t_first, 0);
#pragma unroll 1
    for (int i = 0; i < NSTAGE - 1; ++i) {
        if (SI >= 2) si_slot = (uint32_t)(i * STAGE);
        issue_q(0);
        issue_q(1);
        issue_q(2);
        issue_q(3);
    }
    if (ABL & (1024 | 2048)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // stage 0 (and the side data before it) landed; stages 1, 2 may be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my cntq zero-stores done
    __builtin_amdgcn_s_barrier();
    CGV_LOAD_FRAGS(fa0, fb0, smem, 0);
    if (ABL & (1024 | 4096)) {  // timing probe "the query operand for free" (2048: only its DMA goes, 4096: only its fragment reads): B fragments once, straight from the query tile's first chunk
                       // in global memory (real data); no B DMA, no B fragment reads in the loop (results are wrong)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            fb0[i] = *(const frag*)(bbase + (boff - (uint32_t)A_BYTES) + i * 2048 + xo[0]);
            fb1[i] = *(const frag*)(bbase + (boff - (uint32_t)A_BYTES) + i * 2048 + xo[1]);
        }
    }
    if (ABL & 128) {  // both fragment sets once, from the first stage (real data), never again
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            fa0[i] = *(const frag*)(smem + aoff + i * 2048 + xo[0]);
            fa1[i] = *(const frag*)(smem + aoff + i * 2048 + xo[1]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            fb0[i] = *(const frag*)(smem + boff + i * 2048 + xo[0]);
            fb1[i] = *(const frag*)(smem + boff + i * 2048 + xo[1]);
        }
    }

#define CGV_EPILOGUE(TILE, SEQ)                                                                                    \
    if (!(ABL & 1)) {                                                                                              \
        if (EMIT && EPI != 0)                                                                                      \
            tile_filter_emit<BM, BN, WTM, WTN, MB, NB>(a, acc, thr, TILE, wm, wn, lane, g, qt, tauv, invq, cntq,     \
                                                       invn_s + ((SEQ) & (NINV - 1)) * 256,                        \
                                                       lad_on ? ladc_s : (const float4*)nullptr);                  \
        else                                                                                                       \
            tile_epilogue<BM, BN, WTM, WTN, MB, NB, (MODE == 3 ? 0 : MODE)>(a, acc, TILE, wm, wn, lane, g, qt, tq, tauv, invq, cntq,  \
                                                          invn_s + ((SEQ) & (NINV - 1)) * 256,                     \
                                                          stat_s + ((SEQ) & (NINV - 1)) * 16, a.j0 + jlo + (SEQ), t2); \
    }

    // Tile-structured: [first stage of a tile: zero-C MFMAs] then KC-1 ordinary stages; at a tile
    // boundary the iteration is B phase (last k-step of the previous tile), its epilogue, zero-C A phase.
    const uint32_t ntl = jhi - jlo;
    uint32_t ct = t_first, s = 1;
    if (SI >= 2) si_slot = (uint32_t)(3 * STAGE);
    issue_q(0);  // stage 3 -> slot 3 (never used so far)
    issue_q(1);
    CGV_A_PHASE_Z(smem);
    // SI >= 2 (KC % 4 == 0): every tile starts in ring slot 0, so the stage loop unrolled by the ring size has its LDS
    // addresses (fragment reads: lane base + immediate; DMA: wave base + constant) fixed per unrolled iteration
#define CGV_ITER_AT(SLOT)                                                         \
    {                                                                             \
        const char* sbc = smem + (SLOT) * STAGE;                                  \
        si_slot = (uint32_t)((((SLOT) + 3) & (NSTAGE - 1)) * STAGE);              \
        CGV_B_PHASE(sbc);                                                         \
        CGV_A_PHASE(sbc);                                                         \
    }
#define CGV_TILE_REST_U4                                                          \
    {                                                                             \
        const uint32_t ng = KC >> 2;                                              \
        if (ng == 1) issue_switch_tile();                                         \
        CGV_ITER_AT(1) CGV_ITER_AT(2) CGV_ITER_AT(3)                              \
        _Pragma("unroll 1") for (uint32_t gi = 1; gi < ng; ++gi) {                \
            CGV_ITER_AT(0)                                                        \
            if (gi + 1 == ng) issue_switch_tile();                                \
            CGV_ITER_AT(1) CGV_ITER_AT(2) CGV_ITER_AT(3)                          \
        }                                                                         \
    }
    if constexpr (SI >= 2) {
        CGV_TILE_REST_U4
    } else {
#pragma unroll 1
        for (uint32_t kc = 1; kc < KC; ++kc, ++s) {  // rest of the first tile
            const char* sb = smem + (s & (NSTAGE - 1)) * STAGE;
            if (SI != 0 && kc + 3 == KC) issue_switch_tile();  // this iteration issues stage kc + 3 = the next tile's first
            CGV_B_PHASE(sb);
            CGV_A_PHASE(sb);
        }
    }
    // One tile boundary + the rest of the tile that starts there. BOOT (COARSE_EMIT_BOOT, first boundary only): the
    // rendezvous block sits between the ended tile's last MFMA and its filters. It is a PEELED copy of the loop body - inside
    // the loop the block's register pressure (it spills around the rendezvous, once per launch) reached the steady-state path.
    auto tile_iter = [&](uint32_t tl, auto boot_c) __attribute__((always_inline)) {
        constexpr int BOOT = decltype(boot_c)::value;   // 0 ordinary; 1 / 2: the peeled first two boundaries of COARSE_EMIT_BOOT
        {
            const char* sb = (SI >= 2) ? smem : smem + (s & (NSTAGE - 1)) * STAGE;
            if (SI >= 2) si_slot = (uint32_t)(3 * STAGE);
            if constexpr (BOOT != 0) {
                // the ended tile's last k-step WITHOUT the early reads of the next tile's first fragments: 24 fewer live
                // registers across the boot block (with them the kernel spilled, and the allocator's choice - the queries'
                // inverse norms - was reloaded from scratch, behind a vmcnt(0), in every hit of the whole walk)
                CGV_KSTEP_X(CGV_MMA, fa1, fb1, fa1, CGV_STAGE_SYNC, CGV_NOP_ACTION, CGV_NOP_ACTION, issue_q(0), CGV_NOP_ACTION,
                            issue_q(1), CGV_NOP_ACTION, CGV_NOP_ACTION)
            } else if (!EMIT || EPI == 0) {
                CGV_B_PHASE(sb);
            } else {
                CGV_B_PHASE_LAST(sb, tl - 1);
            }
            if (wave == 0) pace_step(pace, tl + 1, lane);
            const uint32_t nt = next_tile(ct);
            side_wait();
            if (EMIT && EPI != 0 && KC < 3) CGV_THR_ALL(tl - 1);  // short tiles: the side data is only certain to be there now
            if constexpr (BOOT == 1) {   // tile 0 ends: maxima out, rendezvous, thresholds; its accumulators never leave
                boot_publish(a.T1 + ct, invn_s);
                boot_resolve();
                CGV_THR_ALL(0);
                CGV_LOAD_FRAGS(fa0, fb0, sb, 0);   // the starting tile's first fragments (not read early, see above)
            }
            if (EMIT && EPI != 0) {
                // ladder words DMA'd at the previous boundary (with the side data of the tile that just ended): landed for the
                // same reason its inverse norms have. The thresholds formed from now on - the next tile's - use the raised values;
                // the filters below still run on this tile's (lower, valid) ones.
                if (lad_on) ladder_raise((tl - 1) & 1u);
                ftile = a.T1 + ct;
                finv = invn_s + ((tl - 1) & (NINV - 1)) * 256;
                issue_side(nt, tl);  // the tile that starts here (another slot of the side-data ring)
                ct = nt;
                CGV_A_PHASE_ZF(sb);
            } else {
                CGV_EPILOGUE(a.T1 + ct, tl - 1);
                issue_side(nt, tl);  // the tile that starts here
                ct = nt;
                CGV_A_PHASE_Z(sb);
            }
            ++s;
        }
        if constexpr (SI >= 2) {
            CGV_TILE_REST_U4
        } else {
#pragma unroll 1
            for (uint32_t kc = 1; kc < KC; ++kc, ++s) {
                const char* sb = smem + (s & (NSTAGE - 1)) * STAGE;
                if (SI != 0 && kc + 3 == KC) issue_switch_tile();
                CGV_B_PHASE(sb);
                CGV_A_PHASE(sb);
            }
        }
    };
    // COARSE_EMIT_BOOT: every workgroup walks >= 2 tiles (the launcher refuses anything else), so the peeled boundary is
    // unconditional - no branch around MFMAs, no accumulator phis
    if constexpr (MODE == 3) tile_iter(1u, IntC<1>{});
#pragma unroll 1
    for (uint32_t tl = (MODE == 3 ? 2u : 1u); tl < ntl; ++tl) tile_iter(tl, IntC<0>{});
#undef CGV_TILE_REST_U4
#undef CGV_ITER_AT
    // tail: second k-step of the last stage, then the last tile's epilogue
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = Mfma<DT>::mma(fa1[mb], fb1[nb], acc[mb][nb]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy DMA tail (and a short tile's side data)
    __builtin_amdgcn_s_barrier();
    if (EMIT && EPI != 0) CGV_THR_ALL(ntl - 1);
    CGV_EPILOGUE(a.T1 + ct, ntl - 1);
#undef CGV_A_PHASE_Z
#undef CGV_B_PHASE
#undef CGV_B_PHASE_LAST
#undef CGV_A_PHASE_ZF
#undef CGV_FILT
#undef CGV_THR_ALL
#undef CGV_THR
#undef CGV_THR_LOAD
#undef CGV_KSTEP_X
#undef CGV_A_PHASE
#undef CGV_EPILOGUE
#undef CGV_STAGE_SYNC
#undef CGV_KSTEP
#undef CGV_NOP_ACTION
#undef CGV_GAP
#undef CGV_MMAZ
#undef CGV_MMA
#undef CGV_LOAD_FRAGS
#undef CGV_LDA
#undef CGV_LDB
#undef CGV_BDMA
#undef CGV_BDMA_A

    if (MODE == 4) top2_flush<BN, NB>(a, t2, invq, wm, wn, lane, g, qt, cntq);
    __syncthreads();
    if (tid == 0) pace_done(pace);
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

// ---------------------------------------------------------------------------------
// boot_kernel: dense coarse scores of the n_boot BOOT rows (32-row groups spread over the corpus, boot_row())
// against all queries, straight from global memory (no LDS, no atomics): one wave = one 64 x 64 output
// block (2 x 2 MFMA blocks), K loop unrolled so many 16-B loads are in flight. It bootstraps
// the per-query threshold tau (top-k' of n_boot scores) in ~30 us instead of running the
// big-tile kernel on 4 workgroups. dense: [nq][n_boot], column = LOGICAL boot row; rows beyond the
// corpus (the ragged last group can be a boot group) get -inf.
// ---------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void boot_kernel(const char* __restrict__ rows,
                                                  const char* __restrict__ qrows,
                                                  const float* __restrict__ invn_c,
                                                  const float* __restrict__ invn_q, uint32_t n_boot,
                                                  uint32_t nq, uint32_t ld, int metric,
                                                  float* __restrict__ dense, BootMap bmap, uint32_t n,
                                                  const int8_t* __restrict__ rexp_c = nullptr,
                                                  const int8_t* __restrict__ rexp_q = nullptr) {
    typedef typename Mfma<DT>::frag frag;
    const int lane = threadIdx.x & 63;
    const uint32_t nrb = (n_boot + 63) / 64;
    const uint32_t rb = blockIdx.x % nrb, qb = blockIdx.x / nrb;
    const int hi = lane >> 5;
    const uint32_t pbase[2] = {boot_row(bmap, rb * 64), boot_row(bmap, rb * 64 + 32)};  // corpus rows of the two 32-row groups
    const char* ap[2];
    const char* bp[2];
    uint32_t akey[2], bkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint32_t r = pbase[i] + (lane & 31);
        r = r < n ? r : n - 1;
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
    // latency-bound (one wave per 64x64 block, operands straight from L2/HBM): eight k-steps of loads
    // (32 x 16 B per lane, 128 VGPRs) are issued before their 32 MFMAs
    constexpr int BU = 8;
    uint32_t k = 0;
    for (; k + BU <= ks; k += BU) {
        frag fa0[BU], fa1[BU], fb0[BU], fb1[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const uint32_t pc = 2 * (k + u) + hi;  // this lane's 16-byte piece of k-step k+u
            fa0[u] = *(const frag*)(ap[0] + blocked_piece_off(pc, akey[0]));
            fa1[u] = *(const frag*)(ap[1] + blocked_piece_off(pc, akey[1]));
            fb0[u] = *(const frag*)(bp[0] + blocked_piece_off(pc, bkey[0]));
            fb1[u] = *(const frag*)(bp[1] + blocked_piece_off(pc, bkey[1]));
        }
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            acc[0][0] = Mfma<DT>::mma(fa0[u], fb0[u], acc[0][0]);
            acc[0][1] = Mfma<DT>::mma(fa0[u], fb1[u], acc[0][1]);
            acc[1][0] = Mfma<DT>::mma(fa1[u], fb0[u], acc[1][0]);
            acc[1][1] = Mfma<DT>::mma(fa1[u], fb1[u], acc[1][1]);
        }
    }
    for (; k < ks; ++k) {
        const uint32_t pc = 2 * k + hi;
        frag a0 = *(const frag*)(ap[0] + blocked_piece_off(pc, akey[0])), a1 = *(const frag*)(ap[1] + blocked_piece_off(pc, akey[1]));
        frag b0 = *(const frag*)(bp[0] + blocked_piece_off(pc, bkey[0])), b1 = *(const frag*)(bp[1] + blocked_piece_off(pc, bkey[1]));
        acc[0][0] = Mfma<DT>::mma(a0, b0, acc[0][0]);
        acc[0][1] = Mfma<DT>::mma(a0, b1, acc[0][1]);
        acc[1][0] = Mfma<DT>::mma(a1, b0, acc[1][0]);
        acc[1][1] = Mfma<DT>::mma(a1, b1, acc[1][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t q = qb * 64 + j * 32 + (lane & 31);
        if (q >= nq) continue;
        // fp8: this kernel multiplies the e4m3 CODES (v_mfma_f32_32x32x16_fp8_fp8 has no scale operands) while the
        // inverse norms are those of the de-scaled rows: 1 / |codes| = invn * 2^-e, folded in here (exact)
        float iq = (metric == METRIC_DOT) ? 1.0f : invn_q[q];
        if (DT == DT_FP8 && rexp_q) iq = ldexpf(iq, -(int)rexp_q[q]);
        // C layout: registers 4g..4g+3 of a block are 4 CONSECUTIVE corpus rows -> one 16-byte store each
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint32_t lrow = rb * 64 + i * 32 + 8 * g4 + 4 * (lane >> 5);  // dense column
                const uint32_t row = pbase[i] + 8 * g4 + 4 * (lane >> 5);           // corpus row
                float* dst = dense + (uint64_t)q * n_boot + lrow;
                if (lrow + 3 < n_boot && row + 3 < n && (n_boot & 3u) == 0) {
                    float4 v = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2],
                                           acc[i][j][4 * g4 + 3]);
                    if (metric != METRIC_DOT) {
                        float4 ic = *(const float4*)(invn_c + row);
                        if (DT == DT_FP8 && rexp_c) {
                            const int ec = *(const int*)(rexp_c + row);  // 4 signed bytes
                            ic.x = ldexpf(ic.x, -((ec << 24) >> 24));
                            ic.y = ldexpf(ic.y, -((ec << 16) >> 24));
                            ic.z = ldexpf(ic.z, -((ec << 8) >> 24));
                            ic.w = ldexpf(ic.w, -(ec >> 24));
                        }
                        v.x = v.x * ic.x * iq;
                        v.y = v.y * ic.y * iq;
                        v.z = v.z * ic.z * iq;
                        v.w = v.w * ic.w * iq;
                    }
                    *(float4*)dst = v;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (lrow + t < n_boot) {
                            if (row + t < n) {
                                const float v = acc[i][j][4 * g4 + t];
                                float ic = invn_c[row + t];
                                if (DT == DT_FP8 && rexp_c) ic = ldexpf(ic, -(int)rexp_c[row + t]);
                                dst[t] = (metric == METRIC_DOT) ? v : v * ic * iq;
                            } else {
                                dst[t] = -INFINITY;
                            }
                        }
                }
            }
    }
}

}  // namespace cgv
