// kernels_select.h — candidate selection (top-k' by coarse score), the exact
// re-score + final ordering, and the merge of per-shard partial top-k lists.
// All LDS-resident bitonic sorts over 64-bit (score, ~row) keys: integer work,
// HBM traffic is a few KB per query.
#pragma once
#include "common.h"
#include "kernels_coarse.h"

namespace cgv {

constexpr uint32_t SELECT_LDS_KEYS = 8192;  // 64 KiB of keys per workgroup

struct SelectArgs {
    const uint2* cand;         // [W][BN][CAND_CAPS]
    const uint32_t* cand_cnt;  // [W][BN]
    const float* dense;        // optional: [nq][n_dense] dense coarse scores (boot stage) instead of cand
    uint64_t* best;            // [nq][kprime] keys, sorted desc
    uint32_t* nbest;           // [nq]
    float* tau;                // [nq]
    uint32_t* overflow;        // [nq]
    uint32_t nq, nqt, nsplit, bn, kprime, n_dense, lds_keys;
    uint32_t tau_only;         // dense scores of a SAMPLE of the corpus: publish tau, keep no candidates
    uint32_t* floor_ord;       // COARSE_TOP2: [nq] f2ord of the best coarse score left out of the candidate lists (0: none), or NULL;
                               // final_kernel reads its query's word and clears it for the next search (no memset launch per call)
    uint32_t floor_with_tau;   // != 0: the floor comes from an EMITTING SAMPLE (tile_epilogue, SAMPLE mode) and bounds only the rows of
                               // the sampled tiles; the launches behind it had thresholds (tau[] is meaningful, unlike COARSE_TOP2)
    uint64_t* trace;           // diagnostics (CGV_TRACE=1): [nq][8] wall-clock stamps of the kernel's phases, or NULL
};

// diagnostics: phase stamp i of query q (s_memrealtime, 100 MHz), thread 0 only; free when trace == NULL
__device__ inline void phase_stamp(uint64_t* trace, uint32_t q, int i, int tid) {
    if (trace && tid == 0) trace[(uint64_t)q * 8 + i] = wall_clock64();
}

// ---- shared pieces of the select / final kernels ------------------------------------

// Gather the keys of query (qt, ql) into LDS: best[q][0..nb) followed by the nsplit candidate
// sub-lists of the last coarse launch (or the dense boot scores). Returns M (clamped to
// lds_keys; *trunc set when clamped). Memory-latency bound, so it is written as two
// round trips: (1) nbest + all sub-list counts, (2) every candidate load issued before the
// first LDS store (4 independent 8-byte loads per thread per batch).
__device__ inline uint32_t gather_keys(const SelectArgs& a, uint32_t q, uint64_t* keys, uint32_t* pre, int tid,
                                       bool* trunc) {
    const uint32_t qt = q / a.bn, ql = q % a.bn;
    const uint32_t nb = a.nbest[q];
    uint32_t M;
    if (a.dense) {
        M = nb + a.n_dense;
        if (M > a.lds_keys) {
            *trunc = true;
            M = a.lds_keys;
        }
        for (uint32_t e = tid; e < M; e += 256)
            keys[e] = (e < nb) ? a.best[(uint64_t)q * a.kprime + e]
                               : make_key(a.dense[(uint64_t)q * a.n_dense + (e - nb)], e - nb);
        __syncthreads();
        return M;
    }
    if (tid < 64) {  // exclusive prefix of the sub-list counts: pre[s] = nb + sum_{s' < s} cnt[s']
        uint32_t cnt[4] = {0, 0, 0, 0};  // nsplit <= 256
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t sidx = tid * 4 + i;
            if (sidx < a.nsplit) cnt[i] = a.cand_cnt[(uint64_t)(qt + sidx * a.nqt) * a.bn + ql];
        }
        const uint32_t loc = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        uint32_t inc = loc;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off, 64);
            if (tid >= off) inc += v;
        }
        uint32_t run = nb + inc - loc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t sidx = tid * 4 + i;
            if (sidx < a.nsplit) pre[sidx] = run;
            run += cnt[i];
        }
        if (tid == 63) pre[a.nsplit] = nb + inc;
    }
    __syncthreads();
    M = pre[a.nsplit];
    if (M > a.lds_keys) {  // cannot happen with sane thresholds; flag for the exact path
        *trunc = true;
        M = a.lds_keys;
    }
    for (uint32_t e = tid; e < nb && e < M; e += 256) keys[e] = a.best[(uint64_t)q * a.kprime + e];
    uint32_t tps = 256;  // threads per sub-list (power of two)
    while (tps > 1 && tps * a.nsplit > 256) tps >>= 1;
    const uint32_t per_round = 256 / tps, sub = tid % tps;
    for (uint32_t sidx = tid / tps; sidx < a.nsplit; sidx += per_round) {
        const uint32_t lo = pre[sidx], hi = pre[sidx + 1] < M ? pre[sidx + 1] : M;
        const uint2* src = a.cand + ((uint64_t)(qt + sidx * a.nqt) * a.bn + ql) * CAND_CAPS;
        for (uint32_t e0 = lo + sub; e0 < hi; e0 += 4 * tps) {
            uint2 c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t e = e0 + i * tps;
                c[i] = (e < hi) ? src[e - lo] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t e = e0 + i * tps;
                if (e < hi) keys[e] = make_key(__uint_as_float(c[i].x), c[i].y);
            }
        }
    }
    __syncthreads();
    return M;
}

__device__ inline void cmpx_desc(uint64_t& x, uint64_t& y) {  // compare-exchange: x >= y afterwards
    const uint64_t hi = x > y ? x : y, lo = x > y ? y : x;
    x = hi;
    y = lo;
}

// 8-input sorting network (19 compare-exchanges), descending
__device__ inline void sort8_desc(uint64_t* r) {
    cmpx_desc(r[0], r[1]); cmpx_desc(r[2], r[3]); cmpx_desc(r[4], r[5]); cmpx_desc(r[6], r[7]);
    cmpx_desc(r[0], r[2]); cmpx_desc(r[1], r[3]); cmpx_desc(r[4], r[6]); cmpx_desc(r[5], r[7]);
    cmpx_desc(r[1], r[2]); cmpx_desc(r[5], r[6]); cmpx_desc(r[0], r[4]); cmpx_desc(r[3], r[7]);
    cmpx_desc(r[1], r[5]); cmpx_desc(r[2], r[6]);
    cmpx_desc(r[1], r[4]); cmpx_desc(r[3], r[6]);
    cmpx_desc(r[2], r[4]); cmpx_desc(r[3], r[5]);
    cmpx_desc(r[3], r[4]);
}

// Register-resident extraction: every lane keeps its NPL keys SORTED IN REGISTERS; a round is one
// wave-wide max (DPP) + a register shift in the owner lane. keys are unique, 0 = empty.
template <int NPL, class LOAD>
__device__ inline void extract_regs(LOAD load, uint32_t M, uint32_t keep, uint64_t* part, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    uint64_t r[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const uint32_t e = (uint32_t)tid + 256u * i;
        r[i] = e < M ? load(e) : 0ull;
    }
    sort8_desc(r);
    if (NPL == 16) {  // second run, then a bitonic merge of the two sorted runs
        sort8_desc(r + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t t = r[8 + i];
            r[8 + i] = r[15 - i];
            r[15 - i] = t;
        }
#pragma unroll
        for (int stride = 8; stride > 0; stride >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & stride) == 0) cmpx_desc(r[i], r[i + stride]);
    }
    for (uint32_t rd = 0; rd < keep; ++rd) {
        const uint64_t w = wave_max_u64(r[0]);
        if (w != 0ull && r[0] == w) {  // exactly one owner
#pragma unroll
            for (int i = 0; i < NPL - 1; ++i) r[i] = r[i + 1];
            r[NPL - 1] = 0ull;
        }
        if (lane == 0) part[wv * 64 + rd] = w;
    }
}

// 4-way merge of the four sorted per-wave lists part[4][64] -> outk[0..keep): every thread owns one entry (list tid / 64,
// position tid % 64) and finds its rank = its position + the number of larger entries in the other three lists (a binary search
// each: keys are unique, 0 = empty, the lists are sorted descending with the empties last) - ~20 dependent LDS reads per thread,
// all in parallel, instead of `keep` serial rounds of four reads by one lane (round 5; the single-lane form took ~4 us of a
// single-query call's final kernel).
__device__ inline void merge4(const uint64_t* part, uint32_t keep, uint64_t* outk, int tid) {
    __syncthreads();
    const uint32_t w = (uint32_t)tid >> 6, i = (uint32_t)tid & 63u;
    const uint64_t v = i < keep ? part[w * 64u + i] : 0ull;
    uint32_t rank = i;
    if (v != 0ull) {
#pragma unroll
        for (uint32_t d = 1; d < 4; ++d) {
            const uint64_t* lst = part + ((w + d) & 3u) * 64u;
            uint32_t lo = 0, hi = keep;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (lst[mid] > v) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
    }
    if ((uint32_t)tid < keep) outk[tid] = 0ull;
    __syncthreads();
    if (v != 0ull && rank < keep) outk[rank] = v;
    __syncthreads();
}

// Top-`keep` (keep <= 64) of keys[0..M) -> outk[0..keep) sorted descending. 256 threads.
// M <= 4096: register-resident extraction (8 or 16 keys per lane); larger M: lane-private LDS
// rescans. The four per-wave lists (already sorted) are merged by one lane.
__device__ inline void extract_topk(uint64_t* keys, uint32_t M, uint32_t keep, uint64_t* part /*[4*64]*/,
                                    uint64_t* outk /*[64]*/, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    auto from_lds = [&](uint32_t e) { return keys[e]; };
    if (M <= 512) {
        // the common case (a good threshold leaves ~100 keys): ONE wave, 8 keys per lane, its sorted output is the
        // result - no per-wave lists, no merge (measured r03a: 9.4 us of the final kernel's 40 us per workgroup went
        // to the four-wave extraction + single-lane merge)
        if (wv == 0) {
            uint64_t r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t e = (uint32_t)lane + 64u * i;
                r[i] = e < M ? keys[e] : 0ull;
            }
            sort8_desc(r);
            for (uint32_t rd = 0; rd < keep; ++rd) {
                const uint64_t w = wave_max_u64(r[0]);
                if (w != 0ull && r[0] == w) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) r[i] = r[i + 1];
                    r[7] = 0ull;
                }
                if (lane == 0) outk[rd] = w;
            }
        }
        __syncthreads();
        return;
    }
    if (M <= 2048) {
        extract_regs<8>(from_lds, M, keep, part, tid);
    } else if (M <= 4096) {
        extract_regs<16>(from_lds, M, keep, part, tid);
    } else {
        auto lane_max = [&](uint32_t& at) {
            uint64_t mx = 0ull;
            at = 0xFFFFFFFFu;
            for (uint32_t e = tid; e < M; e += 256) {
                const uint64_t v = keys[e];
                if (v > mx) {
                    mx = v;
                    at = e;
                }
            }
            return mx;
        };
        uint32_t my_at;
        uint64_t my_mx = lane_max(my_at);
        for (uint32_t rd = 0; rd < keep; ++rd) {
            const uint64_t w = wave_max_u64(my_mx);
            if (w != 0ull && my_mx == w) {
                keys[my_at] = 0ull;
                my_mx = lane_max(my_at);
            }
            if (lane == 0) part[wv * 64 + rd] = w;
        }
    }
    merge4(part, keep, outk, tid);
}

// The exact scan's chunk reduction for K <= 64 (kernels_exact.h: topk_chunk_kernel sorts all 4096 keys of a chunk with an
// LDS bitonic network - 78 barrier rounds, ~25 us - to keep 16 of them): register-resident extraction instead (every lane
// sorts its <= 16 keys in registers, K rounds of wave maximum, a 4-way merge) - the same keys, the same total order (keys are
// unique), ~3x shorter on the latency-bound single-query path (BASELINE config 1: 10k x 384 f32, one query per call).
// Level 0: chunk of f32 scores -> top-K keys; level > 0: chunk of keys -> top-K keys. grid = (nchunks, nql), out[qi][chunk][K].
static __global__ __launch_bounds__(256) void topk_chunk_small_kernel(const float* __restrict__ scores, const uint64_t* __restrict__ in_keys,
                                                               uint32_t M, uint32_t K, uint64_t* __restrict__ out,
                                                               uint32_t* __restrict__ nan_flag) {
    __shared__ uint64_t part[4 * 64];
    __shared__ uint64_t outk[64];
    const int tid = threadIdx.x;
    const uint32_t chunk = blockIdx.x, qi = blockIdx.y, nchunks = gridDim.x;
    const uint64_t base = (uint64_t)chunk * 4096u;
    const uint32_t cnt = (uint32_t)((M - base) < 4096u ? (M - base) : 4096u);
    const uint32_t keep = K < cnt ? K : cnt;
    auto load = [&](uint32_t e) -> uint64_t {
        const uint64_t ge = base + e;
        if (scores) {
            const float sc = scores[(uint64_t)qi * M + ge];
            if (sc != sc) *nan_flag = 1u;
            return make_key(sc, (uint32_t)ge);
        }
        return in_keys[(uint64_t)qi * M + ge];
    };
    if (cnt <= 2048u)
        extract_regs<8>(load, cnt, keep, part, tid);
    else
        extract_regs<16>(load, cnt, keep, part, tid);
    merge4(part, keep, outk, tid);
    for (uint32_t j = tid; j < K; j += 256) out[((uint64_t)qi * nchunks + chunk) * K + j] = j < keep ? outk[j] : 0ull;
}

// One workgroup (256 threads) per query: merge best[q] with the nsplit candidate
// sub-lists written by the last coarse launch (or with the dense boot scores), keep the
// top-k', publish tau[q] = the k'-th best coarse score seen so far (a valid lower bound of
// the final k'-th best). k' <= 64: register-resident extraction; larger k': LDS bitonic sort.
static __global__ __launch_bounds__(256) void select_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = (uint64_t*)smem;                            // [lds_keys]
    uint32_t* pre = (uint32_t*)(smem + (size_t)a.lds_keys * 8);  // [nsplit + 1]
    __shared__ uint64_t part[4 * 64];
    __shared__ uint64_t outk[64];
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    bool trunc = false;
    if (a.dense && a.lds_keys == 0) {
        // boot stage (nothing selected yet, n_dense <= 4096, k' <= 64): keys straight from the dense
        // scores into registers - no LDS staging, so 8 workgroups fit a CU instead of 2
        const uint32_t M0 = a.n_dense, keep0 = M0 < a.kprime ? M0 : a.kprime;
        const float* d = a.dense + (uint64_t)q * a.n_dense;
        auto from_dense = [&](uint32_t e) { return make_key(d[e], e); };
        if (M0 <= 2048)
            extract_regs<8>(from_dense, M0, keep0, part, tid);
        else
            extract_regs<16>(from_dense, M0, keep0, part, tid);
        merge4(part, keep0, outk, tid);
        for (uint32_t i = tid; i < keep0; i += 256) a.best[(uint64_t)q * a.kprime + i] = outk[i];
        if (tid == 0) {
            a.nbest[q] = a.tau_only ? 0u : keep0;
            a.tau[q] = (M0 >= a.kprime) ? key_score(outk[a.kprime - 1]) : -INFINITY;
        }
        return;
    }
    phase_stamp(a.trace, q, 0, tid);
    const uint32_t M = gather_keys(a, q, keys, pre, tid, &trunc);
    phase_stamp(a.trace, q, 1, tid);
    const uint32_t keep = M < a.kprime ? M : a.kprime;
    if (a.kprime <= 64) {
        extract_topk(keys, M, keep, part, outk, tid);
        phase_stamp(a.trace, q, 2, tid);
        for (uint32_t i = tid; i < keep; i += 256) a.best[(uint64_t)q * a.kprime + i] = outk[i];
        if (tid == 0) {
            a.nbest[q] = a.tau_only ? 0u : keep;
            // fewer than k' keys: nothing was cut HERE, the threshold the launches used so far stays (candidate
            // stages; the dense boot stage starts from nothing: -inf)
            if (M >= a.kprime) a.tau[q] = key_score(outk[a.kprime - 1]);
            else if (a.dense) a.tau[q] = -INFINITY;
            if (trunc) a.overflow[q] = 1u;
        }
    } else {
        const uint32_t P = next_pow2(M < 2 ? 2 : M);
        for (uint32_t i = M + tid; i < P; i += 256) keys[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc<256>(keys, P, tid);
        for (uint32_t i = tid; i < keep; i += 256) a.best[(uint64_t)q * a.kprime + i] = keys[i];
        if (tid == 0) {
            a.nbest[q] = a.tau_only ? 0u : keep;
            if (M >= a.kprime) a.tau[q] = key_score(keys[a.kprime - 1]);
            else if (a.dense) a.tau[q] = -INFINITY;
            if (trunc) a.overflow[q] = 1u;
        }
    }
}

// ---- first threshold from the sample launch (coarse kernels, COARSE_SAMPLE) -----------------------
// dense[q][0..M): maxima of DISJOINT groups of 16 corpus rows (one per lane and 32 x 32 block of the sampled
// tiles; -inf for groups without a valid row). tau[q] = the k'-th largest of them: at least k' distinct rows of
// the corpus score that much or more, so it is a valid lower bound of the final k'-th best coarse score, and a tight
// one - with M / 16 >> k' groups the k' best rows of the sample almost surely sit in k' different groups.
// ONE WAVE per query, values only (32-bit ordered keys, no row ids): <= 16 keys per lane sorted in registers, then
// k' rounds of wave maximum + pop. Also clears nbest / overflow of the query (the launches that follow append).
// Threshold ladder (kernels_coarse.h): also the query's {tau0, delta} and a cleared counter word. delta = lad_scale x (value at
// rank hi_rank of the sample's maxima - tau0): the sample's own spread between two of its order statistics, scaled by the host
// to a quarter of 1.25 x the distance at which the final k'-th best is expected under a normal tail (plan_ladder_scale, plan.cpp)
// - only the tightness of the thresholds depends on that guess, never their validity. Spread 0 / no sample: delta 0, no ladder.
// Emitting sample (round 6, tile_epilogue's SAMPLE mode): dense[q][M .. M + floor_n) holds, per cell of the sample, the best coarse
// score the cell left out of the candidate lists; their maximum is the query's floor word (f2ord; 0 = nothing left out), which
// the final kernel folds into the threshold of its guarantee check.
static __global__ __launch_bounds__(256) void tau_kernel(const float* __restrict__ dense, uint32_t M, uint32_t ld, uint32_t nq,
                                                  uint32_t kprime, float* __restrict__ tau, uint32_t* __restrict__ nbest,
                                                  float4* __restrict__ ladc = nullptr, unsigned long long* __restrict__ lad = nullptr,
                                                  float lad_scale = 0.0f, uint32_t hi_rank = 0, uint32_t floor_n = 0,
                                                  uint32_t* __restrict__ floor_ord = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= nq) return;
    float hi = -INFINITY;
    const float t = kth_largest_wave(dense + (uint64_t)q * ld, M, kprime, lane, hi_rank, ladc ? &hi : nullptr);   // common.h
    if (floor_ord) {
        float fl = -INFINITY;
        for (uint32_t i = (uint32_t)lane; i < floor_n; i += 64u) fl = fmaxf(fl, dense[(uint64_t)q * ld + M + i]);
        const uint32_t fo = wave_max_u32(fl > -INFINITY ? f2ord(fl + 0.0f) : 0u);
        if (lane == 0) floor_ord[q] = fo;
    }
    if (lane == 0) {
        tau[q] = t;
        nbest[q] = 0u;
        if (ladc) {
            const float spread = hi - t;
            const float delta = (t > -INFINITY && spread > 0.0f && spread < INFINITY) ? lad_scale * spread : 0.0f;
            ladc[q] = make_float4(t, delta, delta > 0.0f ? 1.0f / delta : 0.0f, 0.0f);
            lad[q] = 0ull;
        }
    }
}

struct RescoreArgs {
    const uint64_t* best;   // [nq][kprime]
    const uint32_t* nbest;  // [nq]
    const float* tau;       // [nq] final coarse threshold (k'-th best coarse score)
    const char* rows;       // [n][ld]
    const char* qrows;      // [nq][ld]
    const float* norm_q;    // [nq] (dot-product error bound)
    const uint32_t* overflow;
    uint64_t* out_idx;      // [nq][k]
    float* out_score;       // [nq][k]
    uint32_t* fb_flag;      // [nq] 1 = needs exact full scan
    uint32_t* fb_count;     // [1]
    uint32_t* stat_maxerr;  // [1] f2ord(max |coarse-exact|)
    IdMap idmap;
    uint32_t nq, n, D, ld, kprime, k, metric;
    uint32_t rows_per_batch;  // candidate rows staged in LDS per pass
    float eps_scale;        // cosine: eps; dot: eps = eps_scale * |q| * max|c|   (host: coarse_eps_scale())
    float max_norm_c;
    // f32 index with a bf16 shadow (coarse scores come from ROUNDED operands): rounding residuals
    const float* qres;      // [nq][2]: relative |dq|/min(|q|,|q^|) and absolute |dq| of each query; NULL otherwise
    float res_rel_c;        // max over the corpus of |dc| / min(|c|, |c^|)
    float res_abs_c;        // max over the corpus of |dc|
    uint32_t* stat_maxeps;  // [1] f2ord-free max of the eps actually used (non-negative float bits)
    uint2* qstat;           // [nq] {max |coarse - exact| bits, eps bits} of each query, or NULL. Round 5: the batch-wide maxima used
                            // to be raised with one atomicMax per workgroup on ONE word - all 1024 workgroups of a C2 batch start
                            // within a microsecond, so every one of them saw the cleared word, and the atomics serialised at ~12 ns
                            // apiece: the slowest workgroup ended 14 us behind the median (phase stamps, r05). Now a plain store
                            // per query; publish_flags_kernel folds the maxima into the flag words.
    uint64_t* trace;        // diagnostics (CGV_TRACE=1): [nq][8] wall-clock stamps (100 MHz) of the phases, or NULL
    // Small batches (<= 64 queries = <= 64 workgroups): the LAST workgroup to finish does publish_flags_kernel's job itself - one
    // launch boundary and a 4 us kernel less on a call that is all latency (pub_host == NULL: the caller launches the kernel).
    // (Not for large batches: 1024 returning atomics on one word cost 25 us, r03b.)
    uint32_t* pub_flags;    // device flag words
    uint32_t* pub_host;     // their pinned, device-mapped mirror
    uint32_t* pub_count;    // arrival counter: zero at launch, zero again at exit
    uint32_t pub_words;     // n_flags | done word << 8 | error word << 16 | eps word << 24; the done word's marker is nq
    // COARSE_TOP2 searches: a query whose check fails ONLY because of the floor (a cell left out a row that may belong to the
    // top-k) is flagged 2 instead of 1 and leaves what top2_repair_kernel needs to re-scan just the offending cells
    // (kernels_repair.h): the threshold cells are compared with and the exact keys of its re-scored candidates. NULL: off.
    float* rep_theta;       // [nq]
    uint64_t* rep_keys;     // [nq][REPAIR_KEYS] exact keys (score, LOCAL row), descending
    uint32_t* rep_n;        // [nq] how many
    // Emitting sample: a query whose check fails ONLY because of the sample's floor - one of the sample's 64-row
    // cells left out a row that may belong to the top-k - is put right inside this kernel: the cells whose left-out score may
    // reach the top-k (usually one) are scored again with the reference's arithmetic and merged with the re-scored candidates
    // (sample_floor_repair below). No flag, no host round trip, no second exchange on the row-sharded path. NULL: off.
    const float* sfloor;    // the floor area of the sample's dump rows: sfloor[q * sfloor_ld + seq * 4 + cell]
    uint32_t sfloor_ld, sfloor_n;          // row pitch (floats); 4 x sample tiles
    uint32_t s_T1, s_R, s_P;               // the visiting order (stage_tile): sample position seq -> corpus tile
    uint32_t* sfix_count;   // [1] queries put right that way (statistics)
};
constexpr uint32_t REPAIR_KEYS = 64;
constexpr uint32_t SFIX_MAX_CELLS = 24;   // offending sample cells per query at most (64 rows each); more: the exact scan takes it

// End-of-search publication (one wave, launched behind the last kernel of a search): the flag words (fallback count,
// non-finite bits, error maxima, ...) go to the pinned, device-mapped host mirror and are cleared for the next search,
// so the host needs neither a flags D2H copy after the pipeline (a copy-engine launch on the critical path) nor a
// memset before the next one. host[done_word] = marker tells the host the mirror is current. (A ticket counter in
// the last kernel instead - 1024 returning atomics on one word - cost 25 us: r03b.)
// qstat (optional): the per-query {max observed error, eps} words of the final kernel ([nq]; RescoreArgs::qstat): their maxima
// are folded into flag words err_word / eps_word here (both are bit patterns of non-negative floats: unsigned order = float order).
static __global__ void publish_flags_kernel(uint32_t* __restrict__ flags, uint32_t* __restrict__ host, uint32_t n_flags,
                                     uint32_t done_word, uint32_t marker, uint32_t* __restrict__ extra = nullptr,
                                     uint32_t n_extra = 0u, const uint2* __restrict__ qstat = nullptr, uint32_t nq = 0u,
                                     uint32_t err_word = 0u, uint32_t eps_word = 0u) {
    __shared__ uint32_t red[2];
    const uint32_t i = threadIdx.x;
    if (i < 2) red[i] = 0u;
    __syncthreads();
    if (qstat) {
        uint32_t me = 0u, mp = 0u;
        for (uint32_t q = i; q < nq; q += blockDim.x) {
            const uint2 v = qstat[q];
            me = v.x > me ? v.x : me;
            mp = v.y > mp ? v.y : mp;
        }
        me = wave_max_u32(me);
        mp = wave_max_u32(mp);
        if ((i & 63u) == 0u) {
            atomicMax(&red[0], me);
            atomicMax(&red[1], mp);
        }
    }
    __syncthreads();
    for (uint32_t j = i; j < n_extra; j += blockDim.x) extra[j] = 0u;  // rendezvous words of a fused sample + emit launch
    if (i < n_flags) {
        uint32_t v = __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (qstat && i == err_word) v = v > red[0] ? v : red[0];
        if (qstat && i == eps_word) v = v > red[1] ? v : red[1];
        host[i] = (i == done_word) ? marker : v;
        __hip_atomic_store(flags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// Exact reference arithmetic on the k' candidates of each query, exact (score desc, row asc)
// ordering, and the guarantee check  e_k > tau + eps  (every row outside the candidate set has
// coarse <= tau, hence exact <= tau + eps < e_k).
// One workgroup (256 threads) per query. The k' candidate rows are gathered from HBM into
// LDS cooperatively (every 16-byte piece is an independent load: one memory round trip instead
// of a serial walk over 12+ cache lines per row); then 8 lanes per candidate = the 8 AVX2 lanes
// of the reference (common.h) run the FMA chains out of LDS, 32 candidates at a time.
// Dynamic LDS: [query row: ld*esize B][batch rows: (ld*esize + 16) B each, +16 B pad against bank
// conflicts]. rows_per_batch is chosen by the host (>= 1).
// Query row -> LDS (linear element order), issued as early as the caller can: the loads are independent of the
// selection that runs first. Visible to all waves after the caller's next __syncthreads().
template <int DT>
__device__ inline void stage_query_row(const RescoreArgs& a, uint32_t q, char* qs, int tid) {
    const uint32_t pieces = a.ld * Elem<DT>::bytes / 16;
    const Row<DT> qr = make_row<DT>(a.qrows, q, a.ld);
    for (uint32_t pc = tid; pc < pieces; pc += 256) *(uint4*)(qs + (size_t)pc * 16) = *(const uint4*)piece_ptr<DT>(qr, pc);
}

// The emitting sample's way out of a floor violation, inside the final kernel (one workgroup per query; rare: 6e-6 per query on
// C2's 8-GPU shard - but a bench that cycles through four query batches meets the same query every fourth step, and on the
// row-sharded path every such query cost an exact scan AND a second exchange, on every rank).
// theta = e_k - eps: a sampled row outside the candidates and outside a cell with left-out score > theta has coarse <= theta,
// hence exact <= theta + eps < e_k. The offending cells' rows (cell = (sample position, M-half, lane half): 64 rows in the MFMA
// C layout, kernels_coarse.h tile_epilogue) are staged through the re-score's LDS region, 8 lanes per row run the reference's
// chains, rows above the current k-th key that are not candidates already join the keys; the top-k of the union is the exact
// top-k (the k-th score can only rise: no second check). Returns false (uniform) when it gives up: too many cells or keys, a NaN.
template <int DT>
__device__ inline bool sample_floor_repair(const RescoreArgs& a, uint32_t q, uint32_t nres, uint64_t* ekeys /* LDS [CAND_CAPS] */,
                                                 float theta, char* rs, const char* qs, int tid) {
    __shared__ uint32_t fcells[SFIX_MAX_CELLS];
    __shared__ uint32_t fn, fnew, fbad;
    if (tid == 0) {
        fn = 0;
        fnew = 0;
        fbad = 0;
    }
    __syncthreads();
    const float* fl = a.sfloor + (uint64_t)q * a.sfloor_ld;
    for (uint32_t c = (uint32_t)tid; c < a.sfloor_n; c += 256u)
        if (fl[c] > theta) {
            const uint32_t p = atomicAdd(&fn, 1u);
            if (p < SFIX_MAX_CELLS) fcells[p] = c;
        }
    __syncthreads();
    const uint32_t ncell = fn;
    if (ncell > SFIX_MAX_CELLS) return false;   // uniform
    const uint64_t kth = ekeys[a.k - 1];
    const uint32_t rowb = a.ld * Elem<DT>::bytes, pitch = rowb + 16, pieces = rowb / 16;
    const uint32_t total = ncell * 64u, rpb = a.rows_per_batch;
    auto row_of = [&](uint32_t it) -> uint32_t {   // item -> corpus row (0xFFFFFFFF: beyond the corpus' last row)
        const uint32_t c = fcells[it >> 6], i = it & 63u, lg = c & 3u;
        const uint32_t tile = stage_tile(a.s_T1, a.s_R, a.s_P, c >> 2);
        const uint64_t rw = (uint64_t)tile * 256u + (lg >> 1) * 128u + (i >> 4) * 32u + (i & 3u) + 8u * ((i >> 2) & 3u) + 4u * (lg & 1u);
        return rw < a.n ? (uint32_t)rw : 0xFFFFFFFFu;
    };
    for (uint32_t r0 = 0; r0 < total; r0 += rpb) {
        const uint32_t nbat = (total - r0) < rpb ? (total - r0) : rpb;
        __syncthreads();   // (the previous pass' LDS reads are done)
        const uint32_t nload = nbat * pieces;
        for (uint32_t i0 = (uint32_t)tid; i0 < nload; i0 += 256u * 8u) {   // up to 8 independent 16-byte loads per thread in flight
            uint4 v[8];
            uint32_t ok = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + (uint32_t)u * 256u;
                if (i < nload) {
                    const uint32_t slot = i / pieces, pc = i - slot * pieces, row = row_of(r0 + slot);
                    if (row != 0xFFFFFFFFu) {
                        v[u] = *(const uint4*)piece_ptr<DT>(make_row<DT>(a.rows, row, a.ld), pc);
                        ok |= 1u << u;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + (uint32_t)u * 256u;
                if (ok & (1u << u)) {
                    const uint32_t slot = i / pieces, pc = i - slot * pieces;
                    *(uint4*)(rs + (size_t)slot * pitch + (size_t)pc * 16u) = v[u];
                }
            }
        }
        __syncthreads();
        const LdsRow<DT> ql{qs};
        for (uint32_t sl = (uint32_t)tid >> 3; sl < nbat; sl += 32u) {   // uniform within an 8-lane group
            const uint32_t row = row_of(r0 + sl);
            if (row == 0xFFFFFFFFu) continue;
            const LdsRow<DT> cl{rs + (size_t)sl * pitch};
            const float sc = exact_score_group8((int)a.metric, ql, cl, a.D, tid & 7);
            if ((tid & 7) == 0) {
                if (sc != sc) fbad = 1u;
                const uint64_t key = make_key(sc, row);
                if (key > kth) {
                    bool have = false;   // a candidate already (its key is the same: the same arithmetic on the same values)
                    for (uint32_t j = 0; j < nres; ++j) have = have || key_row(ekeys[j]) == row;
                    if (!have) {
                        const uint32_t p = atomicAdd(&fnew, 1u);
                        if (nres + p < CAND_CAPS) ekeys[nres + p] = key;
                    }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nnew = fnew;
    if (fbad != 0u || nres + nnew > CAND_CAPS) return false;   // uniform
    const uint32_t nall = nres + nnew, P = next_pow2(nall < 2u ? 2u : nall);
    for (uint32_t i = nall + (uint32_t)tid; i < P; i += 256u) ekeys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc<256>(ekeys, P, tid);
    for (uint32_t j = (uint32_t)tid; j < a.k; j += 256u) {
        a.out_idx[(uint64_t)q * a.k + j] = map_id(a.idmap, key_row(ekeys[j]));   // (nall >= nres >= k)
        a.out_score[(uint64_t)q * a.k + j] = key_score(ekeys[j]);
    }
    return true;
}

template <int DT>
__device__ inline void rescore_body(const RescoreArgs& a, uint32_t q, uint32_t nb, const uint64_t* ckeys /* LDS [nb] */,
                                    float tau, bool overflow, char* rs /* LDS: staged candidate rows */,
                                    const char* qs /* LDS: the query row (stage_query_row) */, int tid,
                                    float tau_lists = INFINITY /* COARSE_TOP2: the threshold WITHOUT the floor (else unused) */) {
    __shared__ uint64_t ekeys[CAND_CAPS];
    __shared__ uint32_t maxerr, tripped, repair_s, fix_s;
    __shared__ float fix_theta;
    const uint32_t rowb = a.ld * Elem<DT>::bytes, pitch = rowb + 16, pieces = rowb / 16;
    if (tid == 0) {
        maxerr = 0;
        tripped = 0;
        repair_s = 0;
    }
    // eps: bound on |coarse - exact| for THIS query, in score units
    float eps = a.eps_scale;
    if (a.metric == METRIC_DOT) eps = a.eps_scale * a.norm_q[q] * a.max_norm_c;
    if (a.qres) {
        // |cos(q^,c^) - cos(q,c)| <= |u^-u| + |v^-v| <= res_rel(q) + res_rel(c)   (unit vectors u, v);
        // |q^.c^ - q.c| <= |dq||c^| + |q||dc|. 1 % head room for the f32 rounding of the residual sums.
        const float rq_rel = a.qres[2 * q], rq_abs = a.qres[2 * q + 1];
        if (a.metric == METRIC_DOT)
            eps += 1.01f * (rq_abs * a.max_norm_c + (a.norm_q[q] + rq_abs) * a.res_abs_c);
        else
            eps += 1.01f * (rq_rel + a.res_rel_c);
    }
    const float trip = 0.5f * eps;
    // the batch-wide statistics words, looked at before they are raised (below): fetched now, used ~20 us later
    uint32_t seen_maxerr = 0, seen_maxeps = 0;
    if (tid == 0 && !a.qstat) {
        seen_maxerr = __hip_atomic_load(a.stat_maxerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.stat_maxeps) seen_maxeps = __hip_atomic_load(a.stat_maxeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Re-score only the candidates that can reach the top-k. The list is sorted by coarse score c_0 >= c_1 >= ...; with
    // |coarse - exact| <= eps the k best by coarse score have exact >= c_{k-1} - eps, so a candidate with
    // c_i < c_{k-1} - 2 eps cannot be among the exact top-k. On random data that leaves k or k + 1 of the k' = 16
    // candidates: a third fewer rows to gather (the gather is the longest phase of this kernel: 25 MB of 64-byte
    // pieces per C2 batch, ~16 us at the memory system's random-access rate). Everything not re-scored - the other
    // candidates and every row outside the list - has coarse <= tau_eff, the first skipped candidate's score.
    const uint32_t nb_all = nb;
    float tau_eff = tau, tau_eff_lists = tau_lists;
    if (nb > a.k && a.k > 0) {
        const float cut = key_score(ckeys[a.k - 1]) - 2.0f * eps;
        uint32_t m = a.k;
        if (nb <= 64) {   // the list is sorted: the prefix at or above the cut = k + the number of later entries at or above it -
            // one LDS read per lane and a ballot instead of a dependent walk of up to k' LDS round trips in every thread
            const uint32_t l = (uint32_t)tid & 63u;
            const bool in = l >= a.k && l < nb && key_score(ckeys[l]) >= cut;
            m = a.k + (uint32_t)__popcll(__ballot(in));
        } else
        while (m < nb && key_score(ckeys[m]) >= cut) ++m;  // uniform: every thread walks the same short list
        if (m < nb) {
            // (the first skipped candidate bounds the skipped ones; `tau` everything outside the list - it is the larger of the
            // two only on the COARSE_TOP2 path, where a cell's left-out row may beat the list's tail)
            tau_eff = fmaxf(tau, key_score(ckeys[m]));
            tau_eff_lists = fmaxf(tau_lists == INFINITY ? -INFINITY : tau_lists, key_score(ckeys[m]));
            nb = m;
        }
    }
    const uint32_t nres = nb;
    const uint32_t P = next_pow2(nb < 2 ? 2 : nb);
    for (uint32_t i = nb + tid; i < P; i += 256) ekeys[i] = 0ull;
    // Gather: 8 candidates at a time, 32 lanes per candidate, lane l of a candidate's group fetches pieces l, l + 32, ...
    // (no integer division per piece); up to 8 independent 16-byte loads per thread are issued before the first LDS
    // store, so the 16 x 96 pieces of a C2 query travel in ONE memory round trip (r03a: two round trips, 14 us).
    const uint32_t grp = (uint32_t)tid >> 5, l32 = (uint32_t)tid & 31u;
    const uint32_t nj = (pieces + 31) / 32;  // 32-piece slabs per row
    for (uint32_t c0 = 0; c0 < nb; c0 += a.rows_per_batch) {
        const uint32_t nbat = (nb - c0) < a.rows_per_batch ? (nb - c0) : a.rows_per_batch;
        __syncthreads();
        const uint32_t items = ((nbat + 7) / 8) * nj;  // (round of 8 candidates, slab) pairs, slab fastest
        uint32_t cr = 0, j = 0;                         // round and slab of item i0 (uniform, kept incrementally)
        for (uint32_t i0 = 0; i0 < items; i0 += 8) {
            uint4 v[8];
            uint32_t cu[8], pu[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cu[u] = (i0 + u < items) ? cr * 8 + grp : 0xFFFFFFFFu;
                pu[u] = j * 32 + l32;
                if (++j == nj) {
                    j = 0;
                    ++cr;
                }
                if (cu[u] < nbat && pu[u] < pieces) {
                    const uint32_t row = key_row(ckeys[c0 + cu[u]]);
                    v[u] = *(const uint4*)piece_ptr<DT>(make_row<DT>(a.rows, row, a.ld), pu[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (cu[u] < nbat && pu[u] < pieces) *(uint4*)(rs + (size_t)cu[u] * pitch + (size_t)pu[u] * 16) = v[u];
        }
        __syncthreads();
        phase_stamp(a.trace, q, 3, tid);
        const LdsRow<DT> ql{qs};
        for (uint32_t c = (uint32_t)tid >> 3; c < nbat; c += 32) {
            const uint64_t key = ckeys[c0 + c];
            const uint32_t row = key_row(key);
            const float coarse = key_score(key);
            const LdsRow<DT> cl{rs + (size_t)c * pitch};
            const float ex = exact_score_group8(a.metric, ql, cl, a.D, tid & 7);
            if ((tid & 7) == 0) {
                ekeys[c0 + c] = make_key(ex, row);
                const float err = fabsf(ex - coarse);
                if (err == err) atomicMax(&maxerr, __float_as_uint(err));
                // trip-wire: the bound is derived under an explicit model of the matrix pipe's internal
                // accumulation; an observed error beyond HALF of it on any candidate means the model
                // cannot be trusted for this query -> exact full scan (never a silent wrong answer)
                if (!(err <= trip)) tripped = 1u;
            }
        }
    }
    __syncthreads();
    phase_stamp(a.trace, q, 4, tid);
    if (P <= 64) {  // the usual case: one key per lane of wave 0, bitonic network over lane shuffles - no barriers
        if (tid < 64) {
            uint64_t kx = (uint32_t)tid < P ? ekeys[tid] : 0ull;
            for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1)
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                    const uint64_t other = __shfl_xor(kx, (int)j, 64);
                    const bool upper = ((uint32_t)tid & j) != 0, desc = ((uint32_t)tid & k2) == 0;
                    // descending block: the lower lane keeps the larger key
                    const bool take_max = (upper != desc);
                    kx = take_max ? (kx > other ? kx : other) : (kx < other ? kx : other);
                }
            if ((uint32_t)tid < P) ekeys[tid] = kx;
        }
        __syncthreads();
    } else {
        bitonic_sort_desc<256>(ekeys, P, tid);
    }
    phase_stamp(a.trace, q, 5, tid);
    for (uint32_t j = tid; j < a.k; j += 256) {
        uint64_t oi = UINT64_MAX;
        float os = -INFINITY;
        if (j < nb) {
            oi = map_id(a.idmap, key_row(ekeys[j]));
            os = key_score(ekeys[j]);
        }
        a.out_idx[(uint64_t)q * a.k + j] = oi;
        a.out_score[(uint64_t)q * a.k + j] = os;
    }
    phase_stamp(a.trace, q, 6, tid);
    if (tid == 0) {
        // statistics: one word for the whole batch. An unconditional atomicMax from each of the 1024 workgroups
        // serialises at ~12 ns apiece on that word; almost none of them raises the maximum, so look first (a stale
        // read only costs a redundant atomic).
        if (a.qstat && a.pub_host)   // (read by another workgroup of this launch: written through)
            __hip_atomic_store((unsigned long long*)(a.qstat + q),
                               (unsigned long long)maxerr | ((unsigned long long)(tau_eff > -INFINITY ? __float_as_uint(eps) : 0u) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (a.qstat) a.qstat[q] = make_uint2(maxerr, tau_eff > -INFINITY ? __float_as_uint(eps) : 0u);
        else if (maxerr > seen_maxerr) atomicMax(a.stat_maxerr, maxerr);
        bool fb = overflow;
        if (tau_eff > -INFINITY) {  // rows at or below tau_eff were dropped / not re-scored: check the guarantee
            if (!a.qstat && a.stat_maxeps && __float_as_uint(eps) > seen_maxeps) atomicMax(a.stat_maxeps, __float_as_uint(eps));
            if (nres < a.k && nres < nb_all) fb = true;  // (cannot happen: the re-scored prefix holds >= k candidates)
            if (nb_all < a.k) fb = true;                 // (the corpus has more rows than candidates survived)
            else if (!(key_score(ekeys[a.k - 1]) > tau_eff + eps)) fb = true;
            if (tripped) fb = true;
        }
        // the floor alone failed the check (COARSE_TOP2: three of the query's best rows in one cell): flag 2 - the cells whose
        // left-out score could reach the top-k are re-scanned exactly (top2_repair_kernel) instead of the whole corpus.
        // Rows outside the candidates and outside a re-scanned cell have coarse <= theta, hence exact <= theta + eps < e_k.
        uint32_t flag = fb ? 1u : 0u;
        bool fix = false;
        if (fb && (a.rep_theta || a.sfloor) && tau_lists != INFINITY) {
            const float ek = nres >= a.k ? key_score(ekeys[a.k - 1]) : -INFINITY;
            // (why not, in the upper bits of the word - diagnostics only; any non-zero word means "exact scan")
            const uint32_t why = (overflow ? 1u : 0u) | (tripped ? 2u : 0u) | ((nb_all < a.k || nres < a.k) ? 4u : 0u) |
                                 (!(ek > tau_eff_lists + eps) ? 8u : 0u) | (nres > REPAIR_KEYS ? 16u : 0u) |
                                 (!(fabsf(ek) < INFINITY) ? 32u : 0u);
            if (why == 0u && a.rep_theta) {
                a.rep_theta[q] = ek - 1.001f * eps - 1e-30f;
                a.rep_n[q] = nres;
                flag = 2u;
                repair_s = 1u;
            } else if (why == 0u) {   // emitting sample: put right below, by this workgroup
                fix_theta = ek - 1.001f * eps - 1e-30f;
                fix = true;
            } else {
                flag = 1u | (why << 8);
            }
        }
        fix_s = fix ? 1u : 0u;
        if (!fix) {
            a.fb_flag[q] = flag;
            if (fb) atomicAdd(a.fb_count, 1u);
        }
    }
    if (a.rep_theta) {   // uniform
        __syncthreads();
        if (repair_s != 0u)
            for (uint32_t j = tid; j < nres; j += 256) a.rep_keys[(uint64_t)q * REPAIR_KEYS + j] = ekeys[j];
    }
    if (a.sfloor) {   // uniform
        __syncthreads();
        if (fix_s != 0u) {   // uniform (LDS)
            const bool done = sample_floor_repair<DT>(a, q, nres, ekeys, fix_theta, rs, qs, tid);
            if (tid == 0) {
                a.fb_flag[q] = done ? 0u : (1u | (64u << 8));
                if (done) atomicAdd(a.sfix_count, 1u);
                else atomicAdd(a.fb_count, 1u);
            }
        }
    }
    phase_stamp(a.trace, q, 7, tid);
    if (a.pub_host) {   // uniform: a small batch publishes its own flags (RescoreArgs::pub_*)
        __shared__ uint32_t pub_last;
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this query's statistics / fallback count have left
            pub_last = (__hip_atomic_fetch_add(a.pub_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.nq - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (pub_last != 0u && tid < 64) {   // nq <= 64: one statistics word pair per lane
            uint32_t me = 0u, mp = 0u;
            if (a.qstat && (uint32_t)tid < a.nq) {
                const unsigned long long v = __hip_atomic_load((const unsigned long long*)(a.qstat + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                me = (uint32_t)v;
                mp = (uint32_t)(v >> 32);
            }
            me = wave_max_u32(me);
            mp = wave_max_u32(mp);
            const uint32_t pub_n = a.pub_words & 255u, done_word = (a.pub_words >> 8) & 255u, err_word = (a.pub_words >> 16) & 255u,
                           eps_word = a.pub_words >> 24;
            if ((uint32_t)tid < pub_n) {
                uint32_t v = __hip_atomic_load(a.pub_flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.qstat && (uint32_t)tid == err_word) v = v > me ? v : me;
                if (a.qstat && (uint32_t)tid == eps_word) v = v > mp ? v : mp;
                a.pub_host[tid] = ((uint32_t)tid == done_word) ? a.nq : v;
                __hip_atomic_store(a.pub_flags + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) __hip_atomic_store(a.pub_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Dynamic LDS of rescore_kernel / final_kernel: [work region: staged candidate rows (and, in final_kernel, the
// selection's key buffer first)][query row: ld * esize bytes at offset a.qoff].
template <int DT>
__global__ __launch_bounds__(256) void rescore_kernel(const RescoreArgs a, uint32_t qoff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint64_t ckeys[CAND_CAPS];
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    stage_query_row<DT>(a, q, smem + qoff, tid);
    const uint32_t nb = a.nbest[q];
    for (uint32_t i = tid; i < nb; i += 256) ckeys[i] = a.best[(uint64_t)q * a.kprime + i];
    __syncthreads();
    rescore_body<DT>(a, q, nb, ckeys, a.tau[q], a.overflow[q] != 0, smem, smem + qoff, tid);
}

// Last stage, k' <= 64: the final selection (merge of best[q] with the last launch's candidate
// sub-lists) and the exact re-score in ONE launch - the top-k' keys go from the extraction straight
// into the re-score through LDS instead of a best[]/tau[] round trip and a second launch.
// Dynamic LDS = max(select's key buffer + prefix, re-score's staged rows) + the query row behind it (fetched
// first, while the selection runs); the key buffer is dead once the extraction is done.
template <int DT>
__global__ __launch_bounds__(256) void final_kernel(const SelectArgs sa, const RescoreArgs a, uint32_t qoff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint64_t ckeys[CAND_CAPS];
    __shared__ uint64_t part[4 * 64];
    __shared__ uint64_t outk[64];
    uint64_t* keys = (uint64_t*)smem;
    uint32_t* pre = (uint32_t*)(smem + (size_t)sa.lds_keys * 8);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    bool trunc = false;
    phase_stamp(a.trace, q, 0, tid);
    stage_query_row<DT>(a, q, smem + qoff, tid);
    const uint32_t M = gather_keys(sa, q, keys, pre, tid, &trunc);
    phase_stamp(a.trace, q, 1, tid);
    const uint32_t keep = M < sa.kprime ? M : sa.kprime;
    extract_topk(keys, M, keep, part, outk, tid);
    phase_stamp(a.trace, q, 2, tid);
    for (uint32_t i = tid; i < keep; i += 256) ckeys[i] = outk[i];
    // fewer than k' keys: nothing is cut here, but the coarse launches dropped every row at or below THEIR threshold
    float tau = (M >= sa.kprime) ? key_score(outk[sa.kprime - 1]) : sa.tau[q];
    float tau_lists = INFINITY;   // (COARSE_TOP2 only: the threshold of the candidate lists alone, without the floor)
    if (sa.floor_ord) {   // COARSE_TOP2: rows no cell kept score at most the floor; the lists hold no threshold of their own
        if (M < sa.kprime && !sa.floor_with_tau) tau = -INFINITY;
        tau_lists = tau;
        const uint32_t fo = sa.floor_ord[q];
        if (fo != 0u) tau = fmaxf(tau, ord2f(fo));
    }
    const bool overflow = trunc || sa.overflow[q] != 0;
    __syncthreads();
    if (sa.floor_ord && tid == 0) sa.floor_ord[q] = 0u;   // (every thread has read it: the barrier above)
    rescore_body<DT>(a, q, keep, ckeys, tau, overflow, smem, smem + qoff, tid, tau_lists);
}

// id slot 0 of a PROVISIONAL record (pack_topk_kernel below): never a row id
constexpr uint64_t PROVISIONAL_ID = 0xFFFFFFFFFFFFFFFEull;  // == CGV_PROVISIONAL_ID (cgvec.h)

// The same merge without the LDS capacity limit (G * k <= 4096 above): ONE WAVE per query walks the G <= 64 sorted lists
// like a G-way merge - lane g holds the head of list g, a round is a wave maximum of the ordered scores, then the smallest
// id among the lanes that hold it, and the winner advances. k dependent rounds: slow (k = 2048: a few ms per batch), but
// any k up to CGV_MAX_K on any number of shards is served (the LDS merge alone rejected n_shards * k > 4096).
static __global__ __launch_bounds__(256) void merge_topk_wave_kernel(const char* __restrict__ idx_base, uint64_t idx_stride,
                                                              const char* __restrict__ score_base, uint64_t score_stride,
                                                              uint32_t G, uint32_t nq, uint32_t k,
                                                              uint64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                              uint32_t* __restrict__ redo = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= nq) return;
    const bool mine = (uint32_t)lane < G;
    const char* ip = idx_base + ((uint64_t)(mine ? lane : 0) * nq + q) * idx_stride;
    const char* sp = score_base + ((uint64_t)(mine ? lane : 0) * nq + q) * score_stride;
    uint32_t h = 0;
    uint64_t id = UINT64_MAX;
    float sc = -INFINITY;
    auto load_head = [&]() {
        id = UINT64_MAX;
        sc = -INFINITY;
        if (mine && h < k) {
            id = *(const uint64_t*)(ip + (uint64_t)h * 8);
            sc = *(const float*)(sp + (uint64_t)h * 4);
            if (id == PROVISIONAL_ID) {  // this rank will redo the query: the batch's exchange is repeated (pack_topk_kernel)
                if (redo) *redo = 1u;
                id = UINT64_MAX;
                sc = -INFINITY;
            }
        }
    };
    load_head();
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t ord = (id != UINT64_MAX) ? f2ord(sc + 0.0f) : 0u;  // 0 = exhausted (lists are padded at the tail)
        const uint32_t m = wave_max_u32(ord);
        uint64_t oi = UINT64_MAX;
        float os = -INFINITY;
        if (m != 0u) {
            const bool cand = ord == m;
            const uint32_t hi = ~wave_max_u32(cand ? ~(uint32_t)(id >> 32) : 0u);          // smallest high word
            const bool cand2 = cand && (uint32_t)(id >> 32) == hi;
            const uint32_t lo = ~wave_max_u32(cand2 ? ~(uint32_t)id : 0u);                  // then smallest low word
            const bool win = cand2 && (uint32_t)id == lo;
            oi = ((uint64_t)hi << 32) | lo;
            os = ord2f(m);
            if (win) {
                ++h;
                load_head();
            }
        }
        if (lane == 0) {
            out_idx[(uint64_t)q * k + j] = oi;
            out_score[(uint64_t)q * k + j] = os;
        }
    }
}

// Per-shard results -> one packed record row per query for the single all-gather of SURVEY.md §8(e):
// w int32 per query = k u64 ids | k f32 scores | (k odd: one pad word, keeps the next row 8-byte aligned).
__host__ __device__ inline uint32_t packed_width(uint32_t k) { return 3u * k + (k & 1u); }
// PROVISIONAL records (join-free exchange, cgv_search_packed_begin_f32_dev): a query whose top-k the device could not prove
// (prov[q] != 0: it will be re-run through the exact scan once the host looks at the flags) - or every query when
// prov_all is set (an exact-scan-only index) - is packed with id slot 0 = PROVISIONAL_ID. The merge kernels raise their redo
// word when they meet one: every rank merges the same gathered records, so all ranks learn it without another collective.
static __global__ void pack_topk_kernel(const uint64_t* __restrict__ idx, const float* __restrict__ score, uint32_t nq,
                                 uint32_t k, uint32_t* __restrict__ out, const uint32_t* __restrict__ prov = nullptr,
                                 uint32_t prov_all = 0u) {
    const uint32_t w = packed_width(k);
    const uint64_t total = (uint64_t)nq * w;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t q = (uint32_t)(i / w), c = (uint32_t)(i % w);
        uint32_t v = 0u;
        if (c < 2 * k) {
            uint64_t id = idx[(uint64_t)q * k + (c >> 1)];
            if (c < 2 && (prov_all || (prov && prov[q] != 0u))) id = PROVISIONAL_ID;
            v = (c & 1u) ? (uint32_t)(id >> 32) : (uint32_t)id;
        } else if (c < 3 * k) {
            v = __float_as_uint(score[(uint64_t)q * k + (c - 2 * k)]);
        }
        out[i] = v;
    }
}

// Merge G partial top-k lists per query (after the all-gather of per-shard results,
// SURVEY.md §8(e)). Input [g][nq][k]; one workgroup per query; G*k <= 4096.
// (rank g, query q) lists are addressed as base + (g*nq + q) * stride bytes, so the same kernel reads
// separate [g][nq][k] id / score arrays or the packed records of the all-gather (pack_topk_kernel).
static __global__ __launch_bounds__(256) void merge_topk_kernel(const char* __restrict__ idx_base, uint64_t idx_stride,
                                                         const char* __restrict__ score_base,
                                                         uint64_t score_stride, uint32_t G, uint32_t nq,
                                                         uint32_t k, uint64_t* __restrict__ out_idx,
                                                         float* __restrict__ out_score, uint32_t* __restrict__ redo = nullptr) {
    // Global ids are 64-bit, so sort (ordered score, position) keys and carry the id
    // through the position; ties on score are resolved by a second pass on the id.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t M = G * k;
    const uint32_t P = next_pow2(M < 2 ? 2 : M);
    uint64_t* keys = (uint64_t*)smem;         // [P] (ord score << 32 | ~pos)
    uint64_t* ids = keys + P;                 // [P] global ids by position
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    for (uint32_t i = tid; i < P; i += 256) {
        uint64_t key = 0ull, id = UINT64_MAX;
        if (i < M) {
            const uint32_t gi = i / k, j = i % k;
            id = *(const uint64_t*)(idx_base + ((uint64_t)gi * nq + q) * idx_stride + (uint64_t)j * 8);
            const float s = *(const float*)(score_base + ((uint64_t)gi * nq + q) * score_stride + (uint64_t)j * 4);
            if (id == PROVISIONAL_ID) {  // rank gi will redo this query: the batch's exchange is repeated (pack_topk_kernel)
                if (redo) *redo = 1u;
                id = UINT64_MAX;
            }
            if (id != UINT64_MAX) key = make_key(s, i);
        }
        keys[i] = key;
        ids[i] = id;
    }
    __syncthreads();
    // Order by (score desc, id asc): bitonic sort with a comparator that looks the id up.
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P; i += 256) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t ka = keys[i], kb = keys[ixj];
                    const uint32_t sa = (uint32_t)(ka >> 32), sb = (uint32_t)(kb >> 32);
                    bool a_lt_b;  // "a ranks after b"
                    if (sa != sb)
                        a_lt_b = sa < sb;
                    else if (ka == 0ull || kb == 0ull)
                        a_lt_b = (ka == 0ull) && (kb != 0ull);
                    else
                        a_lt_b = ids[key_row(ka)] > ids[key_row(kb)];
                    bool b_lt_a;
                    if (sa != sb)
                        b_lt_a = sb < sa;
                    else if (ka == 0ull || kb == 0ull)
                        b_lt_a = (kb == 0ull) && (ka != 0ull);
                    else
                        b_lt_a = ids[key_row(kb)] > ids[key_row(ka)];
                    const bool desc_blk = ((i & k2) == 0);
                    if (desc_blk ? a_lt_b : b_lt_a) {
                        keys[i] = kb;
                        keys[ixj] = ka;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t j = tid; j < k; j += 256) {
        uint64_t oi = UINT64_MAX;
        float os = -INFINITY;
        if (j < M && keys[j] != 0ull) {
            oi = ids[key_row(keys[j])];
            os = key_score(keys[j]);
        }
        out_idx[(uint64_t)q * k + j] = oi;
        out_score[(uint64_t)q * k + j] = os;
    }
}

}  // namespace cgv
