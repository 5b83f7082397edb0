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
};

// One workgroup (256 threads) per query: merge best[q] with the nsplit candidate
// sub-lists written by the last coarse launch (or with the dense boot scores), keep the
// top-k', publish tau[q] = the k'-th best coarse score seen so far (a valid lower bound of
// the final k'-th best).
//   gather : 256/nsplit threads per sub-list copy it into LDS (prefix sums give the offsets)
//   select : k' <= 64 -> k' rounds of workgroup arg-max over the LDS keys (no full sort:
//            M is ~1e3 and k' ~16); larger k' -> LDS bitonic sort.
__global__ __launch_bounds__(256) void select_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = (uint64_t*)smem;                            // [lds_keys]
    uint32_t* pre = (uint32_t*)(smem + (size_t)a.lds_keys * 8);  // [nsplit + 1]
    __shared__ uint64_t wmax[4];
    __shared__ uint64_t outk[64];
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t qt = q / a.bn, ql = q % a.bn;
    const uint32_t nb = a.nbest[q];
    uint32_t M;
    bool trunc = false;
    if (a.dense) {
        M = nb + a.n_dense;
        if (M > a.lds_keys) {
            trunc = true;
            M = a.lds_keys;
        }
        for (uint32_t e = tid; e < M; e += 256)
            keys[e] = (e < nb) ? a.best[(uint64_t)q * a.kprime + e]
                               : make_key(a.dense[(uint64_t)q * a.n_dense + (e - nb)], e - nb);
    } else {
        // exclusive prefix of the sub-list counts: pre[s] = nb + sum_{s' < s} cnt[s']
        if (tid < 64) {
            const uint32_t per = (a.nsplit + 63) / 64;
            uint32_t loc = 0;
            for (uint32_t i = 0; i < per; ++i) {
                const uint32_t s = tid * per + i;
                if (s < a.nsplit) loc += a.cand_cnt[(uint64_t)(qt + s * a.nqt) * a.bn + ql];
            }
            uint32_t inc = loc;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(inc, off, 64);
                if (tid >= off) inc += v;
            }
            uint32_t run = nb + inc - loc;
            for (uint32_t i = 0; i < per; ++i) {
                const uint32_t s = tid * per + i;
                if (s < a.nsplit) {
                    pre[s] = run;
                    run += a.cand_cnt[(uint64_t)(qt + s * a.nqt) * a.bn + ql];
                }
            }
            if (tid == 63) pre[a.nsplit] = nb + inc;
        }
        __syncthreads();
        M = pre[a.nsplit];
        if (M > a.lds_keys) {  // cannot happen with sane thresholds; flag for the exact path
            trunc = true;
            M = a.lds_keys;
        }
        for (uint32_t e = tid; e < nb && e < M; e += 256) keys[e] = a.best[(uint64_t)q * a.kprime + e];
        // tps threads per sub-list (power of two), sub-lists in rounds of 256/tps
        uint32_t tps = 256;
        while (tps > 1 && tps * a.nsplit > 256) tps >>= 1;
        const uint32_t per_round = 256 / tps;
        const uint32_t sub = tid % tps;
        for (uint32_t s = tid / tps; s < a.nsplit; s += per_round) {
            const uint32_t lo = pre[s], hi = pre[s + 1];
            const uint2* src = a.cand + ((uint64_t)(qt + s * a.nqt) * a.bn + ql) * CAND_CAPS;
            for (uint32_t e = lo + sub; e < hi && e < M; e += tps) {
                const uint2 c = src[e - lo];
                keys[e] = make_key(__uint_as_float(c.x), c.y);
            }
        }
    }
    const uint32_t keep = M < a.kprime ? M : a.kprime;
    if (a.kprime <= 64) {
        __syncthreads();
        // keep rounds of arg-max; keys are unique, the owner zeroes the winner
        for (uint32_t r = 0; r < keep; ++r) {
            uint64_t mx = 0ull;
            for (uint32_t e = tid; e < M; e += 256) {
                const uint64_t v = keys[e];
                mx = v > mx ? v : mx;
            }
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t o = __shfl_xor(mx, off, 64);
                mx = o > mx ? o : mx;
            }
            if ((tid & 63) == 0) wmax[tid >> 6] = mx;
            __syncthreads();
            uint64_t w = wmax[0];
            w = wmax[1] > w ? wmax[1] : w;
            w = wmax[2] > w ? wmax[2] : w;
            w = wmax[3] > w ? wmax[3] : w;
            for (uint32_t e = tid; e < M; e += 256)
                if (keys[e] == w) keys[e] = 0ull;
            if (tid == 0) outk[r] = w;
            __syncthreads();
        }
        for (uint32_t i = tid; i < keep; i += 256) a.best[(uint64_t)q * a.kprime + i] = outk[i];
        if (tid == 0) {
            a.nbest[q] = keep;
            a.tau[q] = (M >= a.kprime) ? key_score(outk[a.kprime - 1]) : -INFINITY;
            if (trunc) a.overflow[q] = 1u;
        }
    } else {
        const uint32_t P = next_pow2(M < 2 ? 2 : M);
        for (uint32_t i = M + tid; i < P; i += 256) keys[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc<256>(keys, P, tid);
        for (uint32_t i = tid; i < keep; i += 256) a.best[(uint64_t)q * a.kprime + i] = keys[i];
        if (tid == 0) {
            a.nbest[q] = keep;
            a.tau[q] = (M >= a.kprime) ? key_score(keys[a.kprime - 1]) : -INFINITY;
            if (trunc) a.overflow[q] = 1u;
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
    uint64_t index_base;
    uint32_t nq, n, D, ld, kprime, k, metric;
    float eps_scale;        // cosine: eps; dot: eps = eps_scale * |q| * max|c|
    float max_norm_c;
};

// One workgroup per query: exact reference arithmetic on the k' candidates
// (8 lanes per candidate, see common.h), exact (score desc, row asc) ordering,
// and the guarantee check  e_k > tau + eps  (every row outside the candidate set has
// coarse <= tau, hence exact <= tau + eps < e_k).
template <int DT>
__global__ __launch_bounds__(256) void rescore_kernel(const RescoreArgs a) {
    __shared__ uint64_t ekeys[CAND_CAPS];
    __shared__ uint32_t maxerr;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t nb = a.nbest[q];
    const int grp = tid >> 3, l = tid & 7;
    const Row<DT> qr = make_row<DT>(a.qrows, q, a.ld);
    if (tid == 0) maxerr = 0;
    const uint32_t P = next_pow2(nb < 2 ? 2 : nb);
    for (uint32_t i = nb + tid; i < P; i += 256) ekeys[i] = 0ull;
    __syncthreads();
    for (uint32_t c = grp; c < nb; c += 32) {
        const uint64_t key = a.best[(uint64_t)q * a.kprime + c];
        const uint32_t row = key_row(key);
        const float coarse = key_score(key);
        const float ex = exact_score_group8<DT>(a.metric, qr, make_row<DT>(a.rows, row, a.ld), a.D, l);
        if (l == 0) {
            ekeys[c] = make_key(ex, row);
            float err = fabsf(ex - coarse);
            if (err == err) atomicMax(&maxerr, __float_as_uint(err));
        }
    }
    __syncthreads();
    bitonic_sort_desc<256>(ekeys, P, tid);
    for (uint32_t j = tid; j < a.k; j += 256) {
        uint64_t oi = UINT64_MAX;
        float os = -INFINITY;
        if (j < nb) {
            oi = a.index_base + key_row(ekeys[j]);
            os = key_score(ekeys[j]);
        }
        a.out_idx[(uint64_t)q * a.k + j] = oi;
        a.out_score[(uint64_t)q * a.k + j] = os;
    }
    if (tid == 0) {
        atomicMax(a.stat_maxerr, maxerr);
        bool fb = a.overflow[q] != 0;
        const float tau = a.tau[q];
        if (tau > -INFINITY && nb > 0) {  // candidates were truncated: check the guarantee
            const uint32_t kk = a.k < nb ? a.k : nb;
            const float ek = key_score(ekeys[kk - 1]);
            float eps = a.eps_scale;
            if (a.metric == METRIC_DOT) eps = a.eps_scale * a.norm_q[q] * a.max_norm_c;
            if (!(ek > tau + eps)) fb = true;
            if (nb < a.k) fb = true;
        }
        a.fb_flag[q] = fb ? 1u : 0u;
        if (fb) atomicAdd(a.fb_count, 1u);
    }
}

// Merge G partial top-k lists per query (after the all-gather of per-shard results,
// SURVEY.md §8(e)). Input [g][nq][k]; one workgroup per query; G*k <= 4096.
__global__ __launch_bounds__(256) void merge_topk_kernel(const uint64_t* __restrict__ idx,
                                                         const float* __restrict__ score, uint32_t G,
                                                         uint32_t nq, uint32_t k,
                                                         uint64_t* __restrict__ out_idx,
                                                         float* __restrict__ out_score) {
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
            id = idx[((uint64_t)gi * nq + q) * k + j];
            const float s = score[((uint64_t)gi * nq + q) * k + j];
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
