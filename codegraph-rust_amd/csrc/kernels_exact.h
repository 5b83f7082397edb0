// kernels_exact.h — exact full-scan path: every (query, row) similarity evaluated with
// the reference's own f32 operation order (common.h), then an exact top-k.
// Used (a) for CGV_DTYPE_F32 corpora — the reference's native Vec<f32> layout
// (BASELINE config C1), (b) as the fallback when the coarse pass cannot PROVE its
// candidate set (guarantee check in rescore_kernel), (c) for tiny corpora.
// HBM/L2-bound VALU work; deliberately not reshaped into a GEMM: the serial FMA chain
// per AVX lane is the contract.
#pragma once
#include "common.h"

namespace cgv {

// scores[qi][row] for qi in [0, nql): query index = qlist ? qlist[qi] : qi.
// Block = 256 threads = 32 groups of 8 lanes; group <-> one corpus row per step.
template <int DT>
__global__ __launch_bounds__(256) void exact_scores_kernel(const char* __restrict__ rows,
                                                           const char* __restrict__ qrows,
                                                           const uint32_t* __restrict__ qlist,
                                                           uint32_t nql, uint32_t n, uint32_t D,
                                                           uint32_t ld, int op /* OP_* of common.h */,
                                                           float* __restrict__ scores) {
    const int tid = threadIdx.x;
    const int grp = tid >> 3, l = tid & 7;
    const uint32_t qi = blockIdx.y;
    const uint32_t q = qlist ? qlist[qi] : qi;
    const Row<DT> qr = make_row<DT>(qrows, q, ld);
    for (uint64_t row = (uint64_t)blockIdx.x * 32 + grp; row < n; row += (uint64_t)gridDim.x * 32) {
        const float s = exact_op_group8(op, qr, make_row<DT>(rows, row, ld), D, l);
        if (l == 0) scores[(uint64_t)qi * n + row] = s;
    }
}

// out[q][j] = op(query q, stored row ids[q][j]) for nq x m (query, row) pairs: the per-hit re-score of
// SemanticSearch::search_by_embedding (search.rs:119-129: get_embedding + cosine_similarity per hit) as
// ONE launch over the ids the kNN returned, instead of m host round trips per query. The query is the
// caller's RAW f32 vector (the reference scores the unrounded query against the stored embedding);
// the row is the stored value upcast to f32 (what get_embedding returns). ids[q][j] == UINT64_MAX or
// beyond the index -> 0.0 (a missing embedding scores 0.0, search.rs:207-217). 8 lanes per pair.
template <int DT>
// qsel == NULL: the dense [nq][m] form (pair p belongs to query p / m); else a PAIR LIST: pair p = (query qsel[p], ids[p]) -
// what one shard of a cgv_sharded handle scores (only the pairs whose rows it owns, sharded.hip).
static __global__ __launch_bounds__(256) void score_ids_kernel(const char* __restrict__ rows, const float* __restrict__ queries,
                                                        const uint64_t* __restrict__ ids, const uint32_t* __restrict__ qsel,
                                                        uint64_t npairs, uint32_t m,
                                                        uint64_t n, uint32_t D, uint32_t ld, int op,
                                                        float* __restrict__ out) {
    const int l = threadIdx.x & 7;
    const uint64_t pair = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (pair >= npairs) return;
    const uint32_t q = qsel ? qsel[pair] : (uint32_t)(pair / m);
    const uint64_t id = ids[pair];
    float s = 0.0f;
    if (id < n) {  // uniform within the 8-lane group
        const Row<DT_F32> qr{(const char*)(queries + (uint64_t)q * D), 0u};
        s = exact_op_group8(op, qr, make_row<DT>(rows, id, ld), D, l);
    }
    if (l == 0) out[pair] = s;
}

constexpr uint32_t TOPK_CHUNK = 4096;

// Level-0: chunk of f32 scores -> sorted top-K keys. Level>0: chunk of keys -> top-K keys.
// grid = (nchunks, nql). out[qi][chunk][K].
static __global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ scores,
                                                         const uint64_t* __restrict__ in_keys,
                                                         uint32_t M, uint32_t K,
                                                         uint64_t* __restrict__ out,
                                                         uint32_t* __restrict__ nan_flag) {
    __shared__ uint64_t keys[TOPK_CHUNK];
    const int tid = threadIdx.x;
    const uint32_t chunk = blockIdx.x, qi = blockIdx.y, nchunks = gridDim.x;
    const uint64_t base = (uint64_t)chunk * TOPK_CHUNK;
    for (uint32_t i = tid; i < TOPK_CHUNK; i += 256) {
        const uint64_t e = base + i;
        uint64_t key = 0ull;
        if (e < M) {
            if (scores) {
                const float s = scores[(uint64_t)qi * M + e];
                if (s != s) *nan_flag = 1u;
                key = make_key(s, (uint32_t)e);
            } else {
                key = in_keys[(uint64_t)qi * M + e];
            }
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc<256>(keys, TOPK_CHUNK, tid);
    for (uint32_t j = tid; j < K; j += 256) out[((uint64_t)qi * nchunks + chunk) * K + j] = keys[j];
}

// Final keys [nql][K] -> caller's out arrays at query slot qlist[qi] (or qi).
static __global__ void emit_topk_kernel(const uint64_t* __restrict__ keys, uint32_t K, uint32_t k,
                                 const uint32_t* __restrict__ qlist, uint32_t nql,
                                 IdMap idmap, uint64_t* __restrict__ out_idx,
                                 float* __restrict__ out_score) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nql * k) return;
    const uint32_t qi = i / k, j = i % k;
    const uint32_t q = qlist ? qlist[qi] : qi;
    const uint64_t key = keys[(uint64_t)qi * K + j];
    uint64_t oi = UINT64_MAX;
    float os = -INFINITY;
    if (key != 0ull) {
        oi = map_id(idmap, key_row(key));
        os = key_score(key);
    }
    out_idx[(uint64_t)q * k + j] = oi;
    out_score[(uint64_t)q * k + j] = os;
}

// normalize_avx2 (simd_ops.rs:189-222) / parallel_normalize_vectors (:386-419), in place on a
// flat f32 [n][dim] matrix: nsq = dot_product_avx2(v, v) (8 lane chains + h-sum + tail), zero
// vectors untouched, every element MULTIPLIED by 1/sqrt(nsq). 8 lanes per row.
static __global__ __launch_bounds__(256) void normalize_rows_kernel(float* __restrict__ rows, uint64_t n, uint32_t dim) {
    const int l = threadIdx.x & 7;
    const uint64_t row = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (row >= n) return;
    float* v = rows + row * dim;
    float acc = 0.0f;
    const uint32_t chunks = dim / 8;
    for (uint32_t j = 0; j < chunks; ++j) acc = fmaf(v[8 * j + l], v[8 * j + l], acc);
    acc = group8_hsum(acc);
    float r = 0.0f;
    for (uint32_t i = chunks * 8; i < dim; ++i) r = r + v[i] * v[i];
    const float nsq = __shfl(acc + r, 0, 8);  // lane 0 of the group holds the h-sum
    if (nsq == 0.0f) return;
    const float inv = 1.0f / sqrtf(nsq);
    for (uint32_t i = l; i < dim; i += 8) v[i] = v[i] * inv;
}

// The SCALAR arm of parallel_normalize_vectors (simd_ops.rs:394-403 on an x86_64 host without AVX2, :406-415 on every other
// host): norm_squared = the elements' squares summed IN ORDER by one accumulator (`iter().map(|&x| x * x).sum()`), nothing
// happens unless norm_squared > 0.0 (a NaN sum leaves the row alone), then every element is DIVIDED by sqrt(norm_squared) - the
// AVX2 arm multiplies by the reciprocal, which rounds differently. One lane owns a row's sum (the order is the contract); a
// wave stages 64 rows x 64 columns through LDS so that global reads stay coalesced (row pitch 65 words: conflict-free walks).
static __global__ __launch_bounds__(256) void normalize_rows_scalar_kernel(float* __restrict__ rows, uint64_t n, uint32_t dim) {
    __shared__ float tile[4][64][65];
    __shared__ float nrm[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t row0 = ((uint64_t)blockIdx.x * 4 + w) * 64;
    if (row0 >= n) return;
    const uint32_t nr = (uint32_t)((n - row0 < 64) ? (n - row0) : 64);
    float acc = 0.0f;
    for (uint32_t c0 = 0; c0 < dim; c0 += 64) {
        const uint32_t nc = (dim - c0 < 64) ? (dim - c0) : 64;
        for (uint32_t r = 0; r < nr; ++r)
            if ((uint32_t)lane < nc) tile[w][r][lane] = rows[(row0 + r) * dim + c0 + lane];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);
        if ((uint32_t)lane < nr)
            for (uint32_t c = 0; c < nc; ++c) {
                const float x = tile[w][lane][c];
                acc = acc + x * x;   // (the library is built -ffp-contract=off: a multiply and an add, as in the reference)
            }
        __builtin_amdgcn_wave_barrier();
    }
    nrm[w][lane] = (acc > 0.0f) ? sqrtf(acc) : 0.0f;   // 0: leave the row alone
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t r = 0; r < nr; ++r) {
        const float d = nrm[w][r];
        if (d == 0.0f) continue;
        float* v = rows + (row0 + r) * dim;
        for (uint32_t i = lane; i < dim; i += 64) v[i] = v[i] / d;
    }
}

// Compact the flagged query ids: qlist[0..count) (order irrelevant).
static __global__ void compact_flags_kernel(const uint32_t* __restrict__ flag, uint32_t nq,
                                     uint32_t* __restrict__ qlist, uint32_t* __restrict__ count) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq && flag[q]) qlist[atomicAdd(count, 1u)] = q;
}

}  // namespace cgv
