// kernels_exact_small.h — the exact scan of a FEW queries as ONE kernel (round 5).
//
// The reference's trait-level call is one query (traits.rs:14 search_similar(&self, &[f32], limit); surreal_store.rs:61-85;
// BASELINE config 1: 10k x 384 f32, single query), and on an exact-scan index that call was five dependent kernels behind the
// query conversion - scores of all rows -> chunk reduction (4096 scores -> K keys) -> reduction of the chunks' keys -> results
// -> flags, 38 us of which ~20 are launch boundaries of kernels with a few microseconds of work. Here a workgroup scores
// its rows (the reference's arithmetic, exact_op_group8: 8 lanes per row = the 8 AVX2 lanes), keeps the keys in LDS, reduces
// them to its K best (ranking by counting for <= 256 keys, else the register-resident extraction of kernels_select.h), writes those
// through to memory and counts itself in; the LAST workgroup of a query to arrive merges the partial lists (only the <= K lists whose
// heads are the K largest can hold a result) and writes the results. No score array, no second and
// third launch. Hand-off without fences: agent-scope (sc1, write-through) stores of the partial keys, s_waitcnt vmcnt(0), an
// agent-scope counter, agent-scope loads in the merging workgroup (MI355X_MICROARCH.md: "sc1 payload -> vmcnt(0) -> sc1 flag").
// Same keys, same total order (keys are unique: score bits, then the row) as the staged reduction it replaces.
#pragma once
#include "kernels_exact.h"
#include "kernels_select.h"

namespace cgv {

constexpr uint32_t EXACT_SMALL_ROWS = 4096;   // rows one workgroup scores at most (their keys live in LDS)
constexpr uint32_t EXACT_SMALL_MAX_Q = 64;    // queries per launch (grid.y) at most; larger groups keep the staged reduction (host: exact_search)

struct ExactSmallArgs {
    const char* rows;
    const char* qrows;
    const uint32_t* qlist;   // query slots, or NULL: 0 .. nql - 1
    uint32_t nql, n, D, ld;
    int op;                  // OP_* of common.h
    uint32_t K, k;           // keys kept per level (power of two, <= 64); results per query
    uint64_t* part;          // [nql][G][K] partial keys
    uint32_t* done;          // [EXACT_SMALL_MAX_Q + 1] arrival counters per query + finished queries: zero at launch, zero again at exit
    IdMap idmap;
    uint64_t* out_idx;       // [.][k] at query slot q
    float* out_score;
    uint32_t* nan_flag;
    // end-of-search publication folded into the launch (publish_flags_kernel's job; pub_host == NULL: the caller launches it):
    // the workgroup that finishes the LAST query copies the flag words to the pinned host mirror and clears them
    uint32_t* pub_flags;     // device flag words
    uint32_t* pub_host;      // their pinned, device-mapped mirror
    uint32_t pub_n, pub_done_word, pub_marker;
};

// Top-`keep` of cnt <= 256 keys in LDS (one per thread) by counting: a key's rank is the number of larger keys (keys are unique,
// 0 = empty) - cnt broadcast LDS reads per thread instead of a register sort + `keep` rounds of wave maximum.
__device__ inline void rank_select_256(const uint64_t* keys, uint32_t cnt, uint32_t keep, uint64_t* outk, int tid) {
    const uint64_t v = (uint32_t)tid < cnt ? keys[tid] : 0ull;
    uint32_t rank = 0;
    for (uint32_t e = 0; e < cnt; ++e) rank += keys[e] > v ? 1u : 0u;
    if ((uint32_t)tid < keep) outk[tid] = 0ull;
    __syncthreads();
    if (v != 0ull && rank < keep) outk[rank] = v;
    __syncthreads();
}

template <int DT>
__global__ __launch_bounds__(256) void exact_small_kernel(const ExactSmallArgs a) {
    __shared__ uint64_t wk[EXACT_SMALL_ROWS];
    __shared__ uint64_t part[4 * 64];
    __shared__ uint64_t outk[64];
    __shared__ uint32_t sel[64];
    __shared__ uint32_t last_s;
    const int tid = threadIdx.x;
    const int grp = tid >> 3, l = tid & 7;
    const uint32_t qi = blockIdx.y, G = gridDim.x, b = blockIdx.x, K = a.K;
    const uint32_t q = a.qlist ? a.qlist[qi] : qi;
    const Row<DT> qr = make_row<DT>(a.qrows, q, a.ld);
    // rows b * 32 + grp + it * 32 G: the same trip count for every 8-lane group of the workgroup (a group beyond the corpus
    // leaves an empty key)
    const uint64_t first = (uint64_t)b * 32u;
    const uint32_t nit = first < a.n ? (uint32_t)(((uint64_t)a.n - first + (uint64_t)G * 32u - 1u) / ((uint64_t)G * 32u)) : 0u;
    bool nan = false;
    for (uint32_t it = 0; it < nit; ++it) {
        const uint64_t row = first + (uint64_t)it * G * 32u + (uint32_t)grp;
        uint64_t key = 0ull;
        if (row < a.n) {   // uniform within the 8-lane group
            const float s = exact_op_group8(a.op, qr, make_row<DT>(a.rows, row, a.ld), a.D, l);
            nan = nan || (s != s);
            key = make_key(s, (uint32_t)row);
        }
        if (l == 0) wk[it * 32u + (uint32_t)grp] = key;
    }
    if (nan && l == 0) __hip_atomic_store(a.nan_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t cnt = nit * 32u;
    const uint32_t keep = K < cnt ? K : cnt;
    if (cnt <= 256u)
        rank_select_256(wk, cnt, keep, outk, tid);
    else
        extract_topk(wk, cnt, keep, part, outk, tid);
    uint64_t* mine = a.part + ((uint64_t)qi * G + b) * K;
    for (uint32_t j = tid; j < K; j += 256)
        __hip_atomic_store(mine + j, j < keep ? outk[j] : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores (and the NaN flag) have left
    __syncthreads();
    if (tid == 0) last_s = (__hip_atomic_fetch_add(a.done + qi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u) ? 1u : 0u;
    __syncthreads();
    if (last_s == 0u) return;
    // The last workgroup of query qi: every partial list (sorted, K keys) is in memory. A key of the final top-K sits in a list
    // whose HEAD is among the K largest heads (K heads above a list's head = K keys above everything in it): the G <= 256
    // heads, one per thread, are ranked by counting, and only those <= K lists are read - K x K keys instead of G x K.
    const uint64_t* all = a.part + (uint64_t)qi * G * K;
    const uint64_t hv = (uint32_t)tid < G ? __hip_atomic_load(all + (uint64_t)tid * K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    part[tid] = hv;
    if (tid < 64) sel[tid] = 0xFFFFFFFFu;
    __syncthreads();
    {
        uint32_t rank = 0;
        for (uint32_t e = 0; e < G; ++e) rank += part[e] > hv ? 1u : 0u;
        if (hv != 0ull && rank < K) sel[rank] = (uint32_t)tid;
    }
    __syncthreads();
    const uint32_t nc = K * K;   // <= EXACT_SMALL_ROWS
    for (uint32_t e = tid; e < nc; e += 256) {
        const uint32_t li = sel[e / K];
        wk[e] = li != 0xFFFFFFFFu ? __hip_atomic_load(all + (uint64_t)li * K + e % K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    __syncthreads();
    if (nc <= 256u)
        rank_select_256(wk, nc, K, outk, tid);
    else
        extract_topk(wk, nc, K, part, outk, tid);
    for (uint32_t j = tid; j < a.k; j += 256) {
        const uint64_t key = j < K ? outk[j] : 0ull;
        uint64_t oi = UINT64_MAX;
        float os = -INFINITY;
        if (key != 0ull) {
            oi = map_id(a.idmap, key_row(key));
            os = key_score(key);
        }
        a.out_idx[(uint64_t)q * a.k + j] = oi;
        a.out_score[(uint64_t)q * a.k + j] = os;
    }
    if (tid == 0) __hip_atomic_store(a.done + qi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.pub_host) {   // uniform
        // queries finished: counter nql of `done`. Every workgroup's NaN flag store had left before it counted itself in, and
        // this workgroup has seen all of its query's arrivals, so the one that sees nql - 1 here sees every flag.
        __syncthreads();
        if (tid == 0)
            last_s = (__hip_atomic_fetch_add(a.done + EXACT_SMALL_MAX_Q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.nql - 1u) ? 1u : 0u;
        __syncthreads();
        if (last_s == 0u) return;
        if ((uint32_t)tid < a.pub_n) {
            const uint32_t v = __hip_atomic_load(a.pub_flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.pub_host[tid] = ((uint32_t)tid == a.pub_done_word) ? a.pub_marker : v;
            __hip_atomic_store(a.pub_flags + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) __hip_atomic_store(a.done + EXACT_SMALL_MAX_Q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace cgv
