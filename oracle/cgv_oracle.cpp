// cgv_oracle.cpp — CPU ORACLE for the codegraph-vector brute-force kNN hot path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
// this library. The product (libcgvec_hip.so) never links, imports or calls it.
//
// What this is: a faithful restatement, in C++ with the same AVX2/FMA intrinsics
// and the same operation order, of the arithmetic in the reference's Rust sources
// (paths relative to /root/reference/):
//   O2  crates/codegraph-vector/src/simd_ops.rs    (AVX2 cosine/dot/L2/normalise,
//                                                   rayon brute-force top-k)
//   O1  crates/codegraph-vector/src/optimization.rs:376-418 (search_baseline)
//   RS  crates/codegraph-vector/src/search.rs:113,119-137,519-533,574-592
//       (prefetch rule, exact re-score formula, min-max normalise)
//   HE  crates/codegraph-vector/src/search.rs:178-205,535-541 (hash embedder;
//       identical copies at codegraph-core/src/integration/graph_vector.rs:78-95,
//       codegraph-vector/src/embedding.rs:700-734)
//   Q8  crates/codegraph-vector/src/optimization.rs:63-150,212-224 (int8 scan)
// Every function cites the lines it follows.
//
// Pinning status: the reference is Rust and cannot be built here (no cargo/rustc,
// un-vendored dependency tree), so it cannot be executed to generate vectors.
// The oracle is pinned against every known-answer test the reference's own test
// modules hold for this path (tests/test_oracle_kats.py lists them with
// file:line): simd_ops.rs:429-447, :450-459, :462-472; rag/context_retriever.rs:
// 505-512; rag/result_ranker.rs:598-605; ml/features.rs:481-489. Those KATs are
// loose (1e-6 tolerances, `len()==10`), so beyond them the source arithmetic
// restated here IS the contract.
//
// Build: g++ -O2 -mavx2 -mfma -mf16c -ffp-contract=off -fopenmp (see Makefile).
// -ffp-contract=off matters: Rust never contracts a*b+c, so the scalar formulas
// below must round the product and the sum separately; the AVX2 paths use explicit
// _mm256_fmadd_ps exactly where the reference does.

#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <set>
#include <string>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#endif

extern "C" {

// ---------------------------------------------------------------------------
// O2: SIMDVectorOps
// ---------------------------------------------------------------------------

// simd_ops.rs:227-242 horizontal_sum_avx2:
//   v_perm = permute2f128(v,v,0x01); v_add1 = v + v_perm;
//   hadd(hadd(v_add1)) -> ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7))
static inline float hsum_avx2(__m256 v) {
    __m256 v_perm = _mm256_permute2f128_ps(v, v, 0x01);
    __m256 v_add1 = _mm256_add_ps(v, v_perm);
    __m256 v_hadd1 = _mm256_hadd_ps(v_add1, v_add1);
    __m256 v_hadd2 = _mm256_hadd_ps(v_hadd1, v_hadd1);
    return _mm256_cvtss_f32(v_hadd2);
}

// simd_ops.rs:15-78 cosine_similarity_avx2 (len mismatch is the caller's error;
// len == 0 -> 0.0).
float cgo_cosine_avx2(const float* a, const float* b, size_t len) {
    if (len == 0) return 0.0f;
    __m256 dot = _mm256_setzero_ps();
    __m256 na = _mm256_setzero_ps();
    __m256 nb = _mm256_setzero_ps();
    size_t chunks = len / 8;
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8);
        __m256 vb = _mm256_loadu_ps(b + i * 8);
        dot = _mm256_fmadd_ps(va, vb, dot);
        na = _mm256_fmadd_ps(va, va, na);
        nb = _mm256_fmadd_ps(vb, vb, nb);
    }
    float dp = hsum_avx2(dot);
    float na_sq = hsum_avx2(na);
    float nb_sq = hsum_avx2(nb);
    float dp_r = 0.0f, na_r = 0.0f, nb_r = 0.0f;
    for (size_t i = chunks * 8; i < len; ++i) {
        float va = a[i], vb = b[i];
        dp_r += va * vb;  // separate mul/add rounding (-ffp-contract=off)
        na_r += va * va;
        nb_r += vb * vb;
    }
    float final_dp = dp + dp_r;
    float final_na = na_sq + na_r;
    float final_nb = nb_sq + nb_r;
    float norm_product = sqrtf(final_na * final_nb);
    if (norm_product == 0.0f) return 0.0f;
    return final_dp / norm_product;
}

// simd_ops.rs:257-278 cosine_similarity_scalar.
float cgo_cosine_scalar(const float* a, const float* b, size_t len) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        float va = a[i], vb = b[i];
        dot += va * vb;
        na += va * va;
        nb += vb * vb;
    }
    float norm_product = sqrtf(na * nb);
    if (norm_product == 0.0f) return 0.0f;
    return dot / norm_product;
}

// simd_ops.rs:281-295 adaptive_cosine_similarity: AVX2 iff avx2&&fma (true on
// every x86_64 host this runs on; the build requires -mavx2 -mfma) && len >= 32.
float cgo_cosine_adaptive(const float* a, const float* b, size_t len) {
    if (len >= 32) return cgo_cosine_avx2(a, b, len);
    return cgo_cosine_scalar(a, b, len);
}

// simd_ops.rs:149-183 dot_product_avx2.
float cgo_dot_avx2(const float* a, const float* b, size_t len) {
    if (len == 0) return 0.0f;
    __m256 dot = _mm256_setzero_ps();
    size_t chunks = len / 8;
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8);
        __m256 vb = _mm256_loadu_ps(b + i * 8);
        dot = _mm256_fmadd_ps(va, vb, dot);
    }
    float dp = hsum_avx2(dot);
    float r = 0.0f;
    for (size_t i = chunks * 8; i < len; ++i) r += a[i] * b[i];
    return dp + r;
}

// simd_ops.rs:105-143 l2_distance_avx2.
float cgo_l2_avx2(const float* a, const float* b, size_t len) {
    if (len == 0) return 0.0f;
    __m256 acc = _mm256_setzero_ps();
    size_t chunks = len / 8;
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8);
        __m256 vb = _mm256_loadu_ps(b + i * 8);
        __m256 d = _mm256_sub_ps(va, vb);
        acc = _mm256_fmadd_ps(d, d, acc);
    }
    float s = hsum_avx2(acc);
    float r = 0.0f;
    for (size_t i = chunks * 8; i < len; ++i) {
        float d = a[i] - b[i];
        r += d * d;
    }
    return sqrtf(s + r);
}

// simd_ops.rs:189-222 normalize_avx2 (in place; zero vector untouched;
// multiplies by the reciprocal 1.0/norm, not a divide).
void cgo_normalize_avx2(float* v, size_t len) {
    if (len == 0) return;
    float nsq = cgo_dot_avx2(v, v, len);
    if (nsq == 0.0f) return;
    float norm = sqrtf(nsq);
    float inv = 1.0f / norm;
    __m256 vinv = _mm256_set1_ps(inv);
    size_t chunks = len / 8;
    for (size_t i = 0; i < chunks; ++i) {
        __m256 x = _mm256_loadu_ps(v + i * 8);
        _mm256_storeu_ps(v + i * 8, _mm256_mul_ps(x, vinv));
    }
    for (size_t i = chunks * 8; i < len; ++i) v[i] *= inv;
}

// simd_ops.rs:394-403 / 406-415: the scalar arm of parallel_normalize_vectors (no AVX2 + FMA, or not x86_64):
// `iter().map(|&x| x * x).sum()` = one accumulator in element order, `if norm_squared > 0.0`, `*x /= norm`.
void cgo_normalize_scalar(float* v, size_t len) {
    float nsq = 0.0f;
    for (size_t i = 0; i < len; ++i) nsq = nsq + v[i] * v[i];
    if (nsq > 0.0f) {
        const float norm = sqrtf(nsq);
        for (size_t i = 0; i < len; ++i) v[i] = v[i] / norm;
    }
}

// simd_ops.rs:85-99 batch_cosine_similarity_avx2: one query vs many rows, serial.
void cgo_batch_cosine_avx2(const float* q, const float* const* rows, size_t n, size_t len,
                           float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = cgo_cosine_avx2(q, rows[i], len);
}

// ---------------------------------------------------------------------------
// O2: ParallelVectorOps::parallel_top_k_search  (simd_ops.rs:361-383)
//   scores: rows.par_iter().enumerate().map(adaptive_cosine(q,row).unwrap_or(0.0))
//   sort:   par_sort_unstable_by(|a,b| b.1.partial_cmp(&a.1).unwrap())  (desc)
//   truncate(k)
// rayon's unstable sort leaves tie order unspecified; the oracle fixes
// (score desc, row index asc), a valid outcome of it. NaN makes the reference
// panic (partial_cmp().unwrap()); the oracle returns -1 in that case.
// metric: 0 = cosine (adaptive), 1 = dot product (dot_product_avx2; the reference
// exposes it as a building block, simd_ops.rs:149-183 — BASELINE C4's metric).
// ---------------------------------------------------------------------------

struct Pair {
    float s;
    uint64_t i;
};

static inline bool pair_before(const Pair& x, const Pair& y) {
    if (x.s > y.s) return true;
    if (x.s < y.s) return false;
    return x.i < y.i;
}

// Full parallel sort (not a k-select), every phase threaded - the shape of rayon's par_sort_unstable_by, which the
// reference calls on all N (score, index) pairs (simd_ops.rs:376-380). libstdc++'s parallel mode (MCSTL: parallel
// multiway mergesort, OpenMP) does the splitting, the chunk sorts AND the merges on all threads. pair_before is a strict
// total order (index breaks score ties), so any correct sort yields the same sequence.
static void parallel_sort_pairs(Pair* v, size_t n, int threads) {
    if (threads <= 1 || n < 4096) {
        std::sort(v, v + n, pair_before);
        return;
    }
#ifdef _OPENMP
    __gnu_parallel::sort(v, v + n, pair_before, __gnu_parallel::multiway_mergesort_tag(threads));
#else
    std::sort(v, v + n, pair_before);
#endif
}

float cgo_search_cosine(const float* a, const float* b, size_t len);

// The (score, index) buffer of the calling thread, kept between calls: the reference allocates a fresh Vec per query
// (`collect()`, simd_ops.rs:366-374), which its allocator serves from a recycled block; re-faulting 16 MB of fresh
// pages per query (what a new std::vector does here) is not part of the algorithm being timed. Pages are first touched
// by the thread that fills them (same static schedule as the scoring loop).
static thread_local Pair* tl_pairs = nullptr;
static thread_local size_t tl_pairs_cap = 0;
static thread_local double tl_score_ms = 0.0, tl_sort_ms = 0.0;

static double now_ms() {
#ifdef _OPENMP
    return omp_get_wtime() * 1e3;
#else
    return 0.0;
#endif
}

// timing of the calling thread's last cgo_parallel_top_k: scoring pass / sort (milliseconds)
void cgo_last_timing(double* score_ms, double* sort_ms) {
    if (score_ms) *score_ms = tl_score_ms;
    if (sort_ms) *sort_ms = tl_sort_ms;
}

// metric: 0 = adaptive cosine (an AVX2 host: simd_ops.rs:281-290), 1 = dot product, 2 = search.rs' sequential cosine,
// 3 = cosine_similarity_scalar for every length (adaptive_cosine_similarity on a host without AVX2 + FMA or not
// x86_64 at all, simd_ops.rs:291-294)
int cgo_parallel_top_k(const float* query, const float* const* rows, uint64_t n, uint64_t dim,
                       uint64_t k, int metric, int threads, uint64_t* out_idx, float* out_score) {
    if (threads <= 0) {
#ifdef _OPENMP
        threads = omp_get_max_threads();
#else
        threads = 1;
#endif
    }
    if (tl_pairs_cap < n) {
        // raw storage: nothing touches the pages on this thread; the scoring loop below writes every element
        free(tl_pairs);
        tl_pairs = (Pair*)malloc(sizeof(Pair) * (size_t)n);
        tl_pairs_cap = tl_pairs ? (size_t)n : 0;
        if (!tl_pairs) return -2;
    }
    Pair* sims = tl_pairs;
    int has_nan = 0;
    const double t0 = now_ms();
#pragma omp parallel for num_threads(threads) schedule(static) reduction(| : has_nan)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float s = metric == 1   ? cgo_dot_avx2(query, rows[i], dim)
                  : metric == 2 ? cgo_search_cosine(query, rows[i], dim)  // search.rs:519-533 == indexer.rs:2965-2979
                  : metric == 3 ? cgo_cosine_scalar(query, rows[i], dim)  // simd_ops.rs:257-278 for every length
                                : cgo_cosine_adaptive(query, rows[i], dim);
        if (s != s) has_nan |= 1;
        sims[i].s = s;
        sims[i].i = (uint64_t)i;
    }
    const double t1 = now_ms();
    tl_score_ms = t1 - t0;
    tl_sort_ms = 0.0;
    if (has_nan) return -1;  // reference panics (simd_ops.rs:379)
    parallel_sort_pairs(sims, (size_t)n, threads);
    tl_sort_ms = now_ms() - t1;
    uint64_t m = k < n ? k : n;
    for (uint64_t j = 0; j < m; ++j) {
        out_idx[j] = sims[j].i;
        out_score[j] = sims[j].s;
    }
    for (uint64_t j = m; j < k; ++j) {
        out_idx[j] = UINT64_MAX;
        out_score[j] = -std::numeric_limits<float>::infinity();
    }
    return (int)m;
}

// Same, over a flat row-major matrix (convenience for tests; identical results).
int cgo_parallel_top_k_flat(const float* query, const float* flat, uint64_t n, uint64_t dim,
                            uint64_t k, int metric, int threads, uint64_t* out_idx,
                            float* out_score) {
    std::vector<const float*> rows(n);
    for (uint64_t i = 0; i < n; ++i) rows[i] = flat + i * dim;
    return cgo_parallel_top_k(query, rows.data(), n, dim, k, metric, threads, out_idx, out_score);
}

// The CPU-baseline layout: N separately heap-allocated rows, like the reference's
// &[Vec<f32>] (simd_ops.rs:363). Returns an opaque handle.
struct RowSet {
    std::vector<float*> rows;
    uint64_t dim;
};
void* cgo_rowset_create(const float* flat, uint64_t n, uint64_t dim) {
    RowSet* rs = new RowSet();
    rs->dim = dim;
    rs->rows.resize(n);
    // every row is its own heap block (the reference's Vec<f32> per embedding), allocated AND first written by the thread
    // that scans it in cgo_parallel_top_k (same static schedule): on a multi-socket host the rows are then spread over
    // the NUMA nodes like the threads, instead of all sitting on the node of the one thread that loaded them
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        rs->rows[i] = (float*)malloc(sizeof(float) * (dim ? dim : 1));
        memcpy(rs->rows[i], flat + (uint64_t)i * dim, sizeof(float) * dim);
    }
    return rs;
}
void cgo_rowset_destroy(void* h) {
    RowSet* rs = (RowSet*)h;
    if (!rs) return;
    for (float* p : rs->rows) free(p);
    delete rs;
}
int cgo_rowset_top_k(void* h, const float* query, uint64_t k, int metric, int threads,
                     uint64_t* out_idx, float* out_score) {
    RowSet* rs = (RowSet*)h;
    return cgo_parallel_top_k(query, rs->rows.data(), rs->rows.size(), rs->dim, k, metric,
                              threads, out_idx, out_score);
}

// ---------------------------------------------------------------------------
// RS: search.rs re-score arithmetic
// ---------------------------------------------------------------------------

// search.rs:519-533 cosine_similarity: sequential iterator sums (no FMA),
// dot / (sqrt(na) * sqrt(nb)); zero norm -> 0.0. (The length-mismatch -> 0.0 branch
// is the caller's concern here.) Also the formula of InMemoryVectorStore's cosine
// (codegraph-core/src/integration/graph_vector.rs:504-516) and of the KAT helpers
// in rag/context_retriever.rs, rag/result_ranker.rs, ml/features.rs.
float cgo_search_cosine(const float* a, const float* b, size_t len) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (size_t i = 0; i < len; ++i) dot += a[i] * b[i];
    for (size_t i = 0; i < len; ++i) na += a[i] * a[i];
    for (size_t i = 0; i < len; ++i) nb += b[i] * b[i];
    float norm_a = sqrtf(na), norm_b = sqrtf(nb);
    if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
    return dot / (norm_a * norm_b);
}

// search.rs:113 prefetch_k = max(3*limit (saturating), limit + 10).
uint64_t cgo_prefetch_k(uint64_t limit) {
    uint64_t t = limit > UINT64_MAX / 3 ? UINT64_MAX : limit * 3;
    uint64_t u = limit + 10;
    return t > u ? t : u;
}

// search.rs:574-592 normalize_scores: min-max to [0,1], range floored at 1e-12.
void cgo_normalize_scores(float* s, size_t n) {
    if (n == 0) return;
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        if (s[i] < mn) mn = s[i];
        if (s[i] > mx) mx = s[i];
    }
    float range = mx - mn;
    if (!(range > 1e-12f)) range = 1e-12f;  // f32::max(1e-12): NaN range -> 1e-12
    for (size_t i = 0; i < n; ++i) s[i] = (s[i] - mn) / range;
}

// search.rs:119-138: exact re-score of candidate rows, STABLE sort desc
// (partial_cmp -> Equal on NaN), truncate(limit). cand[] is the order
// search_similar returned. Writes the surviving candidate positions and scores
// (scores NOT yet min-max normalised). Returns count.
uint64_t cgo_rescore_sort(const float* query, const float* const* cand_rows, uint64_t ncand,
                          uint64_t dim, uint64_t limit, uint64_t* out_pos, float* out_score) {
    std::vector<std::pair<float, uint64_t>> r(ncand);
    for (uint64_t j = 0; j < ncand; ++j)
        r[j] = {cand_rows[j] ? cgo_search_cosine(query, cand_rows[j], dim) : 0.0f, j};
    std::stable_sort(r.begin(), r.end(),
                     [](const std::pair<float, uint64_t>& x, const std::pair<float, uint64_t>& y) {
                         return x.first > y.first;  // desc; NaN compares "equal"
                     });
    uint64_t m = limit < ncand ? limit : ncand;
    for (uint64_t j = 0; j < m; ++j) {
        out_pos[j] = r[j].second;
        out_score[j] = r[j].first;
    }
    return m;
}

// ---------------------------------------------------------------------------
// O1: ModelOptimizer::search_baseline / cosine_distance (optimization.rs:376-418)
//   distance = 1 - dot/(sqrt(na)*sqrt(nb)); +inf on zero norm;
//   STABLE ascending sort (ties keep input order = index asc); take(limit).
// ---------------------------------------------------------------------------
float cgo_cosine_distance(const float* a, const float* b, size_t len) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (size_t i = 0; i < len; ++i) dot += a[i] * b[i];
    for (size_t i = 0; i < len; ++i) na += a[i] * a[i];
    for (size_t i = 0; i < len; ++i) nb += b[i] * b[i];
    float norm_a = sqrtf(na), norm_b = sqrtf(nb);
    if (norm_a == 0.0f || norm_b == 0.0f) return INFINITY;
    return 1.0f - (dot / (norm_a * norm_b));
}

uint64_t cgo_search_baseline(const float* query, const float* flat, uint64_t n, uint64_t dim,
                             uint64_t limit, uint64_t* out_idx, float* out_dist) {
    if (n == 0) return 0;
    std::vector<std::pair<float, uint64_t>> d(n);
    for (uint64_t i = 0; i < n; ++i) d[i] = {cgo_cosine_distance(query, flat + i * dim, dim), i};
    std::stable_sort(d.begin(), d.end(),
                     [](const std::pair<float, uint64_t>& x, const std::pair<float, uint64_t>& y) {
                         return x.first < y.first;
                     });
    uint64_t m = limit < n ? limit : n;
    for (uint64_t j = 0; j < m; ++j) {
        out_idx[j] = d[j].second;
        if (out_dist) out_dist[j] = d[j].first;
    }
    return m;
}

// ---------------------------------------------------------------------------
// HE: deterministic hash embedder (search.rs:178-205 + simple_hash :535-541).
//   djb2: h = h*33 + byte (u32 wrap, seed 5381)
//   LCG : s = s*1103515245 + 12345 (u32 wrap) per component
//   x   = ((s as f32 / u32::MAX as f32) - 0.5) * 2.0 ; then divide by sqrt(sum x^2)
// `s as f32` rounds to nearest-even; `u32::MAX as f32` == 4294967296.0f.
// ---------------------------------------------------------------------------
uint32_t cgo_simple_hash(const uint8_t* text, size_t len) {
    uint32_t h = 5381u;
    for (size_t i = 0; i < len; ++i) h = h * 33u + (uint32_t)text[i];
    return h;
}

void cgo_hash_embed(const uint8_t* text, size_t len, uint32_t dim, float* out) {
    uint32_t s = cgo_simple_hash(text, len);
    const float umax = (float)UINT32_MAX;  // 4294967296.0f
    for (uint32_t i = 0; i < dim; ++i) {
        s = s * 1103515245u + 12345u;
        out[i] = (((float)s / umax) - 0.5f) * 2.0f;
    }
    float nsq = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) nsq += out[i] * out[i];
    float norm = sqrtf(nsq);
    if (norm > 0.0f)
        for (uint32_t i = 0; i < dim; ++i) out[i] /= norm;
}

// ---------------------------------------------------------------------------
// Q8: int8 path (optimization.rs:212-224 quantize_unit_range_symmetric,
// :226-283 quantize_batch (u8 = q + 128), :63-150 search_optimized).
// f32::round is round-half-away-from-zero == roundf.
// ---------------------------------------------------------------------------
void cgo_quantize_u8(const float* v, size_t len, uint8_t* out) {
    for (size_t i = 0; i < len; ++i) {
        float c = v[i] < -1.0f ? -1.0f : (v[i] > 1.0f ? 1.0f : v[i]);
        int q = (c != c) ? 0 : (int)roundf(c * 127.0f);  // Rust: NaN.round() as i32 == 0
        if (q < -127) q = -127;
        if (q > 127) q = 127;
        out[i] = (uint8_t)(q + 128);
    }
}

// optimization.rs:248-262 + :338-343: 4-bit arm of quantize_batch
void cgo_quantize_u4(const float* v, uint64_t n, uint64_t dim, uint8_t* out) {
    const uint64_t half = (dim + 1) / 2;
    for (uint64_t r = 0; r < n; ++r)
        for (uint64_t j = 0; j < dim; j += 2) {
            uint8_t q[2] = {0, 0};
            for (int t = 0; t < 2 && j + t < dim; ++t) {
                float x = v[r * dim + j + t];
                float c = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
                float nrm = (c + 1.0f) / 2.0f;
                int qi = (nrm != nrm) ? 0 : (int)roundf(nrm * 15.0f);
                q[t] = (uint8_t)(qi < 0 ? 0 : (qi > 15 ? 15 : qi));
            }
            out[r * half + j / 2] = (uint8_t)((q[0] & 0x0F) | ((q[1] & 0x0F) << 4));
        }
}

uint64_t cgo_search_optimized_u8(const float* query, const uint8_t* data, uint64_t n, uint64_t dim,
                                 uint64_t limit_in, uint64_t* out_idx) {
    uint64_t limit = limit_in < 1 ? 1 : limit_in;
    if (dim == 0 || n == 0) return 0;
    std::vector<int8_t> qq(dim);
    for (uint64_t i = 0; i < dim; ++i) {
        float c = query[i] < -1.0f ? -1.0f : (query[i] > 1.0f ? 1.0f : query[i]);
        int q = (c != c) ? 0 : (int)roundf(c * 127.0f);
        if (q < -127) q = -127;
        if (q > 127) q = 127;
        qq[i] = (int8_t)q;
    }
    float nq = 0.0f;
    for (uint64_t i = 0; i < dim; ++i) nq += (float)qq[i] * (float)qq[i];
    nq = sqrtf(nq);
    if (nq == 0.0f) return 0;
    std::vector<std::pair<uint64_t, float>> best;
    auto asc = [](const std::pair<uint64_t, float>& x, const std::pair<uint64_t, float>& y) {
        return x.second < y.second;
    };
    for (uint64_t idx = 0; idx < n; ++idx) {
        const uint8_t* row = data + idx * dim;
        int32_t dot = 0, nv = 0;
        for (uint64_t j = 0; j < dim; ++j) {
            int32_t v = (int32_t)row[j] - 128;
            dot += v * (int32_t)qq[j];
            nv += v * v;
        }
        if (nv == 0) continue;
        float score = (float)dot / (nq * sqrtf((float)nv));
        if (best.size() < limit) {
            best.push_back({idx, score});
            if (best.size() == limit) std::stable_sort(best.begin(), best.end(), asc);
        } else if (score > best[0].second) {
            best[0] = {idx, score};
            std::stable_sort(best.begin(), best.end(), asc);
        }
    }
    std::stable_sort(best.begin(), best.end(),
                     [](const std::pair<uint64_t, float>& x, const std::pair<uint64_t, float>& y) {
                         return x.second > y.second;
                     });
    for (size_t j = 0; j < best.size(); ++j) out_idx[j] = best[j].first;
    return best.size();
}

// ---------------------------------------------------------------------------
// Storage-dtype rounding (the reference has f32 only; SURVEY.md §8(c): the oracle
// for bf16/fp16/fp8 configs is the O2 arithmetic on the rounded-then-upcast
// values). All conversions are round-to-nearest-even.
// ---------------------------------------------------------------------------
static inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

uint16_t cgo_f32_to_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}
float cgo_bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }

uint16_t cgo_f32_to_f16(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }
float cgo_f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }

// OCP FP8 E4M3FN (gfx950's format; bias 7, no inf, NaN = S.1111.111, max 448),
// RNE, saturating to +-448 for finite inputs beyond range; NaN -> 0x7f|sign.
uint8_t cgo_f32_to_e4m3(float f) {
    uint32_t u = f2u(f);
    uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7fu);  // NaN
    float af = u2f(a);
    if (af >= 464.0f) return (uint8_t)(sign | 0x7eu);     // >= midpoint(448,512) saturates; inf too
    if (af < 0.0009765625f) return sign;                  // < 2^-10 = half of min subnormal 2^-9
    // scale so that the quantum is an integer: subnormal quantum 2^-9, normal 2^(e-3)
    int e;
    (void)frexpf(af, &e);  // af = m * 2^e, m in [0.5,1)
    int exp = e - 1;       // af in [2^exp, 2^(exp+1))
    if (exp < -6) exp = -6;
    float q = ldexpf(1.0f, exp - 3);     // quantum
    float r = af / q;                    // exact (power of two)
    float rr = nearbyintf(r);            // RNE under default rounding mode
    float v = rr * q;
    if (v > 448.0f) v = 448.0f;
    if (v == 0.0f) return sign;
    // encode v
    int e2;
    float m = frexpf(v, &e2);  // v = m*2^e2, m in [0.5,1)
    int ex = e2 - 1;
    if (ex < -6) {  // subnormal: v = k * 2^-9, k in 1..7
        int k = (int)(v * 512.0f);
        return (uint8_t)(sign | (uint8_t)k);
    }
    int mant = (int)((m * 2.0f - 1.0f) * 8.0f);  // exact
    return (uint8_t)(sign | (uint8_t)((ex + 7) << 3) | (uint8_t)mant);
}
float cgo_e4m3_to_f32(uint8_t b) {
    int sign = b & 0x80;
    int ex = (b >> 3) & 0xf;
    int mant = b & 7;
    float v;
    if (ex == 0xf && mant == 7)
        v = NAN;
    else if (ex == 0)
        v = ldexpf((float)mant, -9);
    else
        v = ldexpf(1.0f + (float)mant / 8.0f, ex - 7);
    return sign ? -v : v;
}

// FP8 storage uses a per-row power-of-two scale (SURVEY.md §7 "fp8 dynamic range": unit-norm
// 768-d components sit at e4m3's subnormal edge): stored byte = e4m3(x * 2^e), value = byte * 2^-e,
// with e the largest exponent that keeps amax * 2^e <= 448. Power-of-two scaling is exact, so the
// dequantised value is exactly representable in f32 and cosine is unaffected by the scale.
int cgo_fp8_row_exponent(const float* row, size_t d) {
    float amax = 0.0f;
    for (size_t i = 0; i < d; ++i) {
        float a = fabsf(row[i]);
        if (a > amax) amax = a;  // NaN never compares greater
    }
    uint32_t u = f2u(amax);
    int bexp = (int)((u >> 23) & 0xff);
    if (amax == 0.0f || bexp == 0 || bexp == 0xff) return 0;  // zero / subnormal / inf: no scaling
    int x = bexp - 127;                                        // amax = m * 2^x, m in [1,2)
    uint32_t mant = u & 0x7fffffu;
    int e = (mant <= 0x600000u) ? 8 - x : 7 - x;               // m <= 1.75 -> fits below 448 = 1.75 * 2^8
    if (e > 100) e = 100;
    if (e < -100) e = -100;
    return e;
}
// descale != 0: the VALUES the storage holds (code * 2^-e; what get_row returns).
// descale == 0: the e4m3 CODES as f32 (x * 2^e rounded) - the domain scores are defined in:
// cosine is invariant under the per-row power-of-two scale, and in the code domain every
// partial sum stays in f32's normal range whatever the magnitude of the original row.
void cgo_round_trip_fp8_rows(const float* in, size_t n, size_t d, float* out, int descale) {
    for (size_t r = 0; r < n; ++r) {
        const int e = cgo_fp8_row_exponent(in + r * d, d);
        for (size_t i = 0; i < d; ++i) {
            const float c = cgo_e4m3_to_f32(cgo_f32_to_e4m3(ldexpf(in[r * d + i], e)));
            out[r * d + i] = descale ? ldexpf(c, -e) : c;
        }
    }
}

// Round an f32 array to `dtype` and back (0=f32 identity, 1=bf16, 2=fp16, 3=fp8e4m3 UNSCALED).
void cgo_round_trip(const float* in, size_t n, int dtype, float* out) {
    for (size_t i = 0; i < n; ++i) {
        switch (dtype) {
            case 1: out[i] = cgo_bf16_to_f32(cgo_f32_to_bf16(in[i])); break;
            case 2: out[i] = cgo_f16_to_f32(cgo_f32_to_f16(in[i])); break;
            case 3: out[i] = cgo_e4m3_to_f32(cgo_f32_to_e4m3(in[i])); break;
            default: out[i] = in[i];
        }
    }
}

// Merge of per-shard partial top-k lists (no reference
// counterpart): G lists of k (score, global idx), select top-k by
// (score desc, idx asc). Padding entries (idx == UINT64_MAX) sort last.
void cgo_merge_topk(const uint64_t* idx, const float* score, uint64_t g, uint64_t k,
                    uint64_t* out_idx, float* out_score) {
    std::vector<Pair> all;
    all.reserve(g * k);
    for (uint64_t j = 0; j < g * k; ++j)
        if (idx[j] != UINT64_MAX) all.push_back({score[j], idx[j]});
    std::sort(all.begin(), all.end(), pair_before);
    for (uint64_t j = 0; j < k; ++j) {
        if (j < all.size()) {
            out_idx[j] = all[j].i;
            out_score[j] = all[j].s;
        } else {
            out_idx[j] = UINT64_MAX;
            out_score[j] = -std::numeric_limits<float>::infinity();
        }
    }
}

int cgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---------------------------------------------------------------------------
// Symbol resolver, embedding phase: crates/codegraph-mcp/src/indexer.rs:2790-2843
// (ai_semantic_match_sync PHASE 2) with char_trigrams / jaccard (:2901-2932) and
// cosine_similarity_static (:2965-2979, same arithmetic as cgo_search_cosine). Literal
// per-target loop over all known symbols; the reference's HashMap order is unspecified, the
// restatement visits symbols in index order and keeps the first strictly greater similarity.
// Lower-casing is ASCII-only (test names are ASCII identifiers).
// ---------------------------------------------------------------------------
static std::string o_lower(const char* s) {
    std::string r(s ? s : "");
    for (size_t i = 0; i < r.size(); ++i)
        if (r[i] >= 'A' && r[i] <= 'Z') r[i] = (char)(r[i] + 32);
    return r;
}
static std::set<std::string> o_trigrams(const std::string& s) {
    std::vector<std::string> chars;  // s.chars()
    for (size_t i = 0; i < s.size();) {
        size_t j = i + 1;
        while (j < s.size() && ((unsigned char)s[j] & 0xC0) == 0x80) ++j;
        chars.push_back(s.substr(i, j - i));
        i = j;
    }
    std::set<std::string> out;
    if (chars.size() < 3) {
        if (!s.empty()) out.insert(s);
        return out;
    }
    for (size_t i = 0; i + 3 <= chars.size(); ++i) out.insert(chars[i] + chars[i + 1] + chars[i + 2]);
    return out;
}
static float o_jaccard(const std::set<std::string>& a, const std::set<std::string>& b) {
    if (a.empty() || b.empty()) return 0.0f;
    float inter = 0.0f;
    for (std::set<std::string>::const_iterator it = a.begin(); it != a.end(); ++it)
        if (b.count(*it)) inter += 1.0f;
    float uni = (float)(a.size() + b.size()) - inter;
    return uni == 0.0f ? 0.0f : inter / uni;
}
float cgo_trigram_jaccard(const char* a, const char* b) { return o_jaccard(o_trigrams(o_lower(a)), o_trigrams(o_lower(b))); }

int64_t cgo_symbol_match_phase2(const char* target, const float* target_emb, uint64_t n, const char* const* names,
                                const float* embs, uint64_t dim, float threshold, float* out_score) {
    const std::string tl = o_lower(target);
    const std::set<std::string> tt = o_trigrams(tl);
    int64_t best = -1;
    float best_s = 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        const std::string nl = o_lower(names[i]);
        const float a = (float)tl.size(), b = (float)nl.size();
        const float r1 = a / b, r2 = b / a;
        const float ratio = r1 < r2 ? r1 : r2;
        if (!(ratio >= 0.5f)) continue;                    // len_ok (:2808-2815)
        if (!(o_jaccard(tt, o_trigrams(nl)) >= 0.2f)) continue;  // overlap >= 0.2 (:2817-2819)
        const float sim = cgo_search_cosine(target_emb, embs + i * dim, dim);
        if (sim > threshold) {                            // :2828
            if (best < 0 || sim > best_s) {               // :2829-2835
                best = (int64_t)i;
                best_s = sim;
            }
        }
    }
    if (out_score) *out_score = best >= 0 ? best_s : 0.0f;
    return best;
}

// ---------------------------------------------------------------------------
// ScalarQuantizer / ProductQuantizer: crates/codegraph-vector/src/persistent.rs:116-477,
// literal single-thread restatements (training order, tie rules and f32 rounding as written).
// ---------------------------------------------------------------------------
static inline float rs_min(float a, float b) { return (b != b) ? a : ((a != a) ? b : (b < a ? b : a)); }  // f32::min
static inline float rs_max(float a, float b) { return (b != b) ? a : ((a != a) ? b : (b > a ? b : a)); }

void cgo_sq_train(const float* v, uint64_t n, uint64_t dim, uint32_t nbits, int uniform, float* scales, float* biases) {
    for (uint64_t d = 0; d < dim; ++d) {
        scales[d] = 1.0f;
        biases[d] = 0.0f;
    }
    const float levels = (float)(1 << nbits);
    if (uniform) {  // :366-385
        float gmin = INFINITY, gmax = -INFINITY;
        for (uint64_t r = 0; r < n; ++r)
            for (uint64_t d = 0; d < dim; ++d) {
                gmin = rs_min(gmin, v[r * dim + d]);
                gmax = rs_max(gmax, v[r * dim + d]);
            }
        const float scale = levels / (gmax - gmin);
        for (uint64_t d = 0; d < dim; ++d) {
            scales[d] = scale;
            biases[d] = gmin;
        }
    } else {  // :386-403
        for (uint64_t d = 0; d < dim; ++d) {
            float lo = INFINITY, hi = -INFINITY;
            for (uint64_t r = 0; r < n; ++r) {
                lo = rs_min(lo, v[r * dim + d]);
                hi = rs_max(hi, v[r * dim + d]);
            }
            const float range = hi - lo;
            if (range > 0.0f) {
                scales[d] = levels / range;
                biases[d] = lo;
            }
        }
    }
}

void cgo_sq_encode(const float* v, uint64_t n, uint64_t dim, uint32_t nbits, const float* scales, const float* biases,
                   uint8_t* out) {  // :410-436
    const uint32_t max_val = (1u << nbits) - 1u;
    const uint32_t bpv = nbits == 8 ? 1 : (nbits == 16 ? 2 : 4);
    for (uint64_t i = 0; i < n * dim; ++i) {
        const uint64_t d = i % dim;
        const float normalized = (v[i] - biases[d]) * scales[d];
        float c = rs_max(normalized, 0.0f);
        c = rs_min(c, (float)max_val);
        const uint32_t q = (uint32_t)c;
        for (uint32_t b = 0; b < bpv; ++b) out[i * bpv + b] = (uint8_t)(q >> (8 * b));
    }
}

void cgo_sq_decode(const uint8_t* codes, uint64_t n, uint64_t dim, uint32_t nbits, const float* scales,
                   const float* biases, float* out) {  // :438-476
    const uint32_t bpv = nbits == 8 ? 1 : (nbits == 16 ? 2 : 4);
    for (uint64_t i = 0; i < n * dim; ++i) {
        uint32_t q = 0;
        for (uint32_t b = 0; b < bpv; ++b) q |= (uint32_t)codes[i * bpv + b] << (8 * b);
        out[i] = (float)q / scales[i % dim] + biases[i % dim];
    }
}

static float o_pq_dist(const float* a, const float* b, uint64_t dsub) {  // :320-328
    float s = 0.0f;
    for (uint64_t i = 0; i < dsub; ++i) {
        const float d = a[i] - b[i];
        s = s + d * d;
    }
    return sqrtf(s);
}

// centroids out: [m][ksub][dsub]
void cgo_pq_train(const float* v, uint64_t n, uint64_t dim, uint64_t m, uint32_t nbits, float* cent) {
    const uint64_t dsub = dim / m, ksub = 1ull << nbits;
    std::vector<uint64_t> assign(n);
    std::vector<float> nc(dsub);
    for (uint64_t sub = 0; sub < m; ++sub) {
        float* c = cent + sub * ksub * dsub;
        for (uint64_t i = 0; i < ksub; ++i)  // :255-258
            memcpy(c + i * dsub, v + (i % n) * dim + sub * dsub, dsub * 4);
        for (int it = 0; it < 50; ++it) {  // :261
            bool changed = false;
            for (uint64_t r = 0; r < n; ++r) {
                uint64_t best = 0;
                float bd = INFINITY;
                for (uint64_t k = 0; k < ksub; ++k) {
                    const float d = o_pq_dist(v + r * dim + sub * dsub, c + k * dsub, dsub);
                    if (d < bd) {
                        bd = d;
                        best = k;
                    }
                }
                if (best != 0) changed = true;  // assignments were reset to 0 (:264, :279-281)
                assign[r] = best;
            }
            for (uint64_t k = 0; k < ksub; ++k) {  // :286-309
                uint64_t cnt = 0;
                for (uint64_t d = 0; d < dsub; ++d) nc[d] = 0.0f;
                for (uint64_t r = 0; r < n; ++r)
                    if (assign[r] == k) {
                        for (uint64_t d = 0; d < dsub; ++d) nc[d] += v[r * dim + sub * dsub + d];
                        ++cnt;
                    }
                if (cnt)
                    for (uint64_t d = 0; d < dsub; ++d) c[k * dsub + d] = nc[d] / (float)cnt;
            }
            if (!changed) break;
        }
    }
}

void cgo_pq_encode(const float* v, uint64_t n, uint64_t dim, uint64_t m, uint32_t nbits, const float* cent, uint8_t* codes) {
    const uint64_t dsub = dim / m, ksub = 1ull << nbits;
    for (uint64_t r = 0; r < n; ++r)
        for (uint64_t sub = 0; sub < m; ++sub) {  // :196-214
            uint64_t best = 0;
            float bd = INFINITY;
            for (uint64_t k = 0; k < ksub; ++k) {
                const float d = o_pq_dist(v + r * dim + sub * dsub, cent + (sub * ksub + k) * dsub, dsub);
                if (d < bd) {
                    bd = d;
                    best = k;
                }
            }
            codes[r * m + sub] = (uint8_t)best;
        }
}


// ---------------------------------------------------------------------------
// SURVEY.md section 8(d) synthetic inputs: counter-based generator keyed (seed, row, col) -> N(0, 1) f32 -> L2-normalised in
// f32. Not reference code (the reference has no generator): the CPU side of the bench's data contract, restated here
// independently of codegraph-rust_amd/csrc/synth.hip from the contract in that file's header, so that the CPU baseline and
// the parity checks can produce any row of any corpus without copying it from the device. Philox4x32-10 (Salmon et al.,
// SC'11; pinned by the Random123 known-answer vectors in tests/test_synth.py), Box-Muller with polynomial ln / sin / cos
// (libm's differ between hosts and devices in the last bit), 64 strided partial sums of squares met by a xor butterfly.
static void cgo_philox4x32_10_impl(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void cgo_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) { cgo_philox4x32_10_impl(ctr4, key2, out4); }

static float cgo_synth_ln(float u) {
    uint32_t b;
    memcpy(&b, &u, 4);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    float p = 0.0909090936f;
    p = p * t2 + 0.111111112f;
    p = p * t2 + 0.142857149f;
    p = p * t2 + 0.2f;
    p = p * t2 + 0.333333343f;
    p = p * t2 + 1.0f;
    return (float)e * 0.693147182f + (2.0f * t) * p;
}
static void cgo_synth_pair(uint32_t xa, uint32_t xb, float* z0, float* z1) {
    const float u = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float v = (float)(xb >> 8) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * cgo_synth_ln(u));
    const float a4 = v * 4.0f;
    const int q = (int)a4;
    const float a = (a4 - (float)q) * 1.57079637f;
    const float a2 = a * a;
    float ps = -2.50521084e-08f;
    ps = ps * a2 + 2.75573188e-06f;
    ps = ps * a2 - 1.98412701e-04f;
    ps = ps * a2 + 8.33333377e-03f;
    ps = ps * a2 - 0.166666672f;
    ps = ps * a2 + 1.0f;
    const float sn = a * ps;
    float pc = 2.08767570e-09f;
    pc = pc * a2 - 2.75573188e-07f;
    pc = pc * a2 + 2.48015876e-05f;
    pc = pc * a2 - 1.38888892e-03f;
    pc = pc * a2 + 4.16666679e-02f;
    pc = pc * a2 - 0.5f;
    pc = pc * a2 + 1.0f;
    float c, s;
    switch (q & 3) {
        case 0: c = pc; s = sn; break;
        case 1: c = -sn; s = pc; break;
        case 2: c = -pc; s = -sn; break;
        default: c = sn; s = -pc; break;
    }
    *z0 = rad * c;
    *z1 = rad * s;
}
// rows [row0, row0 + nrows) of the stream `seed`, dim columns, into out[nrows][dim]; normalise = 0: the N(0, 1) values
void cgo_synth_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, int normalise, float* out) {
    const uint32_t nblk = (dim + 3) / 4;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nrows; ++i) {
        const uint64_t r = row0 + (uint64_t)i;
        float* o = out + (size_t)i * dim;
        float part[64];
        for (int l = 0; l < 64; ++l) part[l] = 0.0f;
        for (uint32_t b = 0; b < nblk; ++b) {
            const uint32_t ctr[4] = {b, (uint32_t)r, (uint32_t)(r >> 32), 0u};
            const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t x[4];
            cgo_philox4x32_10_impl(ctr, key, x);
            float z[4];
            cgo_synth_pair(x[0], x[1], &z[0], &z[1]);
            cgo_synth_pair(x[2], x[3], &z[2], &z[3]);
            for (int j = 0; j < 4; ++j)
                if (4 * b + j < dim) {
                    o[4 * b + j] = z[j];
                    part[b & 63] = part[b & 63] + z[j] * z[j];   // lane b % 64 visits its blocks in ascending order
                }
        }
        if (!normalise) continue;
        for (int off = 32; off >= 1; off >>= 1) {
            float nxt[64];
            for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
            for (int l = 0; l < 64; ++l) part[l] = nxt[l];
        }
        const float nrm = sqrtf(part[0]);
        for (uint32_t c = 0; c < dim; ++c) o[c] = o[c] / nrm;
    }
}

}  // extern "C"
