/* cgvec.h — C ABI of libcgvec_hip.so: MI355X-native brute-force kNN for CodeGraph.
 *
 * This is the drop-in boundary for the CPU kNN of the reference's `codegraph-vector`
 * crate. The reference is pure Rust with no FFI; these entry points
 * are what a Rust `HipKnnBackend: SurrealVectorBackend + VectorStore` shim binds (the
 * shim source is in INTEGRATION.md). Each entry cites the reference interface it
 * replaces (paths relative to the reference repo).
 *
 * Conventions (mirroring codegraph_core::Result / CodeGraphError::Vector(String),
 * crates/codegraph-core/src/error.rs:17-18):
 *   - every call returns an int status (CGV_OK == 0); cgv_last_error() returns a
 *     thread-local, NUL-terminated message for the last non-zero status;
 *   - degenerate inputs succeed with empty results (nq == 0, k == 0, empty index),
 *     as surreal_store.rs:62-64 does; dimension mismatch is an error
 *     (simd_ops.rs:16-18);
 *   - the caller owns every host/device buffer passed in or out; the handle owns all
 *     device memory it allocates; nothing panics or aborts across the boundary;
 *   - row id = insertion index (0-based, + index_base); the NodeId<->row-id map stays
 *     on the caller's side (cf. persistent.rs:492 vector_id_mapping);
 *   - results are ordered (score desc, row id asc) — a valid outcome of the
 *     reference's unstable parallel sort (simd_ops.rs:379) and of its stable sorts
 *     (optimization.rs:395, search.rs:132);
 *   - scores are the reference's own f32 arithmetic (simd_ops.rs:15-78 cosine /
 *     :149-183 dot) evaluated on the stored (rounded-to-dtype, then upcast) values,
 *     reproduced bit-for-bit by the exact re-score kernel;
 *   - short results are padded with (UINT64_MAX, -inf).
 *
 * Threading: cgv_search_* may be called concurrently on one handle — the handle owns a small pool
 * of search contexts (HIP stream + scratch each), so concurrent callers, or one caller using
 * cgv_search_begin/_end, overlap on the device; cgv_add_* / cgv_update_row / cgv_reserve wait for the
 * searches in flight and then run alone; cgv_destroy needs exclusive access — the contract of the
 * reference's `tokio::sync::Mutex<SurrealDbStorage>` (surreal_store.rs:45-47) plus `&mut self` on
 * store_embeddings (traits.rs:13).
 */
#ifndef CGVEC_H
#define CGVEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgv_index cgv_index; /* opaque handle */

/* similarity metric */
#define CGV_METRIC_COSINE 0 /* SIMDVectorOps::adaptive_cosine_similarity, simd_ops.rs:281-295 */
#define CGV_METRIC_DOT 1    /* SIMDVectorOps::dot_product_avx2, simd_ops.rs:149-183 */
#define CGV_METRIC_COSINE_SEQ 2 /* the sequential dot/(sqrt(na)*sqrt(nb)) cosine of SemanticSearch (search.rs:519-533)
                                  and of the symbol resolver (crates/codegraph-mcp/src/indexer.rs:2965-2979) */
#define CGV_METRIC_COSINE_SCALAR 3 /* SIMDVectorOps::cosine_similarity_scalar (simd_ops.rs:257-278) for EVERY length: what
                                  adaptive_cosine_similarity (:281-295) computes on a host WITHOUT AVX2 + FMA and on every
                                  non-x86_64 host (aarch64: `#[cfg(not(target_arch = "x86_64"))]`, :291-294). Pick it when the
                                  results must be bit-equal to a reference that runs on such a host; CGV_METRIC_COSINE is
                                  bit-equal to the reference on an AVX2 host (scalar formula below 32 elements only). */

/* storage dtype of the corpus in HBM (queries are rounded to the same dtype). Results never depend on magnitudes; speed
 * can: rows / queries whose largest magnitude is outside [2^-40, 2^40] (embeddings never are) are answered by the exact scan
 * - such a query alone, such a stored row for the whole index - because the MFMA coarse pass' error bound assumes no
 * under- / overflow in squared norms and products. */
#define CGV_DTYPE_F32 0     /* the reference's own Vec<f32> (node.rs:14) */
#define CGV_DTYPE_BF16 1
#define CGV_DTYPE_FP16 2
#define CGV_DTYPE_FP8E4M3 3 /* OCP e4m3fn, per-row power-of-two scale; cosine metric only; rows and queries whose
                               largest magnitude is outside [2^-48, 2^48] (zero rows excepted) are rejected with
                               CGV_ERR_INVALID_ARG */
#define CGV_DTYPE_F32_SHADOW 4 /* f32 rows (results = the reference's f32 arithmetic on the UNROUNDED inputs, like
                                  CGV_DTYPE_F32) + a bf16 copy that only feeds the MFMA coarse pass of batched
                                  searches; the exactness check accounts for the copy's rounding residual. 1.5x
                                  the memory of CGV_DTYPE_F32, which always takes the exact scan. */

/* status codes */
#define CGV_OK 0
#define CGV_ERR_INVALID_ARG 1
#define CGV_ERR_DIM_MISMATCH 2 /* VectorError::DimensionMismatch, error.rs:5-6 */
#define CGV_ERR_HIP 3          /* HIP runtime failure / no MI355X visible */
#define CGV_ERR_OOM 4
#define CGV_ERR_NONFINITE 5    /* NaN/Inf in corpus or query: the reference panics (simd_ops.rs:379) */
#define CGV_ERR_OUT_OF_RANGE 6 /* VectorError::IndexOutOfBounds, error.rs:14-15 */
#define CGV_ERR_INTERNAL 7
#define CGV_ERR_IO 8           /* corpus file errors; messages follow memory.rs:242-374 */
#define CGV_ERR_BUSY 9         /* the calling thread already holds every search context (begin without end); a cgv_sharded handle
                                  is written / read while a batch begun on it is in flight */

/* Largest k of a search. k <= CGV_FAST_MAX_K runs the MFMA coarse pass + exact re-score; larger k (the
 * over-fetch of SemanticSearch: prefetch_k(max(4*limit, limit+25)), search.rs:113,293) takes the exact
 * full scan on the device - same results, HBM-bound instead of MFMA-bound. */
#define CGV_MAX_K 2048u
#define CGV_FAST_MAX_K 228u

/* Library/ABI version (major<<16 | minor). Minor 9 (round 6): cgv_get_sample_repair_stats. Minor 8 (round 6): cgv_normalize_rows_scalar_f32 (the scalar arm of
 * parallel_normalize_vectors), cgv_get_small_batch_stats, cgv_get_phase_times (profiling level 3), cgv_synth_rows_f32_dev (the
 * bench's counter-based inputs). Minor 7 (round 6): cgv_set_coalesce / cgv_get_coalesce_stats (concurrent small
 * cgv_search_f32 calls share one device batch). Minor 6 (round 5): cgv_set_spin_us, cgv_sharded_force_exchange - the library reads no
 * environment variable; cgv_alloc_pinned / cgv_free_pinned. Minor 4 (round 3): cgv_sharded_search_begin_f32 / _end / _max_batches_in_flight,
 * cgv_set_profiling levels, pinned host buffers used in place by cgv_search_f32. Minor 5 (round 4): cgv_search_packed_begin_f32_dev
 * / cgv_search_packed_end / cgv_merge_packed_flag_dev (the join-free exchange), cgv_host_device_alias, CGV_METRIC_COSINE_SCALAR /
 * CGV_OP_COSINE_SCALAR, CGV_ERR_BUSY from cgv_sharded_* writers / readers while a batch is in flight. */
uint32_t cgv_version(void);

/* Thread-local message for the last failing call on this thread ("" if none). */
const char* cgv_last_error(void);

/* Number of HIP devices visible (0 when none; never fails). */
int cgv_device_count(void);

/* Create an empty index on one device.
 * Replaces: GpuAcceleration::new + set_device (gpu.rs:109-131,…; a mock in the
 * reference) and the construction of a SurrealVectorBackend (surreal_store.rs:32-34).
 * dim in 1..=8192 (config.rs:221-224). */
int cgv_create(uint32_t dim, int metric, int dtype, int device_id, cgv_index** out);

/* Free all device memory. Replaces GpuAcceleration::deallocate (gpu.rs). */
int cgv_destroy(cgv_index* h);

/* Pre-size the corpus allocation for n_rows rows (optional; add grows geometrically). */
int cgv_reserve(cgv_index* h, uint64_t n_rows);

/* Append n rows (flat row-major f32 [n][dim], HOST memory). Rows are rounded
 * (RNE) to the storage dtype on device; per-row inverse norms are computed there.
 * Replaces: VectorStore::store_embeddings (traits.rs:13) /
 * SurrealVectorBackend::upsert_nodes (surreal_store.rs:13) /
 * GpuAcceleration::upload_vectors (gpu.rs:221-246: flat &[f32] + dimension). */
int cgv_add_f32(cgv_index* h, const float* rows_host, uint64_t n);

/* Same, rows already in DEVICE memory of this index's device. */
int cgv_add_f32_dev(cgv_index* h, const float* rows_dev, uint64_t n);

/* Overwrite stored row `id` (local id) with a new f32 row (HOST pointer): the in-place half of
 * UPSERT (SurrealVectorBackend::upsert_nodes on a known node id, surreal_store.rs:13). */
int cgv_update_row_f32(cgv_index* h, uint64_t id, const float* row_host);

/* Number of rows stored / dimension / dtype / metric. */
uint64_t cgv_count(const cgv_index* h);
uint32_t cgv_dim(const cgv_index* h);

/* Added to every row id this index reports (global id = index_base + local row);
 * used when the corpus is row-sharded across devices. */
int cgv_set_index_base(cgv_index* h, uint64_t base);

/* Block-cyclic id map (used by cgv_sharded_*): this index is shard `shard` of `n_shards`, global rows are
 * dealt to the shards in chunks of `chunk_rows`; a local row r is reported as
 * index_base + ((r / chunk_rows) * n_shards + shard) * chunk_rows + r % chunk_rows. n_shards == 1 = identity. */
int cgv_set_id_map(cgv_index* h, uint32_t chunk_rows, uint32_t n_shards, uint32_t shard);

/* Drop the rows >= n_rows (roll back the tail of the index; the capacity is kept). Used by callers that
 * apply one logical insert to several indices (cgv_sharded_add_f32) and must undo it everywhere when one
 * of them rejects its part (NaN/Inf rows, out of memory). */
int cgv_truncate(cgv_index* h, uint64_t n_rows);

/* Batched kNN: nq queries (flat f32 [nq][dim], HOST), top-k each.
 * out_idx / out_score: HOST arrays of nq*k entries, row-major per query.
 * Replaces: ParallelVectorOps::parallel_top_k_search (simd_ops.rs:361-383) per query;
 * SurrealVectorBackend::vector_knn (surreal_store.rs:14-20) — the shim reports
 * distance = 1 - score; and the N-independent-searches batch of
 * SemanticSearch::multi_vector_search (search.rs:358-361).
 * Buffers in pinned memory (hipHostMalloc / hipHostRegister) are used IN PLACE: the conversion kernel reads the
 * queries over PCIe and the last kernel writes the results (no staging copies); pageable buffers are staged by the
 * library. On a non-zero status the out arrays are unspecified (they may already have been written). */
int cgv_search_f32(cgv_index* h, const float* queries_host, uint32_t nq, uint32_t k,
                   uint64_t* out_idx_host, float* out_score_host);

/* Group commit of concurrent small calls. The trait-level call of the reference is ONE query
 * (VectorStore::search_similar(&self, &[f32], limit), traits.rs:14; surreal_store.rs:61-85; caller search.rs:114-117) and its
 * multi-query caller issues B concurrent single-query searches (search.rs:358-361, try_join_all): concurrent cgv_search_f32
 * calls of <= 8 queries each on one handle are therefore merged into ONE device batch (<= max_batch_queries queries, default 64 =
 * one launch that streams the corpus once for all of them) - every caller gets exactly the ids, scores, status and message
 * of a lone call (a NaN / Inf query fails its own caller only). At most max_batches_in_flight (default 1, <=
 * cgv_max_batches_in_flight()) such batches are on the device at once; calls that arrive meanwhile collect in the next batch -
 * a lone caller never waits. window_us (default 250): the thread that starts a batch right behind a MULTI-caller batch lingers
 * at most that long for the callers that batch has just released - until as many have joined as it had, the batch is full, or
 * arrivals stop (for a third of the window, less when few callers are expected back); 0 = never. max_batch_queries == 0 or max_batches_in_flight == 0 switches the merging off (every call is its own batch, as
 * before ABI minor 7).
 * cgv_get_coalesce_stats: out8 = {batches that carried > 1 caller, callers served by them, their queries, eligible calls that ran
 * alone, requests handed back to run alone, largest batch (queries), batches whose leader lingered, 0}. */
#define CGV_COALESCE_MAX_BATCH_QUERIES 64u
#define CGV_COALESCE_BATCHES_IN_FLIGHT 1u
#define CGV_COALESCE_WINDOW_US 250u
int cgv_set_coalesce(cgv_index* h, uint32_t max_batch_queries, uint32_t max_batches_in_flight, uint32_t window_us);
/* Small batches (<= 64 queries: one launch that keeps two rows per 1/1024 of the corpus and checks the result against what it
 * left out): out4 = {searches that took that form, their queries whose check failed (three of the best rows in one cell: adjacent
 * near-duplicates do it), of those answered by re-scanning only the offending cells, of those sent to the exact scan of the whole
 * corpus}. Results are exact either way. */
int cgv_get_small_batch_stats(cgv_index* h, uint64_t* out4);
/* Large batches whose first threshold comes from an EMITTING sample launch (the sampled tiles are not scored again): out1[0] = queries
 * whose check failed only because one 64-row cell of the sample left out a row that may belong to the top-k, and that the final
 * kernel put right by itself (the offending cells re-scored with the reference's arithmetic) - no exact scan, no host round trip,
 * no second exchange on the row-sharded path. Not counted in cgv_stats.fallback_queries. */
int cgv_get_sample_repair_stats(cgv_index* h, uint64_t* out1);
int cgv_get_coalesce_stats(cgv_index* h, uint64_t* out8);

/* Same with DEVICE pointers for queries and outputs (results stay in HBM; the call
 * returns after the work is enqueued and the exactness check has been read back). */
int cgv_search_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k,
                       uint64_t* out_idx_dev, float* out_score_dev);

/* Append n rows given as f64 (flat [n][dim], HOST): the SurrealDB `embedding_<dim>` columns hold
 * Vec<f64> (crates/codegraph-graph/src/surrealdb_storage.rs:1955-1977); narrowed with `as f32`
 * (round to nearest even) on the device, then ingested like cgv_add_f32. */
int cgv_add_f64(cgv_index* h, const double* rows_host, uint64_t n);

/* Corpus files in the reference's mmap format, MemoryOptimizer::save_to_mmap / load_from_mmap
 * (crates/codegraph-vector/src/memory.rs:242-374): 16-byte header {u64 vector_count, u64 dimension}
 * (native endian) followed by row-major f32.
 *   cgv_load_mmap: map the file, validate it like load_from_mmap (too small / dimension mismatch
 *     against the handle's dim / size mismatch -> CGV_ERR_IO or CGV_ERR_DIM_MISMATCH with the
 *     reference's messages), and APPEND its rows: page cache -> pinned staging -> device,
 *     double-buffered so the host copy of one 64-MiB chunk overlaps the H2D + conversion of the
 *     previous one. *out_rows = rows appended.
 *   cgv_write_mmap_f32: save_to_mmap for host rows (n == 0 writes no file, memory.rs:243-245).
 *   cgv_save_mmap: the handle's stored rows (upcast to f32) in the same format. */
int cgv_load_mmap(cgv_index* h, const char* path, uint64_t* out_rows);
int cgv_write_mmap_f32(const char* path, const float* rows_host, uint64_t n, uint32_t dim);
int cgv_save_mmap(cgv_index* h, const char* path);

/* Split form of cgv_search_f32_dev for callers that keep several batches in flight (the handle
 * owns a small pool of search contexts, each with its own HIP stream and scratch — SURVEY.md §8(b)
 * "internal stream pool"): begin enqueues the whole batch behind whatever the caller has queued on
 * the handle's stream (cgv_set_stream) and returns a ticket without waiting for the device; end
 * waits for that batch, runs the exact path for any query whose guarantee check failed, and
 * releases the context. queries / outputs must stay valid and untouched until end returns.
 * begin blocks while all contexts are in flight. ticket 0 (nq == 0 or k == 0) needs no end.
 * cgv_search_f32_dev == begin + end; concurrent callers on one handle overlap the same way. */
int cgv_search_begin_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k,
                             uint64_t* out_idx_dev, float* out_score_dev, uint64_t* ticket);
int cgv_search_end(cgv_index* h, uint64_t ticket);
/* Size of the context pool = how many begin calls one thread may have outstanding before it must call end
 * (a further begin would wait for a context that only this thread can release). */
uint32_t cgv_max_batches_in_flight(const cgv_index* h);

/* Copy stored row `id` (local id, without index_base) back as f32 (upcast of the
 * stored value). Replaces VectorStore::get_embedding (traits.rs:15) /
 * SurrealVectorBackend::get_node_embedding (surreal_store.rs:21). */
int cgv_get_row_f32(cgv_index* h, uint64_t id, float* out_host);

/* ---- the reference's building blocks, with their exact f32 operation order, on device ----
 * op for cgv_batch_similarity_f32: */
#define CGV_OP_COSINE 0              /* SIMDVectorOps::adaptive_cosine_similarity, simd_ops.rs:281-295 */
#define CGV_OP_DOT 1                 /* SIMDVectorOps::dot_product_avx2, simd_ops.rs:149-183 */
#define CGV_OP_L2 2                  /* SIMDVectorOps::l2_distance_avx2, simd_ops.rs:105-143 */
#define CGV_OP_COSINE_SEQ 3          /* cosine_similarity, search.rs:519-533 (sequential sums, no FMA) */
#define CGV_OP_COSINE_DISTANCE_SEQ 4 /* cosine_distance, optimization.rs:404-418 / gpu.rs:324-338 */
#define CGV_OP_COSINE_SCALAR 6       /* cosine_similarity_scalar, simd_ops.rs:257-278, for every length (see
                                        CGV_METRIC_COSINE_SCALAR; 5 is reserved) */

/* One query (HOST, f32[dim]) against the first limit_rows stored rows (0 = all), out_host[i] =
 * op(query, row i). Replaces ParallelVectorOps::parallel_batch_similarity (simd_ops.rs:347-358),
 * SIMDVectorOps::batch_cosine_similarity_avx2 (:85-99) and, with CGV_OP_COSINE_DISTANCE_SEQ and
 * limit_rows = limit, GpuAcceleration::compute_distances / compute_distances_cpu
 * (gpu.rs:248-322: distances of the FIRST `limit` rows, not a top-k). */
int cgv_batch_similarity_f32(cgv_index* h, const float* query_host, int op, uint64_t limit_rows, float* out_host);

/* Scores of given (query, stored row) pairs in ONE device launch: out[q][j] = op(query q, row ids[q][j]) for
 * nq queries (HOST f32 [nq][dim], used UNROUNDED as the reference does) and m LOCAL row ids per query
 * (HOST u64 [nq][m]; UINT64_MAX or an id beyond the index scores 0.0, like a missing embedding in
 * calculate_similarity_score, search.rs:207-217). Replaces the per-hit `get_embedding` + `cosine_similarity`
 * loop of SemanticSearch::search_by_embedding (search.rs:119-129, op = CGV_OP_COSINE_SEQ) - prefetch_k host
 * round trips per query become one call per batch. */
int cgv_score_ids_f32(cgv_index* h, const float* queries_host, uint32_t nq, int op, const uint64_t* ids_host,
                      uint32_t m, float* out_host);

/* ModelOptimizer::search_baseline (optimization.rs:376-402): ascending cosine_distance, stable
 * (ties keep row order), take(limit). out_dist_host may be NULL. Row ids are local (no index_base). */
int cgv_search_baseline_f32(cgv_index* h, const float* query_host, uint32_t limit, uint64_t* out_idx_host,
                            float* out_dist_host, uint32_t* out_n);

/* ParallelVectorOps::parallel_normalize_vectors / SIMDVectorOps::normalize_avx2
 * (simd_ops.rs:386-419, 189-222), in place on a flat HOST matrix [n][dim], computed on device. */
int cgv_normalize_rows_f32(int device_id, float* rows_host, uint64_t n, uint32_t dim);
/* The same entry on a host that never takes the AVX2 kernel - the scalar arm of parallel_normalize_vectors
 * (simd_ops.rs:394-403 without AVX2 + FMA, :406-415 on every non-x86_64 host; the counterpart of CGV_METRIC_COSINE_SCALAR):
 * the squares summed in order by one accumulator, rows whose sum is not > 0 left alone, every element DIVIDED by the norm. */
int cgv_normalize_rows_scalar_f32(int device_id, float* rows_host, uint64_t n, uint32_t dim);

/* SURVEY.md section 8(d) synthetic inputs (measurement plumbing, no reference counterpart): rows [row0, row0 + nrows) of the
 * counter-based stream `seed` - Philox4x32-10 keyed (seed, row, col / 4) -> Box-Muller N(0, 1) f32 -> (normalise != 0)
 * L2-normalised in f32 - written row-major [nrows][dim] to DEVICE memory on `stream`. Any chunking and any rank produce the
 * same values, and the CPU checker's cgo_synth_rows restates the contract bit for bit (csrc/synth.hip states it). */
int cgv_synth_rows_f32_dev(int device_id, uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, int normalise,
                           float* out_dev, void* stream);

/* Merge G partial top-k lists per query on device `device_id` (the step after the
 * all-gather of per-shard partial results): inputs are
 * [g][nq][k] (idx u64, score f32) DEVICE arrays, outputs [nq][k], ordered
 * (score desc, idx asc); padding entries (UINT64_MAX) sort last. `stream` is a
 * hipStream_t (NULL = default stream). */
int cgv_merge_topk_dev(int device_id, const uint64_t* idx_dev, const float* score_dev, uint32_t g,
                       uint32_t nq, uint32_t k, uint64_t* out_idx_dev, float* out_score_dev,
                       void* stream);

/* The same exchange with ONE packed buffer per rank (what sharded.py all-gathers): cgv_pack_topk_dev
 * writes, per query, cgv_packed_width(k) = 3k (+1 if k is odd) int32 words: k u64 ids | k f32 scores;
 * cgv_merge_packed_dev merges g such buffers laid out [g][nq][width] (the all-gather output) with
 * (score desc, id asc). Saves the concatenate / slice copies around the collective. The out arrays of
 * cgv_merge_packed_dev may also be pinned HOST memory (hipHostMalloc / hipHostRegister): the kernel then writes the
 * merged results there directly. */
uint32_t cgv_packed_width(uint32_t k);
int cgv_pack_topk_dev(int device_id, const uint64_t* idx_dev, const float* score_dev, uint32_t nq, uint32_t k,
                      uint32_t* out_rec_dev, void* stream);
int cgv_merge_packed_dev(int device_id, const uint32_t* rec_dev, uint32_t g, uint32_t nq, uint32_t k,
                         uint64_t* out_idx_dev, float* out_score_dev, void* stream);

/* Device-visible address of a pinned / registered HOST range on `device_id` (equal to the host address for hipHostMalloc
 * memory, possibly different for hipHostRegister-ed memory), or NULL when [host_ptr, host_ptr + bytes) is not WHOLLY such
 * memory mapping to one contiguous device range. What cgv_search_f32 uses to decide "in place or staged" per buffer. */
void* cgv_host_device_alias(int device_id, const void* host_ptr, size_t bytes);

/* Pinned (page-locked, device-mapped) host memory for a caller without the HIP runtime in its own link line (a Rust shim, a C
 * program): query batches and result arrays in such memory are used IN PLACE by cgv_search_f32 and - through their device
 * alias (cgv_host_device_alias) - by cgv_search_begin_f32_dev / cgv_search_packed_begin_f32_dev: a worker that keeps one pinned
 * staging buffer saves two copies per batch (hipHostMalloc(hipHostMallocPortable | hipHostMallocMapped) / hipHostFree).
 * cgv_alloc_pinned returns NULL on failure (message in cgv_last_error()). */
void* cgv_alloc_pinned(size_t bytes);
int cgv_free_pinned(void* p);

/* ---- one rank's share of a batch WITHOUT a host join between the shard search and the exchange -------------------------
 * (row-sharded deployment, one process per GPU; the reference has no counterpart - its "batch" is B
 * independent futures, search.rs:358-361.)
 * cgv_search_packed_begin_f32_dev enqueues ON `consumer_stream` (a hipStream_t; NULL = the legacy default stream) the shard
 * search of the batch (queries_dev: device memory or the device alias of pinned host memory, cgv_host_device_alias) and, right
 * behind its last kernel, the packing of the top-k into rec_out_dev (cgv_packed_width(k) int32 words per query, device memory
 * owned by the caller). The caller then enqueues on that stream the all-gather of every rank's records and
 * cgv_merge_packed_flag_dev, and synchronises ONCE - the host never sits between the search and the collective, and the batch
 * never leaves the stream (no cross-stream event waits). Work queued on the handle's stream (cgv_set_stream) before the call is
 * ordered in front of it. A query whose top-k the device could not prove (ties / near-duplicates at the k' boundary, a zero query, an
 * exact-scan-only index ...) is packed PROVISIONAL: id slot 0 = CGV_PROVISIONAL_ID. The merge raises *redo_flag_dev (a
 * device word, or a pinned host word written in place; the caller zeroes it) when any rank's list of any query is
 * provisional. Every rank merges the same gathered records, so ALL ranks read the same flag - no second collective to agree:
 *   flag == 0: the merged results are final; cgv_search_packed_end(ticket) releases the context (and returns the search's
 *              status: a NaN / Inf query fails HERE with CGV_ERR_NONFINITE, its records were provisional);
 *   flag != 0: every rank calls cgv_search_packed_end (the rank that owns the provisional queries runs the exact scan and
 *              re-packs rec_out_dev, *repacked = 1), then all-gathers and merges again.
 * Tickets share the handle's contexts with cgv_search_begin_f32_dev (cgv_max_batches_in_flight). */
#define CGV_PROVISIONAL_ID 0xFFFFFFFFFFFFFFFEull
int cgv_search_packed_begin_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k, uint32_t* rec_out_dev,
                                    void* consumer_stream, uint64_t* ticket);
int cgv_search_packed_end(cgv_index* h, uint64_t ticket, int* repacked);
int cgv_merge_packed_flag_dev(int device_id, const uint32_t* rec_dev, uint32_t g, uint32_t nq, uint32_t k,
                              uint64_t* out_idx_dev, float* out_score_dev, uint32_t* redo_flag_dev, void* stream);

/* Run this handle's work on an external hipStream_t (e.g. PyTorch's current stream).
 * The value is used as-is: NULL is HIP's legacy default ("null") stream — which is what
 * PyTorch uses unless told otherwise — NOT "no stream". */
int cgv_set_stream(cgv_index* h, void* stream);

/* Go back to the handle's own (non-blocking) stream, the default after cgv_create. */
int cgv_use_own_stream(cgv_index* h);

/* Block until all work enqueued by this handle has finished.
 * Replaces GpuAcceleration::synchronize (gpu.rs). */
int cgv_synchronize(cgv_index* h);

/* Observability (cf. GpuAcceleration::get_memory_stats, gpu.rs; PerformanceMonitor). */
typedef struct cgv_stats {
    uint64_t n_rows;
    uint64_t device_bytes;        /* HBM held by the handle */
    uint64_t searches;            /* cgv_search_* calls */
    uint64_t queries;             /* total queries */
    uint64_t fallback_queries;    /* queries re-run through the exact full scan */
    uint64_t overflow_queries;    /* subset of the above caused by candidate-list overflow */
    float last_eps;               /* error bound used by the exactness check (score units) */
    float max_observed_err;       /* max |coarse - exact| seen on re-scored candidates */
    float last_coarse_ms;         /* HIP-event time of the dominant (MFMA coarse) kernel, last search;
                                     valid when profiling is on */
    float last_total_ms;          /* HIP-event time of the whole device pipeline, last search */
    uint64_t coarse_rows;         /* corpus rows covered by the timed coarse launch (last search) */
    uint32_t last_kprime;         /* candidates re-scored per query */
    uint32_t last_path;           /* 0 = exact scan, 1 = MFMA coarse + exact re-score */
} cgv_stats;
int cgv_get_stats(cgv_index* h, cgv_stats* out);

/* HIP-event timing of each search, on the stream its kernels are launched on: 0 = off; 1 = the dominant coarse launch
 * carries a start / end event pair of its own (the dispatch's timestamps, hipExtLaunchKernelGGL: cgv_stats.last_coarse_ms;
 * ~4-8 us per batch); 2 = also two event records around the whole pipeline (last_total_ms). Each record is a packet on
 * the stream: level 2 measured ~10 us per batch on short searches. */
int cgv_set_profiling(cgv_index* h, int enabled);
/* Level 3 adds three more records per search, at the phase boundaries of the MFMA pipeline; out_us4 = device microseconds of
 * the last finished search's {query conversion, first threshold (sample launch + tau, or boot + select), emitting coarse
 * launches + selections, final kernel + publish} - zeros for an exact-scan search, a search that fell back, or level < 3.
 * What the multi-GPU bench line prints per rank, so that a first run on N GPUs shows where a rank's batch goes. */
int cgv_get_phase_times(cgv_index* h, float* out_us4);

/* How a search's end waits for the device: it polls the batch's stream / event for up to spin_us microseconds (default 3000,
 * capped at 1 s) and then blocks in the driver. A batch takes 0.3-1.5 ms and the wake-up of a blocked wait costs tens of
 * microseconds of it, so the default burns one host core for the length of the batch; a caller that runs many handles on few
 * cores (a tokio server: call the search inside spawn_blocking either way) sets 0 = block at once. */
int cgv_set_spin_us(cgv_index* h, uint32_t spin_us);

/* Tuning knob for tests: force the exact full-scan path (1) or auto (0). */
int cgv_set_force_exact(cgv_index* h, int enabled);

/* Debug/validation: dense coarse scores of the MFMA path for small inputs,
 * out_dev[nq][n] (f32, DEVICE). Used by tests to check the GEMM tile mapping. */
int cgv_debug_coarse_scores_dev(cgv_index* h, const float* queries_dev, uint32_t nq,
                                float* out_dev);

/* ---- one index over several devices (cgv_create(.., n_devices, device_ids*); §8(e)) ----
 * The backend object the reference's seam injects (`Arc<dyn SurrealVectorBackend>`, surreal_store.rs:11-22,
 * 32-34; `Box<dyn VectorStore + Send>`, graph_vector.rs:241-253) is ONE object, so it must own all shards:
 * a cgv_sharded is one handle over n_devices device indices in one process, one worker thread per device.
 *   rows    : global row id = insertion index, dealt to the shards block-cyclically in chunks of
 *             CGV_SHARD_CHUNK_ROWS (incremental upserts keep the shards balanced; cgv_set_id_map);
 *   search  : the query batch goes to every device, each shard runs the single-device pipeline, the
 *             per-shard top-k travel as packed 12-byte records in ONE exchange - an RCCL all-gather over
 *             xGMI when the devices are distinct (librccl is loaded on first use), peer / same-device
 *             copies into the root's buffer otherwise (CGV_EXCHANGE_*; the same device may be listed more
 *             than once, which is how a 1-GPU box exercises the whole path) - and the root merges
 *             n_devices * k records per query with (score desc, id asc). Exact: the global top-k is a
 *             subset of the union of the per-shard top-k.
 * Thread-safety: calls on one cgv_sharded are serialised by the handle (a call holds it while it runs; searches split
 * into begin / end overlap on the devices between the two calls). */
typedef struct cgv_sharded cgv_sharded;
#define CGV_SHARD_CHUNK_ROWS 4096u
#define CGV_EXCHANGE_NONE 0 /* one shard */
#define CGV_EXCHANGE_RCCL 1 /* ncclAllGather of the packed records (RCCL over xGMI) */
#define CGV_EXCHANGE_COPY 2 /* hipMemcpyPeerAsync / device-to-device copies into the root's gather buffer */

int cgv_sharded_create(uint32_t dim, int metric, int dtype, uint32_t n_devices, const int* device_ids,
                       cgv_sharded** out);
int cgv_sharded_destroy(cgv_sharded* s);
/* Pre-size every shard for total_rows rows in all (optional). */
int cgv_sharded_reserve(cgv_sharded* s, uint64_t total_rows);
/* Append n rows (HOST f32 [n][dim]); applied to all shards or to none (NaN/Inf -> CGV_ERR_NONFINITE). */
int cgv_sharded_add_f32(cgv_sharded* s, const float* rows_host, uint64_t n);
int cgv_sharded_update_row_f32(cgv_sharded* s, uint64_t id, const float* row_host);
int cgv_sharded_get_row_f32(cgv_sharded* s, uint64_t id, float* out_host);
/* cgv_score_ids_f32 with GLOBAL row ids: each pair is scored on the shard that owns the row. */
int cgv_sharded_score_ids_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, int op, const uint64_t* ids_host,
                              uint32_t m, float* out_host);
uint64_t cgv_sharded_count(const cgv_sharded* s);
uint32_t cgv_sharded_n_shards(const cgv_sharded* s);
/* Borrowed handle of shard i (statistics, profiling switches); do not add to / destroy it. */
cgv_index* cgv_sharded_shard(cgv_sharded* s, uint32_t i);
/* cgv_search_f32 over all shards; out arrays HOST [nq][k], ids global. Any k <= CGV_MAX_K on any number of shards
 * (n_shards * k <= 4096: LDS merge; beyond: a slower G-way wave merge). = begin + end below. */
int cgv_sharded_search_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, uint32_t k,
                           uint64_t* out_idx_host, float* out_score_host);
/* The same in two halves, so that batches overlap on the devices (the counterpart of cgv_search_begin_f32_dev /
 * cgv_search_end for ONE backend object that owns all shards, surreal_store.rs:11-22): begin copies the queries
 * (the caller's buffer is free again on return), enqueues every shard's search and returns a ticket; end waits for the
 * shards, runs pack + exchange + merge and fills the HOST out arrays given to begin (one host join per batch). Up to
 * cgv_sharded_max_batches_in_flight() tickets may be open (CGV_ERR_BUSY beyond); with two in flight the exchange,
 * merge and result copy of batch i run beside the search of batch i + 1. nq == 0 or k == 0: ticket 0, nothing to end.
 * On error the out arrays are unspecified. The calling thread's current HIP device is restored before every
 * cgv_sharded_* call returns. */
int cgv_sharded_search_begin_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, uint32_t k,
                                 uint64_t* out_idx_host, float* out_score_host, uint64_t* ticket);
int cgv_sharded_search_end(cgv_sharded* s, uint64_t ticket);
uint32_t cgv_sharded_max_batches_in_flight(const cgv_sharded* s);
/* Which exchange the handle uses (CGV_EXCHANGE_*); cgv_sharded_set_exchange forces RCCL or COPY
 * (RCCL needs distinct devices). A handle over ONE shard has nothing to exchange (CGV_EXCHANGE_NONE) unless
 * cgv_sharded_force_exchange(s, 1) was called: then its batches go through pack -> one-rank ncclAllGather (or the copy when
 * RCCL cannot be loaded) -> merge as well - how the RCCL branch is executed on a single-GPU box (tests). */
int cgv_sharded_exchange(const cgv_sharded* s);
int cgv_sharded_set_exchange(cgv_sharded* s, int kind);
int cgv_sharded_force_exchange(cgv_sharded* s, int enabled);
typedef struct cgv_sharded_stats {
    uint64_t n_rows, device_bytes, searches, queries, fallback_queries;
    uint32_t n_shards, exchange;
    float last_search_ms;   /* host wall time of the last cgv_sharded_search_f32 */
    float last_exchange_ms; /* of which: pack + exchange + merge + result copy (after the slowest shard finished) */
} cgv_sharded_stats;
int cgv_sharded_get_stats(cgv_sharded* s, cgv_sharded_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* CGVEC_H */
