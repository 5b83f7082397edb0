/* cgvec_store.h — C ABI of the HOST-SIDE MIRROR of the reference's vector-store surface.
 *
 * The reference's toolchain (Rust) is absent from the build image, so the host code above
 * the kNN C ABI (cgvec.h) is written in C++ inside libcgvec_hip.so, mirroring the reference
 * interfaces for this path with the same names, argument meaning and error behaviour:
 *
 *   trait SurrealVectorBackend      crates/codegraph-vector/src/surreal_store.rs:11-22
 *   SurrealVectorStore: VectorStore crates/codegraph-vector/src/surreal_store.rs:25-86
 *   normalize_surreal_node_id       crates/codegraph-vector/src/surreal_store.rs:123-128
 *   surreal_embedding_column_for_dimension
 *                                   crates/codegraph-graph/src/surrealdb_storage.rs:1932-1952
 *   SemanticSearch::{search_by_embedding, semantic_search, hybrid_search,
 *                    multi_vector_search, find_related_code/combine_embeddings,
 *                    search_by_text/encode_query}
 *                                   crates/codegraph-vector/src/search.rs:78-267,271-418,519-592
 *   HasherEmbeddingService::embed   crates/codegraph-core/src/integration/graph_vector.rs:57-95
 *
 * NodeId is a 16-byte UUID (crates/codegraph-core/src/types.rs:8); on this ABI it travels as
 * 16 raw bytes (RFC 4122 byte order, what Uuid::as_bytes() returns). All calls return a
 * cgvec status (CGV_OK == 0); messages through cgv_last_error() as CodeGraphError::Vector.
 * The hot path underneath (vector_knn) is the HIP library — a store cannot be created
 * without a GPU, except the mock-backed one used to replay the reference's own seam test.
 */
#ifndef CGVEC_STORE_H
#define CGVEC_STORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgvs_store cgvs_store; /* SurrealVectorStore + SemanticSearch over one backend */

/* SearchFilters (search.rs:21-27) flattened for the ABI: NULL / 0 = "no constraint". */
typedef struct cgvs_filters {
    const char* const* languages;   /* Debug names of Language, e.g. "Rust" */
    uint32_t n_languages;
    const char* const* node_types;  /* Debug names of NodeType, e.g. "Function" */
    uint32_t n_node_types;
    const char* const* attr_keys;   /* attribute_equals */
    const char* const* attr_values;
    uint32_t n_attrs;
    const char* const* path_prefixes;
    uint32_t n_path_prefixes;
} cgvs_filters;

#define CGVS_COMBINE_OR_MAX 0      /* CombineMode::OrMax      (search.rs:31-34) */
#define CGVS_COMBINE_AND_AVERAGE 1 /* CombineMode::AndAverage */

/* SurrealVectorStore::new(Arc<HipKnnBackend>, ef_search) (surreal_store.rs:32-34): a store over
 * the HIP kNN backend; one device index per embedding column, created on first upsert. */
int cgvs_store_create(int dtype, int device_id, uint32_t ef_search, cgvs_store** out);
/* The same store over SEVERAL GPUs of the node: the seam holds ONE backend object (Arc<dyn SurrealVectorBackend>), so
 * that object owns all shards - every embedding column is a cgv_sharded handle (cgvec.h: rows dealt block-cyclically,
 * per-shard top-k exchanged once per search with RCCL / peer copies, merged on the first device). Everything above the
 * backend (UPSERT, id map, columns, SemanticSearch surface) is unchanged; a device may be listed more than once. */
int cgvs_store_create_sharded(int dtype, uint32_t n_devices, const int* device_ids, uint32_t ef_search, cgvs_store** out);

/* The reference's own seam test backend (surreal_store.rs:167-205 MockBackend): vector_knn
 * returns the canned (id, distance) list and records the column it was asked for. No GPU. */
int cgvs_store_create_mock(const char* const* ids, const float* distances, uint32_t n, uint32_t ef_search,
                           cgvs_store** out);
/* recorded columns of the mock backend, '\n' separated, into buf (NUL-terminated). */
int cgvs_mock_recorded_columns(cgvs_store* s, char* buf, size_t buf_len);

int cgvs_store_destroy(cgvs_store* s);

/* VectorStore::store_embeddings / SurrealVectorBackend::upsert_nodes (traits.rs:13,
 * surreal_store.rs:13): n nodes with ids (16 B each) and embeddings [n][dim] f32. A node id seen
 * before is updated in place (UPSERT). */
int cgvs_upsert_nodes(cgvs_store* s, uint32_t n, const uint8_t* ids16, const float* embeddings, uint32_t dim);

/* Metadata used by filters / hybrid ranking (SemanticSearch::upsert_node_metadata, search.rs:73-76). */
int cgvs_upsert_node_metadata(cgvs_store* s, const uint8_t* id16, const char* language, const char* node_type,
                              const char* file_path, const char* const* attr_keys,
                              const char* const* attr_values, uint32_t n_attrs);

#define CGVS_ID_TEXT 48 /* bytes per id slot of cgvs_vector_knn: "nodes:" + 36-char uuid + NUL, padded */

/* SurrealVectorBackend::vector_knn (surreal_store.rs:14-20): ids as "nodes:<uuid>" (each written
 * at out_ids + i*CGVS_ID_TEXT, NUL-terminated), distance = 1 - cosine ascending. `limit` is served up to the number of
 * rows in the column; more than CGV_MAX_K (2048) neighbours per query is CGV_ERR_INVALID_ARG, never a silent
 * truncation (the SemanticSearch entry points over-fetch prefetch_k(max(4*limit, limit+25)) = 12*limit
 * neighbours, search.rs:113,293: limit <= 170 fits). */
int cgvs_vector_knn(cgvs_store* s, const char* column, const float* query, uint32_t dim, uint32_t limit,
                    uint32_t ef_search, char* out_ids, float* out_dist, uint32_t* out_n);

/* VectorStore::search_similar (surreal_store.rs:61-85): empty query or limit 0 -> 0 results. */
int cgvs_search_similar(cgvs_store* s, const float* query, uint32_t dim, uint32_t limit, uint8_t* out_ids16,
                        uint32_t* out_n);

/* VectorStore::get_embedding (traits.rs:15): *out_dim = 0 when the node is unknown (None). */
int cgvs_get_embedding(cgvs_store* s, const uint8_t* id16, float* out, uint32_t cap, uint32_t* out_dim);

/* SemanticSearch::search_by_embedding (search.rs:91-144) without the colliding QueryHash cache
 * (SURVEY.md §8(a12)): prefetch max(3*limit, limit+10), exact re-score with search.rs:519-533,
 * stable sort desc, truncate, min-max normalise. */
int cgvs_search_by_embedding(cgvs_store* s, const float* query, uint32_t dim, uint32_t limit,
                             uint8_t* out_ids16, float* out_scores, uint32_t* out_n);

/* SemanticSearch::search_by_text (search.rs:78-81) with the reference's deterministic
 * 384-d hash embedder as encode_query (search.rs:178-205). */
int cgvs_search_by_text(cgvs_store* s, const char* text, uint32_t limit, uint8_t* out_ids16,
                        float* out_scores, uint32_t* out_n);

/* SemanticSearch::semantic_search (search.rs:271-313). filters may be NULL. */
int cgvs_semantic_search(cgvs_store* s, const float* query, uint32_t dim, const cgvs_filters* filters,
                         uint32_t limit, uint8_t* out_ids16, float* out_scores, uint32_t* out_n);

/* SemanticSearch::hybrid_search (search.rs:317-344). */
int cgvs_hybrid_search(cgvs_store* s, const float* query, uint32_t dim, const cgvs_filters* filters,
                       float vector_weight, uint32_t limit, uint8_t* out_ids16, float* out_scores,
                       uint32_t* out_n);

/* SemanticSearch::multi_vector_search (search.rs:347-418); the nq queries go to the GPU as ONE
 * batch instead of nq independent futures. Result order among equal scores follows the
 * reference's HashMap iteration order, which is unspecified; this mirror orders ties by node id. */
int cgvs_multi_vector_search(cgvs_store* s, const float* queries, uint32_t nq, uint32_t dim, int mode,
                             const cgvs_filters* filters, uint32_t limit, uint8_t* out_ids16,
                             float* out_scores, uint32_t* out_n);

/* SemanticSearch::combine_embeddings (search.rs:232-266): mean then L2-normalise. */
int cgvs_combine_embeddings(const float* embeddings, uint32_t n, uint32_t dim, float* out);

/* ---- index-time symbol resolution, embedding phase -----------------------------
 * Mirror of ai_semantic_match_sync PHASE 2 (crates/codegraph-mcp/src/indexer.rs:2790-2843) with
 * cosine_similarity_static (:2965-2979) and the char-trigram / Jaccard name filter (:2804-2821,
 * :2901-2932): for each unresolved symbol, the known symbol that passes the name filter with the
 * highest similarity, if it is > threshold (0.75 in the reference). The |unresolved| x |known|
 * similarities are ONE batched device search (CGV_METRIC_COSINE_SEQ); the name filter runs on the
 * host for the best-ranked candidates only. Ties (the reference iterates a HashMap) -> lowest index.
 * dtype F32 / F32_SHADOW reproduce the reference's similarities bit for bit (F32_SHADOW through the
 * batched MFMA path); bf16/fp16/fp8 evaluate them on the rounded embeddings. */
typedef struct cgvs_resolver cgvs_resolver;
int cgvs_resolver_create(uint32_t dim, int dtype, int device_id, cgvs_resolver** out);
int cgvs_resolver_destroy(cgvs_resolver* r);
/* symbol_embeddings (indexer.rs:1808-1867): names + flat f32 [n][dim]; index = insertion order */
int cgvs_resolver_add_symbols(cgvs_resolver* r, uint32_t n, const char* const* names, const float* embeddings);
uint64_t cgvs_resolver_count(const cgvs_resolver* r);
/* out_index[q] = index of the matched known symbol or -1; out_score[q] (may be NULL) = its similarity */
int cgvs_resolver_match(cgvs_resolver* r, uint32_t nq, const char* const* targets, const float* target_embeddings,
                        float threshold, int64_t* out_index, float* out_score);
float cgvs_trigram_jaccard(const char* a, const char* b);            /* indexer.rs:2901-2932 on lower-cased names */
int cgvs_symbol_name_eligible(const char* target, const char* name); /* indexer.rs:2804-2821 */

/* EmbeddingReRanker::rerank once the embeddings exist (crates/codegraph-vector/src/reranker.rs:113-157 with
 * cosine_similarity :94-109): out_order = candidate indices by similarity to the query, descending, ties in
 * candidate order (stable sort, :148); out_score = the similarities in that order. */
int cgvs_rerank_embeddings(int device_id, const float* query, const float* candidates, uint32_t n, uint32_t dim,
                           uint32_t* out_order, float* out_score);

/* Free functions of the mirrored surface. */
const char* cgvs_embedding_column_for_dimension(uint32_t dim);            /* surrealdb_storage.rs:1932-1952 */
int cgvs_normalize_node_id(const char* raw, char* out, size_t out_len);   /* surreal_store.rs:123-128 */
int cgvs_parse_node_id(const char* text, uint8_t* out_id16);              /* Uuid::parse_str */
int cgvs_format_node_id(const uint8_t* id16, char* out37);                /* hyphenated lower-case */
uint32_t cgvs_simple_hash(const char* text);                              /* search.rs:535-541 */
int cgvs_hash_embed(const char* text, uint32_t dim, float* out);          /* search.rs:178-205 */
uint64_t cgvs_prefetch_k(uint64_t limit);                                 /* search.rs:113 */
void cgvs_normalize_scores(float* scores, uint32_t n);                    /* search.rs:574-592 */
float cgvs_cosine_similarity(const float* a, const float* b, uint32_t n); /* search.rs:519-533 */

#ifdef __cplusplus
}
#endif
#endif /* CGVEC_STORE_H */
