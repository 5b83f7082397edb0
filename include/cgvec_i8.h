/* cgvec_i8.h — C ABI of the int8 "optimized" scan.
 *
 * Reference: crates/codegraph-vector/src/optimization.rs
 *   ModelOptimizer::quantize_unit_range_symmetric  :212-224   round(clamp(x,-1,1)*127) as i8
 *   ModelOptimizer::quantize_batch (8-bit arm)     :226-283   row-major u8 = q + 128
 *   OptimizationResult::search_optimized           :63-150    one f32 query against the u8 rows:
 *       query quantised the same way; per row integer dot and integer |v|^2 (i32), rows with
 *       |v|^2 == 0 skipped; score = dot as f32 / (|q| * sqrt(|v|^2 as f32)); a running buffer of
 *       `limit` (idx, score) kept ascending by stable sorts, its minimum replaced by any strictly
 *       larger score; final stable sort descending -> row indices.
 *
 * Here the O(N*D) part (integer dot + norm + the f32 score of every row) runs on the GPU with
 * v_dot4_i32_i8 over the same u8 bytes; the O(N) buffer policy — whose tie behaviour depends on
 * the visiting order (among equal minima the most recently inserted entry is replaced first) — is
 * replayed on the host over the candidates that can enter the buffer at all, so the returned
 * indices equal the reference's, ties included. Same status / error convention as cgvec.h.
 * No CPU fallback: creating a handle without a GPU fails.
 */
#ifndef CGVEC_I8_H
#define CGVEC_I8_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgv_i8_index cgv_i8_index; /* OptimizationResult{optimized_data, metadata} resident in HBM */

/* metadata "dimension" = dim, "quantization_bits" = 8 (optimization.rs:66-82). */
int cgv_i8_create(uint32_t dim, int device_id, cgv_i8_index** out);
int cgv_i8_destroy(cgv_i8_index* h);

/* Append n rows already quantised by the reference (QuantizedBatch.data, optimization.rs:268-272):
 * row-major u8 [n][dim], HOST memory. */
int cgv_i8_add_u8(cgv_i8_index* h, const uint8_t* data_host, uint64_t n);

/* Append n f32 rows (HOST, flat [n][dim]), quantised on the device exactly like
 * quantize_batch's 8-bit arm (round half away from zero; NaN -> 0 like Rust's `as i32`). */
int cgv_i8_add_f32(cgv_i8_index* h, const float* rows_host, uint64_t n);

uint64_t cgv_i8_count(const cgv_i8_index* h); /* metadata "vector_count" */

/* Copy stored row `id` (u8 codes) back to the host. */
int cgv_i8_get_row_u8(cgv_i8_index* h, uint64_t id, uint8_t* out_host);

/* OptimizationResult::search_optimized (optimization.rs:63-150). limit 0 is treated as 1 (:64);
 * an all-zero quantised query or an empty index gives *out_n = 0 (:71-73, :113-115).
 * out_idx_host must hold max(limit,1) entries. */
int cgv_i8_search_optimized(cgv_i8_index* h, const float* query_host, uint32_t query_len, uint64_t limit,
                            uint64_t* out_idx_host, uint64_t* out_n);

/* Diagnostic: the f32 score of every row for one query (NaN where the reference skips the row). */
int cgv_i8_scores_f32(cgv_i8_index* h, const float* query_host, uint32_t query_len, float* out_host);

/* ModelOptimizer::quantize_batch, 8-bit arm, as a free function on the device (no handle kept). */
int cgv_quantize_u8_f32(int device_id, const float* rows_host, uint64_t n, uint32_t dim, uint8_t* out_host);

/* ModelOptimizer::quantize_batch, 4-bit arm (optimization.rs:248-262, quantize_unit_range_u4 :338-343):
 * out_host is [n][ceil(dim/2)] bytes, two codes per byte, low nibble first. (search_optimized itself
 * only scans 8-bit data: for other widths it returns 0..limit, optimization.rs:76-82.) */
int cgv_quantize_u4_f32(int device_id, const float* rows_host, uint64_t n, uint32_t dim, uint8_t* out_host);

#ifdef __cplusplus
}
#endif
#endif /* CGVEC_I8_H */
