/* cgvec_quant.h — C ABI of the reference's vector quantisers, trained and applied
 * on the device with the reference's exact f32 operation order.
 *
 * Reference: crates/codegraph-vector/src/persistent.rs
 *   ScalarQuantizer   :331-477   train: global (uniform) or per-dimension min / max, scale = 2^nbits / range,
 *                                bias = min; encode: ((v - bias) * scale).max(0).min(2^nbits - 1) as u32 packed as
 *                                u8 / u16-LE / u32-LE; decode: q / scale + bias
 *   ProductQuantizer  :116-329   m sub-quantisers of dsub = dim/m dims, ksub = 2^nbits centroids each; k-means:
 *                                centroids start as the first ksub training sub-vectors (index i % n), at most 50
 *                                iterations of {assign to the nearest centroid by sequential Euclidean distance,
 *                                first minimum wins; centroid = sequential f32 sum of its vectors / count}; the
 *                                loop stops early only when every vector is assigned to centroid 0 (the reference
 *                                resets `assignments` each iteration, :264); encode = nearest centroid per sub-vector
 * Same status / error convention as cgvec.h; messages follow the reference's CodeGraphError::Vector texts.
 * No CPU fallback.
 */
#ifndef CGVEC_QUANT_H
#define CGVEC_QUANT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgv_sq cgv_sq; /* ScalarQuantizer */
int cgv_sq_create(uint32_t dim, uint32_t nbits, int uniform, int device_id, cgv_sq** out); /* :345-353 */
int cgv_sq_destroy(cgv_sq* q);
int cgv_sq_train_f32(cgv_sq* q, const float* rows_host, uint64_t n);                        /* :355-408 */
int cgv_sq_params(cgv_sq* q, float* scales_out, float* biases_out);                         /* [dim] each */
uint32_t cgv_sq_bytes_per_value(const cgv_sq* q);                                           /* 1, 2 or 4 (:424-433) */
int cgv_sq_encode_f32(cgv_sq* q, const float* rows_host, uint64_t n, uint8_t* out_host);    /* :410-436 */
int cgv_sq_decode(cgv_sq* q, const uint8_t* codes_host, uint64_t n, float* out_host);       /* :438-476 */

typedef struct cgv_pq cgv_pq; /* ProductQuantizer */
int cgv_pq_create(uint32_t dim, uint32_t m, uint32_t nbits, int device_id, cgv_pq** out);   /* :134-152; nbits <= 8 */
int cgv_pq_destroy(cgv_pq* q);
int cgv_pq_train_f32(cgv_pq* q, const float* rows_host, uint64_t n);                        /* :155-187, :245-318 */
int cgv_pq_centroids(cgv_pq* q, float* out_host);                                           /* [m][ksub][dsub] */
int cgv_pq_encode_f32(cgv_pq* q, const float* rows_host, uint64_t n, uint8_t* codes_out);   /* [n][m], :190-218 */
int cgv_pq_decode(cgv_pq* q, const uint8_t* codes_host, uint64_t n, float* out_host);       /* :221-242 */

#ifdef __cplusplus
}
#endif
#endif /* CGVEC_QUANT_H */
