/* Replays, from plain C, the exact call sequence of the Rust shim in INTEGRATION.md §3 (HipKnnBackend over
 * include/cgvec_store.h) - the shim has no logic, so this program IS its behaviour:
 *   new()                 cgvs_store_create(dtype, device 0, ef_search)
 *   upsert_nodes(batch A) cgvs_upsert_nodes            n rows, ids = UUIDs with the row number in the last 4 bytes
 *   vector_knn            cgvs_vector_knn("embedding_<dim>", query, limit)        -> "nodes:<uuid>", 1 - cosine asc
 *   upsert_nodes(known)   cgvs_upsert_nodes            row 7 again with a NEW embedding (= the second query)
 *   vector_knn            one hit for node 7, by its new embedding, no stale row
 *   get_node_embedding    cgvs_get_embedding           node 7 -> the new embedding; unknown id -> None (dim 0)
 *   search_similar        cgvs_search_similar          VectorStore path (ids only)
 * reference seam: crates/codegraph-vector/src/surreal_store.rs:11-22, 61-85 (UPSERT: schema `UPSERT nodes ...`).
 * Usage: shim_replay <in.bin> <out.txt> [dtype]   in.bin: u32 n, dim, nq(=2), k; n*dim f32 rows; 2*dim f32 queries.
 * out.txt: one line per hit "<phase> <rank> <row> <distance bits hex>"; phase 0 = first knn with query 0,
 * 1 = knn with query 1 after the re-upsert, 2 = search_similar with query 1. Exit 3 = no GPU. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cgvec.h"
#include "cgvec_store.h"

static int die(const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, cgv_last_error());
    return rc == CGV_ERR_HIP ? 3 : 2;
}
static void make_id(uint32_t row, uint8_t* id16) {
    static const uint8_t pre[12] = {0x6b, 0xa7, 0xb8, 0x10, 0x9d, 0xad, 0x41, 0xd1, 0x80, 0xb4, 0x00, 0xc0};
    memcpy(id16, pre, 12);
    id16[12] = (uint8_t)(row >> 24);
    id16[13] = (uint8_t)(row >> 16);
    id16[14] = (uint8_t)(row >> 8);
    id16[15] = (uint8_t)row;
}
/* "nodes:xxxxxxxx-xxxx-xxxx-xxxx-xxxxxxxxxxxx" -> row number (last 8 hex digits) */
static long row_of_text(const char* t) {
    if (strncmp(t, "nodes:", 6) != 0 || strlen(t) != 6 + 36) return -1;
    return strtol(t + 6 + 28, NULL, 16);
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    const int dtype = argc > 3 ? atoi(argv[3]) : CGV_DTYPE_F32_SHADOW;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    uint32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 1;
    const uint32_t n = hdr[0], dim = hdr[1], nq = hdr[2], k = hdr[3];
    if (nq != 2 || n < 16) return 1;
    float* rows = (float*)malloc((size_t)n * dim * 4);
    float* q = (float*)malloc((size_t)nq * dim * 4);
    if (fread(rows, 4, (size_t)n * dim, f) != (size_t)n * dim) return 1;
    if (fread(q, 4, (size_t)nq * dim, f) != (size_t)nq * dim) return 1;
    fclose(f);
    uint8_t* ids = (uint8_t*)malloc((size_t)n * 16);
    for (uint32_t i = 0; i < n; ++i) make_id(i, ids + (size_t)i * 16);

    cgvs_store* s = NULL;
    int rc = cgvs_store_create(dtype, 0, 64, &s);
    if (rc) return die("cgvs_store_create", rc);
    /* upsert_nodes, two batches like an indexer that streams nodes */
    if ((rc = cgvs_upsert_nodes(s, n / 2, ids, rows, dim))) return die("cgvs_upsert_nodes", rc);
    if ((rc = cgvs_upsert_nodes(s, n - n / 2, ids + (size_t)(n / 2) * 16, rows + (size_t)(n / 2) * dim, dim))) return die("cgvs_upsert_nodes", rc);

    FILE* o = fopen(argv[2], "w");
    if (!o) return 1;
    char col[32];
    snprintf(col, sizeof col, "%s", cgvs_embedding_column_for_dimension(dim));
    char* txt = (char*)malloc((size_t)k * CGVS_ID_TEXT);
    float* dist = (float*)malloc((size_t)k * 4);
    uint32_t got = 0;
    if ((rc = cgvs_vector_knn(s, col, q, dim, k, 64, txt, dist, &got))) return die("cgvs_vector_knn", rc);
    for (uint32_t i = 0; i < got; ++i) {
        uint32_t bits;
        memcpy(&bits, dist + i, 4);
        fprintf(o, "0 %u %ld %08x\n", i, row_of_text(txt + (size_t)i * CGVS_ID_TEXT), bits);
    }
    /* re-upsert of a KNOWN id with a new embedding (the second query itself): UPSERT, not append */
    if ((rc = cgvs_upsert_nodes(s, 1, ids + 7 * 16, q + dim, dim))) return die("cgvs_upsert_nodes (known id)", rc);
    if ((rc = cgvs_vector_knn(s, col, q + dim, dim, k, 64, txt, dist, &got))) return die("cgvs_vector_knn", rc);
    int seen7 = 0;
    for (uint32_t i = 0; i < got; ++i) {
        uint32_t bits;
        memcpy(&bits, dist + i, 4);
        const long r = row_of_text(txt + (size_t)i * CGVS_ID_TEXT);
        seen7 += (r == 7);
        fprintf(o, "1 %u %ld %08x\n", i, r, bits);
    }
    if (seen7 != 1 || row_of_text(txt) != 7) {
        fprintf(stderr, "re-upserted node: expected exactly one hit, first; got %d\n", seen7);
        return 2;
    }
    /* get_node_embedding: the NEW embedding for node 7, None for an unknown id */
    float* back = (float*)malloc(4096 * 4);
    uint32_t bdim = 0;
    if ((rc = cgvs_get_embedding(s, ids + 7 * 16, back, 4096, &bdim))) return die("cgvs_get_embedding", rc);
    if (bdim != dim) return 2;
    if (dtype == CGV_DTYPE_F32 || dtype == CGV_DTYPE_F32_SHADOW)
        if (memcmp(back, q + dim, (size_t)dim * 4) != 0) {
            fprintf(stderr, "get_embedding after re-upsert: stale embedding\n");
            return 2;
        }
    uint8_t unknown[16];
    make_id(n + 1000, unknown);
    if ((rc = cgvs_get_embedding(s, unknown, back, 4096, &bdim))) return die("cgvs_get_embedding (unknown)", rc);
    if (bdim != 0) return 2;
    /* the wrong column name is not a query-length guess: an unknown column has no rows */
    if ((rc = cgvs_vector_knn(s, "embedding_4096", q, dim, k, 64, txt, dist, &got)) == CGV_OK && got != 0) return 2;
    /* VectorStore::search_similar: ids only; empty query / limit 0 -> no results, no error (surreal_store.rs:62-64) */
    uint8_t* sid = (uint8_t*)malloc((size_t)k * 16);
    if ((rc = cgvs_search_similar(s, q + dim, dim, k, sid, &got))) return die("cgvs_search_similar", rc);
    for (uint32_t i = 0; i < got; ++i)
        fprintf(o, "2 %u %u 0\n", i, ((uint32_t)sid[i * 16 + 12] << 24) | ((uint32_t)sid[i * 16 + 13] << 16) | ((uint32_t)sid[i * 16 + 14] << 8) | sid[i * 16 + 15]);
    if ((rc = cgvs_search_similar(s, q, 0, k, sid, &got)) || got != 0) return 2;
    if ((rc = cgvs_search_similar(s, q, dim, 0, sid, &got)) || got != 0) return 2;
    fclose(o);
    cgvs_store_destroy(s);
    return 0;
}
