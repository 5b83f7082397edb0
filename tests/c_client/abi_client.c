/* A plain-C client of include/cgvec.h: what a compiled host (the Rust shim of INTEGRATION.md, a C++
 * service, ...) does with the library. No Python, no torch: dlopen-free static linking against
 * libcgvec_hip.so. Usage: abi_client <in.bin> <out.bin> [dtype] [n_shards]
 *   n_shards > 0: the same flow through ONE sharded handle (cgv_sharded_*) over n_shards shards placed
 *   on devices i % device_count (a 1-GPU box lists device 0 several times), which must give the same
 *   ids (global id = insertion index) and scores as the single index.
 *   in.bin : u32 n, u32 dim, u32 nq, u32 k, then n*dim f32 corpus rows, then nq*dim f32 queries
 *   out.bin: nq*k u64 ids, then nq*k f32 scores
 * Exit code 3 = no GPU (the library has no CPU fallback). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cgvec.h"

static int die(const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, cgv_last_error());
    return rc == CGV_ERR_HIP ? 3 : 2;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.bin out.bin [dtype]\n", argv[0]);
        return 1;
    }
    const int dtype = argc > 3 ? atoi(argv[3]) : CGV_DTYPE_BF16;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    uint32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 1;
    const uint32_t n = hdr[0], dim = hdr[1], nq = hdr[2], k = hdr[3];
    float* rows = (float*)malloc((size_t)n * dim * 4);
    float* q = (float*)malloc((size_t)nq * dim * 4);
    if (fread(rows, 4, (size_t)n * dim, f) != (size_t)n * dim) return 1;
    if (fread(q, 4, (size_t)nq * dim, f) != (size_t)nq * dim) return 1;
    fclose(f);

    const int n_shards = argc > 4 ? atoi(argv[4]) : 0;
    if (n_shards > 0) {
        int devs[64];
        const int ndev = cgv_device_count();
        if (ndev == 0) {
            fprintf(stderr, "no HIP device visible: libcgvec_hip has no CPU fallback\n");
            return 3;
        }
        for (int i = 0; i < n_shards && i < 64; ++i) devs[i] = i % ndev;
        cgv_sharded* sh = NULL;
        int src = cgv_sharded_create(dim, CGV_METRIC_COSINE, dtype, (uint32_t)n_shards, devs, &sh);
        if (src) return die("cgv_sharded_create", src);
        /* three appends that do not line up with the 4096-row chunks of the block-cyclic placement */
        const uint32_t a = n / 3, b = n / 2;
        if ((src = cgv_sharded_add_f32(sh, rows, a))) return die("cgv_sharded_add_f32", src);
        if ((src = cgv_sharded_add_f32(sh, rows + (size_t)a * dim, b - a))) return die("cgv_sharded_add_f32", src);
        if ((src = cgv_sharded_add_f32(sh, rows + (size_t)b * dim, n - b))) return die("cgv_sharded_add_f32", src);
        if (cgv_sharded_count(sh) != n || cgv_sharded_n_shards(sh) != (uint32_t)n_shards) return 2;
        uint64_t* sidx = (uint64_t*)malloc((size_t)nq * k * 8);
        float* ssc = (float*)malloc((size_t)nq * k * 4);
        if ((src = cgv_sharded_search_f32(sh, q, nq, k, sidx, ssc))) return die("cgv_sharded_search_f32", src);
        float* sback = (float*)malloc((size_t)dim * 4);
        if (cgv_sharded_get_row_f32(sh, (uint64_t)n + 5, sback) != CGV_ERR_OUT_OF_RANGE) return 2;
        if ((src = cgv_sharded_get_row_f32(sh, 1, sback))) return die("cgv_sharded_get_row_f32", src);
        cgv_sharded_stats sst;
        if ((src = cgv_sharded_get_stats(sh, &sst))) return die("cgv_sharded_get_stats", src);
        printf("rows=%llu shards=%u exchange=%u fallback=%llu search_ms=%.3f exchange_ms=%.3f\n",
               (unsigned long long)sst.n_rows, sst.n_shards, sst.exchange, (unsigned long long)sst.fallback_queries,
               sst.last_search_ms, sst.last_exchange_ms);
        f = fopen(argv[2], "wb");
        if (!f) return 1;
        fwrite(sidx, 8, (size_t)nq * k, f);
        fwrite(ssc, 4, (size_t)nq * k, f);
        fwrite(sback, 4, dim, f);
        fclose(f);
        cgv_sharded_destroy(sh);
        return 0;
    }

    cgv_index* h = NULL;
    int rc = cgv_create(dim, CGV_METRIC_COSINE, dtype, 0, &h);
    if (rc) return die("cgv_create", rc);
    /* two appends, like a store that grows */
    if ((rc = cgv_add_f32(h, rows, n / 2))) return die("cgv_add_f32", rc);
    if ((rc = cgv_add_f32(h, rows + (size_t)(n / 2) * dim, n - n / 2))) return die("cgv_add_f32", rc);
    if (cgv_count(h) != n || cgv_dim(h) != dim) return 2;

    uint64_t* idx = (uint64_t*)malloc((size_t)nq * k * 8);
    float* sc = (float*)malloc((size_t)nq * k * 4);
    if ((rc = cgv_search_f32(h, q, nq, k, idx, sc))) return die("cgv_search_f32", rc);

    /* the degenerate inputs of surreal_store.rs:62-64 succeed without touching the outputs */
    if ((rc = cgv_search_f32(h, q, 0, k, idx, sc)) || (rc = cgv_search_f32(h, q, nq, 0, idx, sc))) return die("empty", rc);
    /* a dimension error is an error, with a message (simd_ops.rs:16-18) */
    float* back = (float*)malloc((size_t)dim * 4);
    if (cgv_get_row_f32(h, (uint64_t)n + 5, back) != CGV_ERR_OUT_OF_RANGE || !strlen(cgv_last_error())) return 2;
    if ((rc = cgv_get_row_f32(h, 1, back))) return die("cgv_get_row_f32", rc);

    /* the trait-level call is ONE query (traits.rs:14): every query again, alone, through pageable buffers - the same rows */
    {
        uint64_t* i1 = (uint64_t*)malloc((size_t)k * 8);
        float* s1 = (float*)malloc((size_t)k * 4);
        for (uint32_t i = 0; i < nq && i < 12; ++i) {
            if ((rc = cgv_search_f32(h, q + (size_t)i * dim, 1, k, i1, s1))) return die("cgv_search_f32 (one query)", rc);
            if (memcmp(i1, idx + (size_t)i * k, (size_t)k * 8) || memcmp(s1, sc + (size_t)i * k, (size_t)k * 4)) {
                fprintf(stderr, "single-query call %u differs from the batch's answer\n", i);
                return 2;
            }
        }
        free(i1);
        free(s1);
    }
    /* host in / host out with batches in flight, from plain C: pinned staging buffers from the library, their device aliases
     * into cgv_search_begin_f32_dev, cgv_search_end in order - the results land in the pinned arrays in place */
    {
        const uint32_t depth = cgv_max_batches_in_flight(h) < 2 ? 1 : 2;
        const size_t qb = (size_t)nq * dim * 4, ib = (size_t)nq * k * 8, sb = (size_t)nq * k * 4;
        float* pq[2] = {NULL, NULL};
        uint64_t* pi[2] = {NULL, NULL};
        float* ps[2] = {NULL, NULL};
        uint64_t t[2] = {0, 0};
        for (uint32_t j = 0; j < depth; ++j) {
            pq[j] = (float*)cgv_alloc_pinned(qb);
            pi[j] = (uint64_t*)cgv_alloc_pinned(ib);
            ps[j] = (float*)cgv_alloc_pinned(sb);
            if (!pq[j] || !pi[j] || !ps[j]) return die("cgv_alloc_pinned", CGV_ERR_HIP);
            memcpy(pq[j], q, qb);
            memset(pi[j], 0, ib);
        }
        for (uint32_t j = 0; j < depth; ++j) {
            const float* dq = (const float*)cgv_host_device_alias(0, pq[j], qb);
            uint64_t* di = (uint64_t*)cgv_host_device_alias(0, pi[j], ib);
            float* ds = (float*)cgv_host_device_alias(0, ps[j], sb);
            if (!dq || !di || !ds) return die("cgv_host_device_alias", CGV_ERR_HIP);
            if ((rc = cgv_search_begin_f32_dev(h, dq, nq, k, di, ds, &t[j]))) return die("cgv_search_begin_f32_dev", rc);
        }
        for (uint32_t j = 0; j < depth; ++j)
            if ((rc = cgv_search_end(h, t[j]))) return die("cgv_search_end", rc);
        for (uint32_t j = 0; j < depth; ++j) {
            if (memcmp(pi[j], idx, ib) || memcmp(ps[j], sc, sb)) {
                fprintf(stderr, "pipelined batch %u differs from the blocking call's answer\n", j);
                return 2;
            }
            /* and the blocking call uses pinned buffers in place too */
            memset(pi[j], 0, ib);
            if ((rc = cgv_search_f32(h, pq[j], nq, k, pi[j], ps[j]))) return die("cgv_search_f32 (pinned)", rc);
            if (memcmp(pi[j], idx, ib) || memcmp(ps[j], sc, sb)) return 2;
            if ((rc = cgv_free_pinned(pq[j])) || (rc = cgv_free_pinned(pi[j])) || (rc = cgv_free_pinned(ps[j]))) return die("cgv_free_pinned", rc);
        }
        if ((rc = cgv_set_spin_us(h, 0))) return die("cgv_set_spin_us", rc);   /* block at once: same answer */
        if ((rc = cgv_search_f32(h, q, nq, k, idx, sc))) return die("cgv_search_f32 (no spin)", rc);
        if ((rc = cgv_set_spin_us(h, 3000))) return die("cgv_set_spin_us", rc);
    }

    cgv_stats st;
    if ((rc = cgv_get_stats(h, &st))) return die("cgv_get_stats", rc);
    printf("rows=%llu device_bytes=%llu path=%u fallback=%llu\n", (unsigned long long)st.n_rows,
           (unsigned long long)st.device_bytes, st.last_path, (unsigned long long)st.fallback_queries);

    f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(idx, 8, (size_t)nq * k, f);
    fwrite(sc, 4, (size_t)nq * k, f);
    fwrite(back, 4, dim, f);
    fclose(f);
    cgv_destroy(h);
    return 0;
}
