/* callers.c - T native threads, each in a SERIAL loop of small cgv_search_f32 calls on ONE index: the reference's threading model
 * (VectorStore is Send + Sync and is called from a multi-thread tokio runtime through spawn_blocking, SURVEY.md section 8(b);
 * multi_vector_search issues B concurrent single-query searches, search.rs:358-361) with native threads, as a Rust host has them -
 * Python threads would serialise on the interpreter lock between calls. Plain C99 + pthreads against include/cgvec.h.
 *
 * Two forms:
 *   - a shared library (libcgv_callers.so; bench.py's `coalesced_callers` block and tests/test_gpu_coalesce.py load it with
 *     ctypes and pass an index they have filled): cgv_callers_run();
 *   - a program (built with -DCALLERS_MAIN by tests/test_c_client.py): abi_client's input file, one call per query, every
 *     caller's results written out for the comparison with the oracle.
 * Measurement / test infrastructure, not product code. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cgvec.h"

/* start gate: every caller arrives warmed up, the main thread opens it (or cancels the run) */
struct gate_s {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint32_t arrived;
    int open, cancel;
};

typedef struct {
    cgv_index* h;
    const float* queries;
    uint32_t n_queries, dim, k, threads, calls, warm, nq_per_call, tid;
    uint64_t* out_idx;
    float* out_score;
    double* lat_us;
    struct gate_s* gate;
    int rc;
    char err[256];
} caller_t;

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}

static void* caller_main(void* p) {
    caller_t* a = (caller_t*)p;
    const size_t per = (size_t)a->nq_per_call * a->k;
    uint64_t* wi = (uint64_t*)malloc(per * 8);   /* pageable scratch for the warm-up calls (a Rust Vec) */
    float* ws = (float*)malloc(per * 4);
    for (uint32_t i = 0; i < a->warm && a->rc == 0; ++i) {
        const size_t q0 = (((size_t)a->tid + (size_t)i * a->threads) * a->nq_per_call) % a->n_queries;
        a->rc = cgv_search_f32(a->h, a->queries + q0 * a->dim, a->nq_per_call, a->k, wi, ws);
    }
    if (a->rc) snprintf(a->err, sizeof(a->err), "%s", cgv_last_error());
    free(wi);
    free(ws);
    pthread_mutex_lock(&a->gate->mu);
    a->gate->arrived++;
    pthread_cond_broadcast(&a->gate->cv);
    while (!a->gate->open) pthread_cond_wait(&a->gate->cv, &a->gate->mu);
    const int cancelled = a->gate->cancel;
    pthread_mutex_unlock(&a->gate->mu);
    for (uint32_t i = 0; i < a->calls && a->rc == 0 && !cancelled; ++i) {
        const size_t call = (size_t)a->tid * a->calls + i;
        const size_t q0 = (call * a->nq_per_call) % a->n_queries;
        const double t0 = now_us();
        a->rc = cgv_search_f32(a->h, a->queries + q0 * a->dim, a->nq_per_call, a->k, a->out_idx + call * per, a->out_score + call * per);
        a->lat_us[call] = now_us() - t0;
        if (a->rc) snprintf(a->err, sizeof(a->err), "%s", cgv_last_error());
    }
    return NULL;
}

/* Call c (= tid * calls + i) searches queries [(c * nq_per_call) % n_queries, + nq_per_call) (n_queries must be a multiple of
 * nq_per_call) and writes out_idx / out_score [c][nq_per_call][k] and lat_us[c]. Returns the first failing status (message in
 * err256) or 0; *wall_s = from the release of the start barrier to the last thread's exit. */
int cgv_callers_run(void* index, const float* queries, uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t threads,
                    uint32_t calls, uint32_t warm, uint32_t nq_per_call, uint64_t* out_idx, float* out_score, double* lat_us,
                    double* wall_s, char* err256) {
    if (!index || !queries || !threads || !nq_per_call || n_queries % nq_per_call) return CGV_ERR_INVALID_ARG;
    caller_t* a = (caller_t*)calloc(threads, sizeof(caller_t));
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    struct gate_s gate;
    memset(&gate, 0, sizeof(gate));
    pthread_mutex_init(&gate.mu, NULL);
    pthread_cond_init(&gate.cv, NULL);
    uint32_t started = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        a[t].h = (cgv_index*)index;
        a[t].queries = queries;
        a[t].n_queries = n_queries;
        a[t].dim = dim;
        a[t].k = k;
        a[t].threads = threads;
        a[t].calls = calls;
        a[t].warm = warm;
        a[t].nq_per_call = nq_per_call;
        a[t].tid = t;
        a[t].out_idx = out_idx;
        a[t].out_score = out_score;
        a[t].lat_us = lat_us;
        a[t].gate = &gate;
        if (pthread_create(&th[t], NULL, caller_main, &a[t]) != 0) break;
        ++started;
    }
    int rc = 0;
    pthread_mutex_lock(&gate.mu);
    while (gate.arrived < started) pthread_cond_wait(&gate.cv, &gate.mu);   /* every caller has finished its warm-up calls */
    gate.cancel = started != threads;
    gate.open = 1;
    pthread_cond_broadcast(&gate.cv);
    pthread_mutex_unlock(&gate.mu);
    const double t0 = now_us();
    for (uint32_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (wall_s) *wall_s = 1e-6 * (now_us() - t0);
    if (started != threads) {
        rc = CGV_ERR_INTERNAL;
        if (err256) snprintf(err256, 256, "pthread_create failed after %u threads", started);
    }
    for (uint32_t t = 0; t < started && rc == 0; ++t)
        if (a[t].rc) {
            rc = a[t].rc;
            if (err256) snprintf(err256, 256, "caller %u: %.200s", t, a[t].err);
        }
    pthread_mutex_destroy(&gate.mu);
    pthread_cond_destroy(&gate.cv);
    free(a);
    free(th);
    return rc;
}

#ifdef CALLERS_MAIN
/* usage: callers <in.bin> <out.bin> [dtype] [threads]
 *   in.bin : u32 n, u32 dim, u32 nq, u32 k, then n*dim f32 corpus rows, then nq*dim f32 queries (abi_client's format)
 *   out.bin: nq*k u64 ids, then nq*k f32 scores - query j answered by ONE single-query call of thread j % threads
 * Exit code 3 = no GPU (the library has no CPU fallback). */
int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.bin out.bin [dtype] [threads]\n", argv[0]);
        return 1;
    }
    const int dtype = argc > 3 ? atoi(argv[3]) : CGV_DTYPE_BF16;
    const uint32_t threads = argc > 4 ? (uint32_t)atoi(argv[4]) : 16u;
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
    if (nq % threads) {
        fprintf(stderr, "nq must be a multiple of the thread count\n");
        return 1;
    }
    cgv_index* h = NULL;
    int rc = cgv_create(dim, CGV_METRIC_COSINE, dtype, 0, &h);
    if (rc) {
        fprintf(stderr, "cgv_create failed (%d): %s\n", rc, cgv_last_error());
        return rc == CGV_ERR_HIP ? 3 : 2;
    }
    if ((rc = cgv_add_f32(h, rows, n))) {
        fprintf(stderr, "cgv_add_f32 failed (%d): %s\n", rc, cgv_last_error());
        return 2;
    }
    uint64_t* oi = (uint64_t*)malloc((size_t)nq * k * 8);
    float* os = (float*)malloc((size_t)nq * k * 4);
    double* lat = (double*)malloc((size_t)nq * 8);
    double wall = 0.0;
    char err[256] = "";
    /* call c = tid * calls + i answers query c: calls = nq / threads */
    rc = cgv_callers_run(h, q, nq, dim, k, threads, nq / threads, 2, 1, oi, os, lat, &wall, err);
    if (rc) {
        fprintf(stderr, "callers failed (%d): %s\n", rc, err);
        return 2;
    }
    uint64_t st[8];
    if (cgv_get_coalesce_stats(h, st)) return 2;
    fprintf(stderr, "coalesce: batches %llu requests %llu lone %llu alone %llu max_batch %llu\n", (unsigned long long)st[0],
            (unsigned long long)st[1], (unsigned long long)st[3], (unsigned long long)st[4], (unsigned long long)st[5]);
    f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(oi, 8, (size_t)nq * k, f);
    fwrite(os, 4, (size_t)nq * k, f);
    fclose(f);
    cgv_destroy(h);
    return 0;
}
#endif
