// search.hip - the kNN pipeline behind the C ABI: storage and ingest of an index, the launch of every kernel of a batched search
// (query conversion, sample, emitting coarse launches, selections, final re-score, the exact scan and the small-batch repair),
// the search-context pool and the host side of one batch. Entry points: abi.hip; planner: plan.cpp; shared types: cgvec_internal.h.
//
// Pipeline of one batched search (bf16/fp16 corpus):
//   prep_rows(queries)                         round queries to the storage dtype, norms
//   [boot]    dense scores of 16 tiles spread over the corpus -> select   k' candidates + first tau
//   [stage 1] coarse(strided sample, tau)    -> select      tighter tau from a few % of the rows
//   [stage 2] coarse(all other tiles, tau)   -> select      DOMINANT KERNEL (MFMA GEMM)
//   rescore                                    exact reference arithmetic on k' candidates,
//                                              (score desc, row asc), guarantee check
//   [fallback] exact full scan for queries whose candidate set could not be proven.
// f32 corpora (the reference's native layout) take the exact full-scan path directly.
//
// There is NO CPU fallback anywhere in this file: without a HIP device every entry
// point that needs one fails with CGV_ERR_HIP.
#include "cgvec_internal.h"

namespace cgvi {

// bytes of storage holding rows [0, nrows) (nrows rounded up to whole 256-row tiles for the
// blocked layout, whose tiles are contiguous)
size_t storage_bytes(const cgv_index* h, uint64_t nrows) {
    const uint64_t r = (nrows + 255) / 256 * 256;
    return (size_t)r * h->ld * h->esize;
}

size_t shadow_bytes(const cgv_index* h, uint64_t nrows) {
    const uint64_t r = (nrows + 255) / 256 * 256;
    return (size_t)r * h->lds * 2;
}

size_t device_bytes(const cgv_index* h) {
    size_t b = 0;
    if (h->srows) b += shadow_bytes(h, h->cap);
    if (h->rows) b += storage_bytes(h, h->cap) + (size_t)h->cap * 8 + ((size_t)h->cap / 32 + 1) * 8;
    b += h->addstage.bytes;
    for (const SearchCtx& c : h->ctx) b += c.bytes();
    return b;
}

template <int DT>
void launch_prep(const float* in, uint64_t n, uint32_t D, uint32_t ld, uint64_t row0, char* out, float* norm,
                 float* invn, int8_t* rexp, uint32_t* nonfinite, hipStream_t s, uint32_t* z0, uint32_t* z1) {
    if (n == 0) return;
    uint64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(prep_rows_kernel<DT>, dim3((unsigned)blocks), dim3(256), 0, s, in, n, D, ld, row0, out,
                       norm, invn, rexp, nonfinite, z0, z1);
}

// Convert n f32 rows into the index' storage at absolute rows [row0, row0+n).
int prep_dispatch(int dtype, const float* in, uint64_t n, uint32_t D, uint32_t ld, uint64_t row0, char* out,
                  float* norm, float* invn, int8_t* rexp, uint32_t* nonfinite, hipStream_t s,
                  uint32_t* z0, uint32_t* z1) {
    switch (dtype) {
        case CGV_DTYPE_F32: launch_prep<DT_F32>(in, n, D, ld, row0, out, norm, invn, rexp, nonfinite, s, z0, z1); break;
        case CGV_DTYPE_BF16: launch_prep<DT_BF16>(in, n, D, ld, row0, out, norm, invn, rexp, nonfinite, s, z0, z1); break;
        case CGV_DTYPE_FP16: launch_prep<DT_FP16>(in, n, D, ld, row0, out, norm, invn, rexp, nonfinite, s, z0, z1); break;
        case CGV_DTYPE_FP8E4M3: launch_prep<DT_FP8>(in, n, D, ld, row0, out, norm, invn, rexp, nonfinite, s, z0, z1); break;
        default: return fail(CGV_ERR_INVALID_ARG, "dtype not supported by this build");
    }
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

int grow(cgv_index* h, uint64_t need) {
    if (need <= h->cap) return CGV_OK;
    if (need > 0xFFFFFF00ull) return fail(CGV_ERR_INVALID_ARG, "more than 2^32-256 rows per device index");
    uint64_t ncap = std::max<uint64_t>(need, h->cap + h->cap / 2);
    ncap = std::max<uint64_t>(ncap, 1024);
    ncap = (ncap + 255) / 256 * 256;
    char* rows = nullptr;
    float *norm = nullptr, *invn = nullptr, *bmin = nullptr, *bmax = nullptr;
    int8_t* rexp = nullptr;
    size_t rb = storage_bytes(h, ncap);
    size_t nblk = (size_t)ncap / 32 + 8;
    HIPCHK(hipMalloc((void**)&rows, rb));
    HIPCHK(hipMemsetAsync(rows, 0, rb, h->stream));  // padding rows / columns must read as zero
    HIPCHK(hipMalloc((void**)&norm, ncap * 4));
    HIPCHK(hipMalloc((void**)&invn, ncap * 4));
    HIPCHK(hipMalloc((void**)&bmin, nblk * 4));
    HIPCHK(hipMalloc((void**)&bmax, nblk * 4));
    HIPCHK(hipMalloc((void**)&rexp, ncap));
    HIPCHK(hipMemsetAsync(rexp, 0, ncap, h->stream));  // padding rows: scale 2^0 (an E8M0 byte of 255 would be NaN)
    char* srows = nullptr;
    if (h->shadow) {
        HIPCHK(hipMalloc((void**)&srows, shadow_bytes(h, ncap)));
        HIPCHK(hipMemsetAsync(srows, 0, shadow_bytes(h, ncap), h->stream));
        if (h->n) HIPCHK(hipMemcpyAsync(srows, h->srows, shadow_bytes(h, h->n), hipMemcpyDeviceToDevice, h->stream));
    }
    if (h->n) {
        HIPCHK(hipMemcpyAsync(rows, h->rows, storage_bytes(h, h->n), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(norm, h->norm, h->n * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(invn, h->invn, h->n * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(rexp, h->rexp, h->n, hipMemcpyDeviceToDevice, h->stream));
        size_t ob = (size_t)(h->n + 31) / 32;
        HIPCHK(hipMemcpyAsync(bmin, h->blk_min, ob * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(bmax, h->blk_max, ob * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->rows) {
        (void)hipFree(h->rows);
        (void)hipFree(h->norm);
        (void)hipFree(h->invn);
        (void)hipFree(h->blk_min);
        (void)hipFree(h->blk_max);
        (void)hipFree(h->rexp);
        if (h->srows) (void)hipFree(h->srows);
    }
    h->srows = srows;
    h->rows = rows;
    h->rexp = rexp;
    h->norm = norm;
    h->invn = invn;
    h->blk_min = bmin;
    h->blk_max = bmax;
    h->cap = ncap;
    return CGV_OK;
}

// Enqueue the ingest of cnt f32 rows (device memory) at absolute rows [row0, row0+cnt): storage
// conversion, norms, per-32-row-block norm bounds, running max norm. No synchronisation; capacity
// must already be there. ingest_finish() reads the flags back and publishes the new row count.
int ingest_enqueue(cgv_index* h, const float* rows_dev, uint64_t cnt, uint64_t row0, uint64_t n_valid) {
    hipStream_t s = h->stream;
    int rc = prep_dispatch(h->dtype, rows_dev, cnt, h->D, h->ld, row0, h->rows, h->norm, h->invn, h->rexp,
                           h->flags + F_NONFINITE_C, s);
    if (rc) return rc;
    if (h->shadow) {  // second pass: bf16 copy, ITS norms (the coarse pass works on it), rounding residuals
        hipLaunchKernelGGL(shadow_rows_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, s, rows_dev, cnt, h->D, h->lds,
                           row0, h->srows, h->norm, h->invn, (float*)nullptr, h->resmax_dev);
        HIPCHK(hipGetLastError());
    }
    const uint64_t n_new = row0 + cnt;
    if (n_valid < n_new) n_valid = n_new;  // rows that exist once this ingest is done (update_row: unchanged count)
    const uint64_t b0 = row0 / 32, b1 = (n_new + 31) / 32;
    hipLaunchKernelGGL(block_norm_stats_kernel, dim3((unsigned)((b1 - b0 + 255) / 256)), dim3(256), 0, s,
                       h->norm, n_valid, b0, b1, h->blk_min, h->blk_max);
    hipLaunchKernelGGL(max_norm_kernel, dim3(1), dim3(1024), 0, s, h->norm, row0, n_new, h->max_norm_dev);
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

// State an add must be able to return to when a later chunk of it fails (NaN/Inf rows, OOM):
// the caller sees a failed add, so no row of it may stay in the index.

// Drop every row >= snap.n and restore the corpus-wide statistics the dropped rows were folded into.
// The storage beyond snap.n is dead capacity (overwritten by the next add).
int ingest_rollback(cgv_index* h, const IngestSnapshot& snap) {
    hipStream_t s = h->stream;
    (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    h->n = snap.n;
    h->max_norm_c = snap.max_norm;
    h->res_rel_c = snap.res_rel;
    h->res_abs_c = snap.res_abs;
    HIPCHK(hipMemsetAsync(h->flags, 0, F_COUNT * 4, s));
    HIPCHK(hipMemcpyAsync(h->max_norm_dev, &h->max_norm_c, 4, hipMemcpyHostToDevice, s));
    if (h->shadow) {
        const float r2[2] = {snap.res_rel, snap.res_abs};
        HIPCHK(hipMemcpyAsync(h->resmax_dev, r2, 8, hipMemcpyHostToDevice, s));
    }
    if (h->rows && (snap.n & 31u)) {  // the partially filled 32-row block at the boundary: bounds over the surviving rows
        const uint64_t b0 = snap.n / 32;
        hipLaunchKernelGGL(block_norm_stats_kernel, dim3(1), dim3(256), 0, s, h->norm, snap.n, b0, b0 + 1, h->blk_min,
                           h->blk_max);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
    return CGV_OK;
}

int ingest_finish(cgv_index* h, uint64_t n_new) {
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->h_flags, h->flags, F_COUNT * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h->h_flags + F_COUNT, h->max_norm_dev, 4, hipMemcpyDeviceToHost, s));
    if (h->shadow) HIPCHK(hipMemcpyAsync(h->h_flags + F_COUNT + 1, h->resmax_dev, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h->h_flags[F_NONFINITE_C] & 1u)  // nothing is published; the caller rolls back to its snapshot
        return fail(CGV_ERR_NONFINITE,
                    "corpus rows contain NaN/Inf (the reference panics on NaN at simd_ops.rs:379); the add was not applied");
    if (h->h_flags[F_NONFINITE_C] & 2u)
        return fail(CGV_ERR_INVALID_ARG,
                    "fp8 storage: a row's largest magnitude is outside [2^-48, 2^48]; the add was not applied");
    if (h->h_flags[F_NONFINITE_C] & 4u) {  // sticky: the coarse pass' error bound does not cover such rows
        h->wide_range = true;
        HIPCHK(hipMemsetAsync(h->flags + F_NONFINITE_C, 0, 4, s));
    }
    h->n = n_new;
    memcpy(&h->max_norm_c, h->h_flags + F_COUNT, 4);
    if (h->shadow) {
        memcpy(&h->res_rel_c, h->h_flags + F_COUNT + 1, 4);
        memcpy(&h->res_abs_c, h->h_flags + F_COUNT + 2, 4);
    }
    return CGV_OK;
}

int add_dev_locked(cgv_index* h, const float* rows_dev, uint64_t cnt) {
    if (cnt == 0) return CGV_OK;
    int rc = grow(h, h->n + cnt);
    if (rc) return rc;
    if ((rc = ingest_enqueue(h, rows_dev, cnt, h->n))) return rc;
    return ingest_finish(h, h->n + cnt);
}

// Run `body` (one or more add_dev_locked / ingest calls) atomically: on failure the index is exactly
// what it was before (row count, statistics, sticky flags).


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it for every
// kernel that needs more than 64 KiB of dynamic LDS once per device (cgv_create calls this with the
// device current), so that several devices in one process (cgv_sharded_*) and concurrent first
// searches all find it in place.
std::mutex g_attr_mu;
std::vector<char> g_attr_done;

int ensure_kernel_attrs(int device) {
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if ((size_t)device < g_attr_done.size() && g_attr_done[device]) return CGV_OK;
#define CGV_ATTR(K, BYTES) HIPCHK(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
    if (int rc = coarse_attrs_bf16()) return rc;
    if (int rc = coarse_attrs_fp16()) return rc;
    if (int rc = coarse_attrs_fp8()) return rc;
    CGV_ATTR(select_kernel, SELECT_LDS_KEYS * 8 + 65536);
    const int cap = 96 * 1024;
    CGV_ATTR(rescore_kernel<DT_BF16>, cap);
    CGV_ATTR(rescore_kernel<DT_FP16>, cap);
    CGV_ATTR(rescore_kernel<DT_FP8>, cap);
    CGV_ATTR(rescore_kernel<DT_F32>, cap);
    CGV_ATTR(final_kernel<DT_BF16>, cap);
    CGV_ATTR(final_kernel<DT_FP16>, cap);
    CGV_ATTR(final_kernel<DT_FP8>, cap);
    CGV_ATTR(final_kernel<DT_F32>, cap);
#undef CGV_ATTR
    if (g_attr_done.size() <= (size_t)device) g_attr_done.resize((size_t)device + 1, 0);
    g_attr_done[device] = 1;
    return CGV_OK;
}

int launch_coarse(int dtype, int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (dtype == CGV_DTYPE_BF16) return launch_coarse_bf16(mode, a, W, s, ev0, ev1);
    if (dtype == CGV_DTYPE_FP16) return launch_coarse_fp16(mode, a, W, s, ev0, ev1);
    if (dtype == CGV_DTYPE_FP8E4M3) return launch_coarse_fp8(mode, a, W, s, ev0, ev1);
    return fail(CGV_ERR_INTERNAL, "coarse path: unsupported dtype");
}

SelectArgs make_select_args(SearchCtx* c, uint32_t nq, uint32_t nqt, uint32_t nsplit, uint32_t kprime,
                            const float* dense, uint32_t n_dense, uint64_t expected, size_t* lds_out, uint32_t extra_keys) {
    SelectArgs sa;
    sa.cand = c->cand.as<uint2>();
    sa.cand_cnt = c->candcnt.as<uint32_t>();
    sa.dense = dense;
    sa.best = c->best.as<uint64_t>();
    sa.nbest = c->nbest.as<uint32_t>();
    sa.tau = c->tau.as<float>();
    sa.overflow = c->overflow.as<uint32_t>();
    sa.nq = nq;
    sa.nqt = nqt;
    sa.nsplit = nsplit;
    sa.bn = BN;
    sa.kprime = kprime;
    sa.n_dense = n_dense;
    sa.tau_only = 0;
    sa.floor_ord = nullptr;
    sa.floor_with_tau = 0;
    sa.trace = nullptr;
    // LDS key capacity: the dense boot stage needs exactly kprime + n_dense; candidate stages get
    // the full 8192 (64 KiB) so that only pathological emission counts overflow into the exact path.
    sa.lds_keys = dense ? next_pow2(kprime + n_dense) : SELECT_LDS_KEYS;
    if (sa.lds_keys > SELECT_LDS_KEYS) sa.lds_keys = SELECT_LDS_KEYS;
    if (dense && n_dense <= 4096 && kprime <= 64) sa.lds_keys = 0;  // boot stage: register-only path
    // Candidate stages: size the key buffer from the planner's expected per-query count (x8 head room,
    // >= 1024 keys) instead of always 64 KiB: at <= 16 KiB the workgroup fits beside a coarse
    // workgroup of the NEXT batch (141 KB of the CU's 160 KB), so the two overlap. More candidates
    // than the buffer holds only flags the query for the exact path (correct, slower).
    if (!dense && expected > 0) {
        // (extra_keys: entries that are in the lists for certain - the candidates of an emitting sample - on top of the estimate)
        const uint64_t want = next_pow2((uint32_t)std::min<uint64_t>(8 * expected + extra_keys + kprime, SELECT_LDS_KEYS));
        sa.lds_keys = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1024), SELECT_LDS_KEYS);
    }
    *lds_out = (size_t)sa.lds_keys * 8 + ((size_t)nsplit + 1) * 4;
    return sa;
}

int launch_select(SearchCtx* c, uint32_t nq, uint32_t nqt, uint32_t nsplit, uint32_t kprime,
                  const float* dense, uint32_t n_dense, hipStream_t s, uint64_t expected,
                  bool tau_only, uint32_t extra_keys) {
    size_t lds = 0;
    SelectArgs sa = make_select_args(c, nq, nqt, nsplit, kprime, dense, n_dense, expected, &lds, extra_keys);
    sa.tau_only = tau_only ? 1u : 0u;
    hipLaunchKernelGGL(select_kernel, dim3(nq), dim3(256), lds, s, sa);
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

// two device flag words -> the context's pinned, device-mapped mirror (plain stores, as publish_flags_kernel does)
__global__ void mirror_words_kernel(const uint32_t* __restrict__ dev, uint32_t* __restrict__ host, uint32_t w0, uint32_t w1) {
    if (threadIdx.x == 0) host[w0] = dev[w0];
    if (threadIdx.x == 1) host[w1] = dev[w1];
}

__global__ void iota_kernel(uint32_t* p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}


// Arrival counters of the kernels that finish a small search themselves (kernels_exact_small.h: [0, 64) per query, [64] finished
// queries; rescore_body: [65]): zeroed once, every kernel leaves them zero.
constexpr uint32_t XDONE_WORDS = EXACT_SMALL_MAX_Q + 16, XDONE_PUBLISH = EXACT_SMALL_MAX_Q + 1;
int ensure_xdone(SearchCtx* c, hipStream_t s) {
    if (c->xdone.bytes >= XDONE_WORDS * 4) return CGV_OK;
    if (int rc = c->xdone.ensure(XDONE_WORDS * 4)) return rc;
    HIPCHK(hipMemsetAsync(c->xdone.p, 0, c->xdone.bytes, s));
    return CGV_OK;
}
// Exact full scan for the queries in qlist_dev[0..nql) (device array of query slots).
// publish (optional): the caller wants the context's flag words published to the pinned mirror behind the scan (marker = the
// value of the F_DONE word); *publish is set to false when the scan's own kernel did it (kernels_exact_small.h).
int exact_search(cgv_index* h, SearchCtx* c, const uint32_t* qlist_dev, uint32_t nql, uint32_t k, uint64_t* out_idx,
                 float* out_score, hipStream_t s, int op, bool local_ids, bool* publish,
                 uint32_t marker) {
    const IdMap idmap = local_ids ? IdMap{0, 0, 1, 0, 0} : h->idmap;
    if (op < 0)
        op = (h->metric == CGV_METRIC_DOT) ? OP_DOT
             : (h->metric == CGV_METRIC_COSINE_SEQ ? OP_COSINE_SEQ : (h->metric == CGV_METRIC_COSINE_SCALAR ? OP_COSINE_SCALAR : OP_COSINE));
    const uint64_t n = h->n;
    const uint32_t K = next_pow2(std::max<uint32_t>(k, 2));
    uint64_t qg = std::max<uint64_t>(1, (512ull << 20) / (n * 4));
    qg = std::min<uint64_t>(qg, nql);
    qg = std::min<uint64_t>(qg, 65535);
    int rc;
    if ((rc = c->scores.ensure((size_t)qg * n * 4))) return rc;
    const uint32_t nch0 = (uint32_t)((n + TOPK_CHUNK - 1) / TOPK_CHUNK);
    if ((rc = c->keysA.ensure((size_t)qg * nch0 * K * 8))) return rc;
    if ((rc = c->keysB.ensure((size_t)qg * ((size_t)nch0 * K / TOPK_CHUNK + 1) * K * 8))) return rc;
    if (!qlist_dev && nql > qg) {   // several groups of queries: the kernels index a group's queries through a list
        if ((rc = c->qlist.ensure((size_t)nql * 4))) return rc;
        hipLaunchKernelGGL(iota_kernel, dim3((nql + 255) / 256), dim3(256), 0, s, c->qlist.as<uint32_t>(), nql);
        qlist_dev = c->qlist.as<uint32_t>();
    }
    // A few queries (the trait-level call is ONE): scores, both reductions and the results in ONE kernel (kernels_exact_small.h)
    {
        const uint32_t G = (uint32_t)std::min<uint64_t>(256, (n + 31) / 32);   // workgroups per query: 32 rows per pass each
        // (up to 8 queries: measured on config 1's corpus, 32 queries per call take 141 us in this form against 123 staged -
        // 32 merging workgroups and 32 KiB of LDS per scanning workgroup - while 8 are equal and one is 25 us faster)
        if (K <= 64 && nql <= 8 && G >= 1 && n <= (uint64_t)EXACT_SMALL_ROWS * G && n < (1ull << 32) && exact_small_enabled()) {
            if ((rc = c->keysA.ensure((size_t)nql * G * K * 8))) return rc;
            if ((rc = ensure_xdone(c, s))) return rc;
            ExactSmallArgs xa;
            xa.rows = h->rows;
            xa.qrows = c->qrows.as<char>();
            xa.qlist = qlist_dev;
            xa.nql = nql;
            xa.n = (uint32_t)n;
            xa.D = h->D;
            xa.ld = h->ld;
            xa.op = op;
            xa.K = K;
            xa.k = k;
            xa.part = c->keysA.as<uint64_t>();
            xa.done = c->xdone.as<uint32_t>();
            xa.idmap = idmap;
            xa.out_idx = out_idx;
            xa.out_score = out_score;
            xa.nan_flag = c->flags + F_NAN;
            xa.pub_flags = c->flags;
            xa.pub_host = (publish && *publish) ? c->h_flags_dev : nullptr;
            xa.pub_n = (uint32_t)F_COUNT;
            xa.pub_done_word = (uint32_t)F_DONE;
            xa.pub_marker = marker;
            if (xa.pub_host) *publish = false;
            switch (h->dtype) {
                case CGV_DTYPE_F32: hipLaunchKernelGGL(exact_small_kernel<DT_F32>, dim3(G, nql), dim3(256), 0, s, xa); break;
                case CGV_DTYPE_BF16: hipLaunchKernelGGL(exact_small_kernel<DT_BF16>, dim3(G, nql), dim3(256), 0, s, xa); break;
                case CGV_DTYPE_FP16: hipLaunchKernelGGL(exact_small_kernel<DT_FP16>, dim3(G, nql), dim3(256), 0, s, xa); break;
                case CGV_DTYPE_FP8E4M3: hipLaunchKernelGGL(exact_small_kernel<DT_FP8>, dim3(G, nql), dim3(256), 0, s, xa); break;
                default: return fail(CGV_ERR_INTERNAL, "exact path: unsupported dtype");
            }
            HIPCHK(hipGetLastError());
            return CGV_OK;
        }
    }
    for (uint32_t q0 = 0; q0 < nql; q0 += (uint32_t)qg) {
        const uint32_t g = (uint32_t)std::min<uint64_t>(qg, nql - q0);
        const uint32_t* ql = qlist_dev ? qlist_dev + q0 : nullptr;   // NULL (one group): the queries 0 .. nql - 1 themselves
        float* sc = c->scores.as<float>();
        switch (h->dtype) {
            case CGV_DTYPE_F32: launch_exact_scores<DT_F32>(h, c, ql, g, sc, op, s); break;
            case CGV_DTYPE_BF16: launch_exact_scores<DT_BF16>(h, c, ql, g, sc, op, s); break;
            case CGV_DTYPE_FP16: launch_exact_scores<DT_FP16>(h, c, ql, g, sc, op, s); break;
            case CGV_DTYPE_FP8E4M3: launch_exact_scores<DT_FP8>(h, c, ql, g, sc, op, s); break;
            default: return fail(CGV_ERR_INTERNAL, "exact path: unsupported dtype");
        }
        uint64_t* cur = c->keysA.as<uint64_t>();
        uint64_t* nxt = c->keysB.as<uint64_t>();
        uint32_t nch = nch0;
        // K <= 64 (every k <= 64: the reference's limits are 10-100): register-resident extraction instead of a 4096-key LDS
        // bitonic sort per chunk (kernels_select.h) - the chunk reductions are the longest kernels of a single-query call
        const bool small_k = K <= 64;
        if (small_k)
            hipLaunchKernelGGL(topk_chunk_small_kernel, dim3(nch, g), dim3(256), 0, s, sc, (const uint64_t*)nullptr,
                               (uint32_t)n, K, cur, c->flags + F_NAN);
        else
            hipLaunchKernelGGL(topk_chunk_kernel, dim3(nch, g), dim3(256), 0, s, sc, (const uint64_t*)nullptr,
                               (uint32_t)n, K, cur, c->flags + F_NAN);
        while (nch > 1) {
            const uint32_t M = nch * K;
            const uint32_t nch2 = (M + TOPK_CHUNK - 1) / TOPK_CHUNK;
            if (small_k)
                hipLaunchKernelGGL(topk_chunk_small_kernel, dim3(nch2, g), dim3(256), 0, s, (const float*)nullptr,
                                   (const uint64_t*)cur, M, K, nxt, c->flags + F_NAN);
            else
                hipLaunchKernelGGL(topk_chunk_kernel, dim3(nch2, g), dim3(256), 0, s, (const float*)nullptr,
                                   (const uint64_t*)cur, M, K, nxt, c->flags + F_NAN);
            std::swap(cur, nxt);
            nch = nch2;
        }
        hipLaunchKernelGGL(emit_topk_kernel, dim3((g * k + 255) / 256), dim3(256), 0, s, (const uint64_t*)cur,
                           K, k, ql, g, idmap, out_idx, out_score);
        HIPCHK(hipGetLastError());
    }
    return CGV_OK;
}

__global__ void pad_out_kernel(uint64_t* idx, float* sc, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        idx[i] = UINT64_MAX;
        sc[i] = -INFINITY;
    }
}

// Searches in flight per DEVICE, over every handle of this process (a cgv_sharded handle keeps several cgv_index on one
// device): the fused sample + emit launch (COARSE_EMIT_BOOT) holds its workgroups at a rendezvous and wants the device to
// itself, so only a search that finds the device idle takes that form (search_enqueue); the others use the three-launch form,
// which never waits. (Best effort - another process is invisible here; the rendezvous is bounded for that reason.)
#ifdef CGV_ABLATE_BUILD   // (the production library has no fused launch: no counter, no bookkeeping - ADVICE r4)
constexpr int MAX_DEVICES = 64;
std::atomic<int> g_dev_inflight[MAX_DEVICES];
void dev_inflight_add(const cgv_index* h, int d) {
    if (h->device >= 0 && h->device < MAX_DEVICES) g_dev_inflight[h->device].fetch_add(d, std::memory_order_relaxed);
}
int dev_inflight(const cgv_index* h) {
    return (h->device >= 0 && h->device < MAX_DEVICES) ? g_dev_inflight[h->device].load(std::memory_order_relaxed) : 2;
}
#endif

template <int DT>
void launch_boot(cgv_index* h, SearchCtx* c, uint32_t n_boot, uint32_t nq, float* dense, BootMap bmap, hipStream_t s) {
    const uint32_t nrb = (n_boot + 63) / 64, nqb = (nq + 63) / 64;
    hipLaunchKernelGGL(boot_kernel<DT>, dim3(nrb * nqb), dim3(64), 0, s,
                       (const char*)(h->shadow ? h->srows : h->rows),
                       (const char*)(h->shadow ? c->qshadow.p : c->qrows.p), (const float*)h->invn,
                       (const float*)c->qinvn.p, n_boot, nq, h->shadow ? h->lds : h->ld, h->metric, dense, bmap, (uint32_t)h->n,
                       (const int8_t*)(DT == DT_FP8 ? h->rexp : nullptr), (const int8_t*)(DT == DT_FP8 ? c->qrexp.p : nullptr));
}

// Enqueue one batch on the context's stream (no host synchronisation); search_finish() completes it.
// Caller holds h->mu and owns the context.
int search_enqueue(cgv_index* h, SearchCtx* c, const float* qdev, uint32_t nq, uint32_t k, uint64_t* out_idx,
                   float* out_score) {
    hipStream_t s = c->cur();
    int rc;
    h->st.searches++;
    h->st.queries += nq;
    c->nq = nq;
    c->k = k;
    c->out_idx = out_idx;
    c->out_score = out_score;
    c->mfma = false;
    c->timed_coarse = false;
    c->coarse_rows = 0;
    c->kprime = 0;
    c->published = false;
    c->boot_used = false;
    c->top2 = false;
    c->exact_enqueued = false;
    // flag + pacing words: zero after a search that ran to completion (its last kernel resets them), else cleared here
    if (!c->flags_clean) HIPCHK(hipMemsetAsync(c->flags, 0, CTX_FLAG_WORDS * 4, s));
    c->flags_clean = false;
    if (h->n == 0) {
        uint64_t tot = (uint64_t)nq * k;
        hipLaunchKernelGGL(pad_out_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, out_idx,
                           out_score, tot);
        HIPCHK(hipGetLastError());
        c->mfma = true;  // nothing left to do in search_finish
        HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, F_COUNT * 4, hipMemcpyDeviceToHost, s));
        return CGV_OK;
    }

    if (h->profiling > 1) HIPCHK(hipEventRecord(c->ev[0], s));
    // --- queries: round to storage dtype, norms ---
    if ((rc = c->qrows.ensure(storage_bytes(h, nq)))) return rc;  // whole 256-query tiles (DMA reads them)
    if ((rc = c->qnorm.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->qinvn.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->qrexp.ensure((size_t)nq + 16))) return rc;
    if ((rc = c->fbflag.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->qlist.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->nbest.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->overflow.ensure((size_t)nq * 4))) return rc;
    rc = prep_dispatch(h->dtype, qdev, nq, h->D, h->ld, 0, c->qrows.as<char>(), c->qnorm.as<float>(),
                       c->qinvn.as<float>(), c->qrexp.as<int8_t>(), c->flags + F_NONFINITE_Q, s,
                       c->nbest.as<uint32_t>(), c->overflow.as<uint32_t>());  // also clears nbest / overflow
    if (rc) return rc;
    if (h->shadow) {  // bf16 copy of the queries + their rounding residuals
        if ((rc = c->qshadow.ensure(shadow_bytes(h, nq)))) return rc;
        if ((rc = c->qres.ensure((size_t)nq * 8))) return rc;
        // second pass over the f32 COPY prep just made in device memory (row stride ld, zero padded - the sums do not
        // change), not over the caller's buffer: that one may be pinned host memory, read over PCIe
        hipLaunchKernelGGL(shadow_rows_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, (const float*)c->qrows.as<float>(), (uint64_t)nq,
                           h->ld, h->lds, (uint64_t)0, c->qshadow.as<char>(), c->qnorm.as<float>(), c->qinvn.as<float>(),
                           c->qres.as<float>(), (uint32_t*)nullptr);
        HIPCHK(hipGetLastError());
    }

    if (h->profiling > 2) HIPCHK(hipEventRecord(c->pev[0], s));
    // f32 + shadow: the coarse scores carry bf16 rounding error (~2e-3), so more candidates are re-scored
    const uint32_t kprime = h->shadow ? std::min<uint32_t>(((4 * k + 16 + 7) / 8) * 8, 256u) : kprime_of(k, h->dtype == CGV_DTYPE_FP8E4M3);
    const bool mfma = !h->force_exact && !h->wide_range && (h->dtype != CGV_DTYPE_F32 || h->shadow) && kprime <= CAND_CAPS &&
                      (!h->shadow || k <= 60);
    c->mfma = mfma;
    c->kprime = mfma ? kprime : 0u;

    if (mfma) {
        const uint32_t nqt = (nq + BN - 1) / BN;
        const uint32_t nsplit_max = std::max<uint32_t>(1u, (uint32_t)h->n_cu / nqt);
        const int cdt = h->shadow ? CGV_DTYPE_BF16 : h->dtype;  // dtype the coarse pass runs in
        // threshold ladder (kernels_coarse.h): the bf16 / fp16 emitting kernel tightens its thresholds inside the launch
        const int lad_mode = (cdt == CGV_DTYPE_BF16 || cdt == CGV_DTYPE_FP16) ? tun().ladder : 0;
        StagePlan p = plan_stages(h->n, kprime, nqt, (uint32_t)h->n_cu, nsplit_max, lad_mode >= 2 ? 1 : 0,
                                  /* allow_emit = */ lad_mode == 0 && tun().fuse_sample == 0 && nq > 64);
        // (an emitting sample's lists are continued by the first emitting launch: that one must have a workgroup per sampled tile)
        if (p.sample_emits && (p.counts.empty() || std::min<uint32_t>(p.counts[0], nsplit_max) < std::min<uint32_t>(p.sample_tiles, nsplit_max)))
            p = plan_stages(h->n, kprime, nqt, (uint32_t)h->n_cu, nsplit_max, lad_mode >= 2 ? 1 : 0, false);
        // emitting launches: workgroups per query tile. Measurement flavour, knob `epi` bit 10: two 4-wave workgroups per CU
        // (experiments/kernels_coarse_wg2.h; the select kernels take up to 256 lists per query)
        uint32_t nsplit_emit = nsplit_max;
#ifdef CGV_ABLATE_BUILD
        {
            const uint32_t kc_ = (h->shadow ? h->lds : h->ld) / kchunk_of(cdt);
            if ((tun().epi & 1024) != 0 && nqt >= 2 && (cdt == CGV_DTYPE_BF16 || cdt == CGV_DTYPE_FP16) && kc_ >= 3 && kc_ % 3 == 0 &&
                2 * nsplit_max <= 256)
                nsplit_emit = 2 * nsplit_max;
        }
#endif
        const uint32_t Wmax = nqt * nsplit_emit;
        if ((rc = c->tau.ensure((size_t)nq * 4))) return rc;
        if ((rc = c->nbest.ensure((size_t)nq * 4))) return rc;
        if ((rc = c->overflow.ensure((size_t)nq * 4))) return rc;
        if ((rc = c->best.ensure((size_t)nq * kprime * 8))) return rc;
        if ((rc = c->cand.ensure((size_t)Wmax * BN * CAND_CAPS * 8))) return rc;
        if ((rc = c->candcnt.ensure((size_t)Wmax * BN * 4))) return rc;

        CoarseArgs a;
        a.rows = h->shadow ? h->srows : h->rows;
        a.qrows = h->shadow ? c->qshadow.as<char>() : c->qrows.as<char>();
        a.invn_c = h->invn;
        a.invn_q = c->qinvn.as<float>();
        a.blk_min = h->blk_min;
        a.blk_max = h->blk_max;
        a.tau = c->tau.as<float>();
        a.cand = c->cand.as<uint2>();
        a.cand_cnt = c->candcnt.as<uint32_t>();
        a.overflow = c->overflow.as<uint32_t>();
        a.dump = nullptr;
        a.sample_ld = 0;
        a.sample_vals = 16;
        a.epi = (uint32_t)tun().epi;
        a.n = (uint32_t)h->n;
        a.nq = nq;
        a.ld = h->shadow ? h->lds : h->ld;
        a.kc = a.ld / kchunk_of(cdt);
        a.T1 = 0;  // the visiting order covers all tiles (the first threshold comes from a sample)
        a.R = p.R;
        a.P = p.P;
        a.nqt = nqt;
        a.metric = h->metric;
        a.qgroup = query_group(nqt, a.ld, cdt);
        a.rexp_c = h->rexp;
        a.rexp_q = c->qrexp.as<int8_t>();
        const uint32_t* pace_words = (Wmax <= PACE_WORDS && tun().pace) ? c->flags + F_COUNT : nullptr;
        a.pace = const_cast<uint32_t*>(pace_words);

        // Fused form (round 4, MEASUREMENT FLAVOUR ONLY): the first emitting launch takes its own first threshold from the first
        // tile of every workgroup (COARSE_EMIT_BOOT, kernels_coarse.h) - no sample launch, no tau_kernel, the sample tiles scored
        // once. Measured break-even at best against the three launches it replaces (HISTORY.md §9.1, profiles/r04_fused_launch_ab.txt:
        // a rendezvous right behind the first tile exposes the launch's ramp skew, and 128 accumulators per lane held across it
        // either spill or, stashed, cost more L2 / HBM traffic than scoring the tile again), so the production library does not
        // carry it; knob `fuse_sample` of scripts/ab.py.
        const uint32_t nsplit0 = p.counts.empty() ? 0u : std::min<uint32_t>(p.counts[0], nsplit_max);
        const uint32_t fvals = sample_vals_of(std::max<uint32_t>(nsplit0, 1u));
        const bool can_fuse = p.sample_tiles > 0 && nsplit0 > 0 && (cdt == CGV_DTYPE_BF16 || cdt == CGV_DTYPE_FP16) && a.kc >= 4 &&
                              a.kc % 4 == 0 && nqt > 1 && nqt * 4u <= BOOT_WORDS && nsplit0 * fvals >= 4u * kprime &&
                              nsplit0 <= SAMPLE_TILES_MAX && p.counts[0] >= 2 * nsplit0;  // (every workgroup walks >= 2 tiles)
#ifdef CGV_ABLATE_BUILD
        const bool fuse = can_fuse && tun().fuse_sample > 0 && dev_inflight(h) <= 1;  // (this search is the one in flight)
#else
        const bool fuse = false;
        (void)can_fuse;
#endif
        c->boot_used = fuse;
        a.tau_out = c->tau.as<float>();
        a.boot_sync = c->flags + F_COUNT + PACE_WORDS;
        a.kprime = kprime;
        a.floor_ord = nullptr;
        a.cellb = nullptr;
        a.sample_emit = 0;
        a.sample_floor = 0;
        a.scand = nullptr;
        a.append_splits = 0;
        c->repair = false;
        a.lad = nullptr;
        a.ladc = nullptr;
        const float* sfloor_p = nullptr;   // emitting sample: the floor area of its dump rows (final_kernel's in-kernel repair)
        uint32_t sfloor_ld = 0, sfloor_n = 0;
        // Small batches (one query tile of <= 64 queries - the trait-level call is ONE query, traits.rs:14): the corpus streams at the
        // HBM rate whatever happens to the scores, so the staged thresholds (sample launch, tau_kernel, emitting launches, select)
        // are pure latency. COARSE_TOP2 visits all tiles in ONE launch without a threshold: per-cell top-2 + floor (kernels_coarse.h).
        constexpr uint32_t TOP2_MAX_NQ = 64;
        const bool top2 = tun().top2 != 0 && nq <= TOP2_MAX_NQ && kprime <= 64 && p.ntiles > BOOT_TILES && !fuse;
        c->top2 = top2;
        uint32_t top2_nsplit = 0;
        if (top2) {
            if (c->floor.bytes < (size_t)TOP2_MAX_NQ * 4) c->floor_clean = false;
            if ((rc = c->floor.ensure((size_t)TOP2_MAX_NQ * 4))) return rc;
            if (!c->floor_clean) HIPCHK(hipMemsetAsync(c->floor.p, 0, (size_t)TOP2_MAX_NQ * 4, s));
            c->floor_clean = false;   // (true again once this search's final kernel has run: search_finish)
            a.floor_ord = c->floor.as<uint32_t>();
            if (nq > TOP2_QPW) {   // more than 16 queries: spread over the four wave columns (kernels_coarse.h: top2_col_of)
                const size_t tile_bytes = (size_t)a.kc * BLOCK_BYTES;
                if (c->qspread.bytes < tile_bytes) {
                    if ((rc = c->qspread.ensure(tile_bytes))) return rc;
                    HIPCHK(hipMemsetAsync(c->qspread.p, 0, c->qspread.bytes, s));   // (columns without a query: zeros, once)
                }
                hipLaunchKernelGGL(top2_spread_queries_kernel, dim3((nq * a.kc * 4u + 255u) / 256u), dim3(256), 0, s, a.qrows,
                                   c->qspread.as<char>(), nq, a.kc);
                HIPCHK(hipGetLastError());
                a.qrows = c->qspread.as<char>();
            }
            a.j0 = 0;
            a.cnt = p.ntiles;
            a.nsplit = std::min<uint32_t>(p.ntiles, nsplit_max);
            a.pace = nullptr;
            top2_nsplit = a.nsplit;
            // every cell's left-out score, for the cheap way out of a floor violation (kernels_repair.h)
            c->repair = tun().top2_repair != 0;
            if (c->repair) {
                if ((rc = c->cellb.ensure((size_t)4 * a.nsplit * 64 * 4))) return rc;
                if ((rc = c->reptheta.ensure((size_t)TOP2_MAX_NQ * 4))) return rc;
                if ((rc = c->repkeys.ensure((size_t)TOP2_MAX_NQ * REPAIR_KEYS * 8))) return rc;
                if ((rc = c->repn.ensure((size_t)TOP2_MAX_NQ * 4 + 16))) return rc;
                if ((rc = c->repnew.ensure((size_t)TOP2_MAX_NQ * REPAIR_NEW_KEYS * 8))) return rc;
                if (c->repcnt.bytes == 0) {   // scan -> merge hand-over words: zero when idle (the merge kernel clears what it read)
                    if ((rc = c->repcnt.ensure((size_t)TOP2_MAX_NQ * 2 * 4))) return rc;
                    HIPCHK(hipMemsetAsync(c->repcnt.p, 0, c->repcnt.bytes, s));
                }
                // (rows too long to stage even one beside the query row: the exact scan takes floor violations, as before round 6)
                if ((size_t)2 * h->ld * h->esize + 16 > REPAIR_LDS_BYTES) c->repair = false;
                a.cellb = c->repair ? c->cellb.as<float>() : nullptr;
                c->t2_nsplit = a.nsplit;
                c->t2_cnt = a.cnt;
                c->t2_R = a.R;
                c->t2_P = a.P;
            }
            if (h->profiling > 2) HIPCHK(hipEventRecord(c->pev[1], s));   // (no threshold phase: the one launch counts as emitting)
            // (the timed launch carries its own event pair: the dispatch's start / end, no marker packets on the stream - coarse_launch.h)
            const bool ext_ev = h->profiling && tun().launch_events != 0, rec_ev = h->profiling && !ext_ev;   // (A/B: marker events)
            if (rec_ev) HIPCHK(hipEventRecord(c->ev[1], s));
            if ((rc = launch_coarse(cdt, COARSE_TOP2, a, a.nsplit, s, ext_ev ? c->ev[1] : nullptr, ext_ev ? c->ev[2] : nullptr))) return rc;
            if (rec_ev) HIPCHK(hipEventRecord(c->ev[2], s));
            if (h->profiling) {
                c->timed_coarse = true;
                c->coarse_rows = h->n;
            }
        } else if (fuse) {
            if ((rc = c->dump.ensure((size_t)nq * nsplit0 * fvals * 4))) return rc;

        } else if (p.sample_tiles > 0) {
            // first threshold: block maxima of the first sample_tiles tiles of the visiting order -> k'-th largest
            const uint32_t vals = sample_vals_of(p.sample_tiles);
            const uint32_t M = p.sample_tiles * vals;
            // an emitting sample also leaves, per query, the best score each of its 4 cells per tile kept out of the lists
            const uint32_t floor_n = p.sample_emits ? 4u * p.sample_tiles : 0u, dump_ld = M + floor_n;
            if ((rc = c->dump.ensure((size_t)nq * dump_ld * 4))) return rc;
            if (p.sample_emits && (rc = c->floor.ensure((size_t)std::max<uint32_t>(nq, 64u) * 4))) return rc;
            if (p.sample_emits && (rc = c->scand.ensure((size_t)nqt * std::min<uint32_t>(p.sample_tiles, nsplit_max) * BN * 8 * 8))) return rc;
            a.scand = p.sample_emits ? c->scand.as<uint2>() : nullptr;   // (the first emitting launch reads what the sample wrote)
            CoarseArgs sa = a;
            sa.dump = c->dump.as<float>();
            sa.pace = nullptr;
            sa.sample_ld = dump_ld;
            sa.sample_vals = vals;
            sa.sample_emit = p.sample_emits ? 1u : 0u;
            sa.sample_floor = M;
            if (p.sample_emits) {
                sfloor_p = c->dump.as<float>() + M;
                sfloor_ld = dump_ld;
                sfloor_n = floor_n;
            }
            sa.j0 = 0;
            sa.cnt = p.sample_tiles;
            sa.nsplit = std::min<uint32_t>(p.sample_tiles, nsplit_max);
            if ((rc = launch_coarse(cdt, COARSE_SAMPLE, sa, nqt * sa.nsplit, s))) return rc;
            const uint32_t hi_rank = std::max<uint32_t>(1u, kprime / 4u);
            const float lad_scale = lad_mode ? plan_ladder_scale((double)p.sample_tiles * BM, (double)p.ntiles * BM, kprime, hi_rank) : 0.0f;
            // (the ladder lives in the ring-unrolled instantiations: K a multiple of 4 chunks - every headline shape)
            const bool lad_on = lad_mode != 0 && lad_scale > 0.0f && a.kc >= 4 && a.kc % 4 == 0 && (a.epi & 25u) == 1u;
            if (lad_on) {
                if ((rc = c->lad.ensure((size_t)nqt * BN * 8))) return rc;
                if ((rc = c->ladc.ensure((size_t)nqt * BN * 16))) return rc;
                a.lad = c->lad.as<unsigned long long>();
                a.ladc = c->ladc.as<float4>();
            }
            hipLaunchKernelGGL(tau_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, (const float*)c->dump.as<float>(), M, dump_ld, nq,
                               kprime, c->tau.as<float>(), c->nbest.as<uint32_t>(), lad_on ? c->ladc.as<float4>() : (float4*)nullptr,
                               lad_on ? c->lad.as<unsigned long long>() : (unsigned long long*)nullptr, lad_scale, hi_rank, floor_n,
                               p.sample_emits ? c->floor.as<uint32_t>() : (uint32_t*)nullptr);
            HIPCHK(hipGetLastError());
        } else {
            // boot rows: a sample of 32-row groups when coarse launches follow (the ragged last group may be one of
            // them: its missing rows score -inf), else the whole corpus
            const bool sampled = !p.counts.empty();
            const uint32_t n_boot = sampled ? p.T1 * BM : (uint32_t)h->n;
            const uint32_t ngroups = (uint32_t)((h->n + 31) / 32);
            const BootMap bmap{sampled ? golden_stride(ngroups) : 1u, std::max<uint32_t>(ngroups, 1u)};
            if ((rc = c->dump.ensure((size_t)nq * n_boot * 4))) return rc;
            // boot: dense scores of the n_boot boot rows -> top-k' -> first tau
            if (h->dtype == CGV_DTYPE_BF16 || h->shadow)
                launch_boot<DT_BF16>(h, c, n_boot, nq, c->dump.as<float>(), bmap, s);
            else if (h->dtype == CGV_DTYPE_FP16)
                launch_boot<DT_FP16>(h, c, n_boot, nq, c->dump.as<float>(), bmap, s);
            else
                launch_boot<DT_FP8>(h, c, n_boot, nq, c->dump.as<float>(), bmap, s);
            HIPCHK(hipGetLastError());
            if ((rc = launch_select(c, nq, nqt, 0, kprime, c->dump.as<float>(), n_boot, s, 0, sampled))) return rc;
        }
        if (h->profiling > 2 && !top2) HIPCHK(hipEventRecord(c->pev[1], s));
        // an emitting sample covered the first sample_tiles positions of the visiting order: the launches start behind it, and
        // the first of them continues the sample's candidate lists
        const bool semit = p.sample_emits && !top2;
        uint32_t j0 = semit ? p.sample_tiles : 0u;
        if (semit) c->floor_clean = false;   // (the floor words now hold this search's sample floors, not zeros)
        const bool fused_final = kprime <= 64 && !p.counts.empty();  // extraction path of select (k' <= 64)
        uint32_t last_nsplit = top2_nsplit;
        uint64_t last_expected = top2_nsplit;  // (TOP2: <= 8 candidates per (workgroup, query) list; x 8 head room in make_select_args)
        uint32_t last_extra = 0;
        for (size_t st = 0; st < p.counts.size() && !top2; ++st) {
            const uint32_t cnt = p.counts[st];
            a.j0 = j0;
            a.cnt = cnt;
            a.nsplit = std::min<uint32_t>(cnt, nsplit_emit);
            a.append_splits = (semit && st == 0) ? std::min<uint32_t>(p.sample_tiles, nsplit_max) : 0u;
            const uint32_t extra_keys = (semit && st == 0) ? 4u * kprime : 0u;   // the sample's rows above the first threshold: ~k' per query
            // Soft lockstep pays where the workgroups of a group can drift apart: launches of 100+ tiles per workgroup
            // (C5: 1000+). On short walks (C2: 55 tiles) the group stays together by itself and the per-tile poll only
            // costs (r03b: C2 step 1.458 -> 1.453 ms, the 125 k-row shard 0.381 -> 0.378 without it).
            a.pace = (cnt / a.nsplit >= 128u || tun().pace > 1) ? const_cast<uint32_t*>(pace_words) : nullptr;
            const bool dominant = (st + 1 == p.counts.size());
            const bool timed_any = h->profiling && dominant;
            const bool timed = timed_any && tun().launch_events != 0;   // the dispatch's own start / end events (coarse_launch.h)
            if (timed_any && !timed) HIPCHK(hipEventRecord(c->ev[1], s));   // (A/B: marker events in front of and behind the launch)
            if (st == 0 && fuse) {
                CoarseArgs fa = a;
                fa.dump = c->dump.as<float>();
                fa.sample_vals = fvals;
                fa.sample_ld = nsplit0 * fvals;
                fa.pace = nullptr;
                if ((rc = launch_coarse(cdt, COARSE_EMIT_BOOT, fa, nqt * fa.nsplit, s, timed ? c->ev[1] : nullptr, timed ? c->ev[2] : nullptr))) return rc;
            } else if ((rc = launch_coarse(cdt, COARSE_EMIT, a, nqt * a.nsplit, s, timed ? c->ev[1] : nullptr, timed ? c->ev[2] : nullptr))) {
                return rc;
            }
            if (timed_any) {
                if (!timed) HIPCHK(hipEventRecord(c->ev[2], s));
                c->timed_coarse = true;
                c->coarse_rows = std::min<uint64_t>((uint64_t)cnt * BM, h->n);
            }
            // expected emissions per query of this launch: k' * rows / rows seen before it
            uint64_t expected = (uint64_t)kprime * cnt / std::max<uint64_t>(1, semit ? (uint64_t)j0 : (uint64_t)(p.sample_tiles ? p.sample_tiles : p.T1) + j0) + 1;
            if (a.lad) {   // the ladder tightens inside the launch: ~k' ln(rows / seen), doubled for the width of its levels
                const double seen_t = (double)std::max<uint64_t>(1, (uint64_t)p.sample_tiles + j0);
                expected = std::min<uint64_t>(expected, (uint64_t)(2.0 * kprime * (log(((double)cnt + seen_t) / seen_t) + 1.0)) + 1);
            }
            if (dominant && fused_final) {  // the last selection happens inside final_kernel
                last_nsplit = a.nsplit;
                last_expected = expected;
                last_extra = extra_keys;
            } else if ((rc = launch_select(c, nq, nqt, a.nsplit, kprime, nullptr, 0, s, expected, false, extra_keys))) {
                return rc;
            }
            j0 += cnt;
        }
        if (h->profiling > 2) HIPCHK(hipEventRecord(c->pev[2], s));
        RescoreArgs r;
        r.best = c->best.as<uint64_t>();
        r.nbest = c->nbest.as<uint32_t>();
        r.tau = c->tau.as<float>();
        r.rows = h->rows;
        r.qrows = c->qrows.as<char>();
        r.norm_q = c->qnorm.as<float>();
        r.overflow = c->overflow.as<uint32_t>();
        r.out_idx = out_idx;
        r.out_score = out_score;
        r.fb_flag = c->fbflag.as<uint32_t>();
        r.fb_count = c->flags + F_FB_COUNT;
        r.stat_maxerr = c->flags + F_MAXERR;
        r.idmap = h->idmap;
        r.nq = nq;
        r.n = (uint32_t)h->n;
        r.D = h->D;
        r.ld = h->ld;
        r.kprime = kprime;
        r.k = k;
        r.metric = h->metric;
        // K = 16 for every dtype: the boot stage scores its rows with v_mfma_f32_32x32x16_* (fp8 included),
        // and (ld/16)*17 >= (ld/64)*65 covers the K = 64 block-scaled instruction of the fp8 main kernel
        r.eps_scale = coarse_eps_scale(h->shadow ? h->lds : h->ld, h->ld, 16u, h->metric, h->dtype == CGV_DTYPE_FP8E4M3);
        r.max_norm_c = h->max_norm_c;
        r.qres = h->shadow ? c->qres.as<float>() : nullptr;
        r.res_rel_c = h->res_rel_c;
        r.res_abs_c = h->res_abs_c;
        r.stat_maxeps = c->flags + F_MAXEPS;
        r.rep_theta = (top2 && c->repair) ? c->reptheta.as<float>() : nullptr;
        r.rep_keys = c->repkeys.as<uint64_t>();
        r.rep_n = c->repn.as<uint32_t>();
        r.sfloor = (semit && tun().sample_repair != 0) ? sfloor_p : nullptr;
        r.sfloor_ld = sfloor_ld;
        r.sfloor_n = sfloor_n;
        r.s_T1 = a.T1;
        r.s_R = a.R;
        r.s_P = a.P;
        r.sfix_count = c->flags + F_COMPACT;   // (a free word while the pipeline runs: published and cleared with the others)
        if ((rc = c->qstat.ensure((size_t)nq * 8))) return rc;
        r.qstat = c->qstat.as<uint2>();   // per-query statistics, folded into the flag words by publish_flags_kernel
        // ... or, for a small batch, by the last workgroup of the final kernel itself (RescoreArgs::pub_*)
        const bool self_publish = nq <= 64 && !c->boot_used && tun().self_publish;
        r.pub_flags = c->flags;
        r.pub_host = nullptr;
        r.pub_count = nullptr;
        r.pub_words = (uint32_t)F_COUNT | ((uint32_t)F_DONE << 8) | ((uint32_t)F_MAXERR << 16) | ((uint32_t)(h->shadow ? F_MAXEPS : F_COUNT) << 24);
        if (self_publish) {
            if ((rc = ensure_xdone(c, s))) return rc;
            r.pub_host = c->h_flags_dev;
            r.pub_count = c->xdone.as<uint32_t>() + XDONE_PUBLISH;
        }
        c->h_flags[F_DONE] = 0;  // (no kernel of this context is in flight: the host may write its mirror)
        c->published = true;     // publish_flags_kernel behind the last kernel, below
#ifdef CGV_ABLATE_BUILD
        static const bool tracing = getenv("CGV_TRACE") != nullptr;  // diagnostics: phase stamps of the final kernel
#else
        constexpr bool tracing = false;
#endif
        r.trace = nullptr;
        if (tracing) {
            if ((rc = c->trace.ensure((size_t)nq * 64))) return rc;
            r.trace = c->trace.as<uint64_t>();
        }
        c->eps = r.eps_scale;
        {
            const size_t rowb = (size_t)h->ld * h->esize, pitch = rowb + 16;
            // LDS for staged rows: small enough (with the query row) to fit beside a coarse workgroup of the
            // next batch in flight; k' candidates then take one or two passes
            // all k' rows in ONE pass when they fit 40 KB (k' = 16 at D = 768: 25 KB; r03a: two passes of 9 + 7 rows cost
            // 14 us per query); several passes beyond
            const size_t budget = h->shadow ? 45 * 1024 : 40 * 1024;
            uint32_t rpb = (uint32_t)std::max<size_t>(1, std::min<size_t>(kprime, budget / pitch));
            r.rows_per_batch = rpb;
            // [work region][query row]: the query row is fetched first and must survive the selection's key buffer
            size_t work = (size_t)rpb * pitch;
            if (fused_final) {
                size_t sel_lds = 0;
                SelectArgs sa = make_select_args(c, nq, nqt, last_nsplit, kprime, nullptr, 0, last_expected, &sel_lds, last_extra);
                sa.floor_ord = (top2 || semit) ? c->floor.as<uint32_t>() : nullptr;
                sa.floor_with_tau = semit ? 1u : 0u;
                work = (std::max(work, sel_lds) + 15) / 16 * 16;
                const uint32_t qoff = (uint32_t)work;
                const size_t lds = work + rowb;
                if (h->dtype == CGV_DTYPE_F32)
                    hipLaunchKernelGGL(final_kernel<DT_F32>, dim3(nq), dim3(256), lds, s, sa, r, qoff);
                else if (h->dtype == CGV_DTYPE_BF16)
                    hipLaunchKernelGGL(final_kernel<DT_BF16>, dim3(nq), dim3(256), lds, s, sa, r, qoff);
                else if (h->dtype == CGV_DTYPE_FP16)
                    hipLaunchKernelGGL(final_kernel<DT_FP16>, dim3(nq), dim3(256), lds, s, sa, r, qoff);
                else
                    hipLaunchKernelGGL(final_kernel<DT_FP8>, dim3(nq), dim3(256), lds, s, sa, r, qoff);
            } else {
                work = (work + 15) / 16 * 16;
                const uint32_t qoff = (uint32_t)work;
                const size_t lds = work + rowb;
                if (h->dtype == CGV_DTYPE_F32)
                    hipLaunchKernelGGL(rescore_kernel<DT_F32>, dim3(nq), dim3(256), lds, s, r, qoff);
                else if (h->dtype == CGV_DTYPE_BF16)
                    hipLaunchKernelGGL(rescore_kernel<DT_BF16>, dim3(nq), dim3(256), lds, s, r, qoff);
                else if (h->dtype == CGV_DTYPE_FP16)
                    hipLaunchKernelGGL(rescore_kernel<DT_FP16>, dim3(nq), dim3(256), lds, s, r, qoff);
                else
                    hipLaunchKernelGGL(rescore_kernel<DT_FP8>, dim3(nq), dim3(256), lds, s, r, qoff);
            }
        }
        HIPCHK(hipGetLastError());
        if (!self_publish) {
            hipLaunchKernelGGL(publish_flags_kernel, dim3(1), dim3(256), 0, s, c->flags, c->h_flags_dev, (uint32_t)F_COUNT,
                               (uint32_t)F_DONE, nq, c->boot_used ? c->flags + F_COUNT + PACE_WORDS : (uint32_t*)nullptr,
                               c->boot_used ? std::min<uint32_t>(nqt * 4u, BOOT_WORDS) : 0u, (const uint2*)c->qstat.as<uint2>(), nq,
                               (uint32_t)F_MAXERR, (uint32_t)(h->shadow ? F_MAXEPS : F_COUNT));
            HIPCHK(hipGetLastError());
        }
    }
    // Exact-scan-only batches (the reference's own f32 layout: BASELINE config 1, one query per call; forced exact; k beyond the
    // fast path): the scan goes onto the stream right here. Round 4 waited for the query conversion on the host first, then
    // enqueued the scan and waited again, then copied the results and waited a third time: 385 us for one query against 10k x
    // 384 rows, almost none of it device work. (A packed search keeps the old order: its records must be PROVISIONAL until
    // search_finish has seen the flags - cgv_search_packed_begin_f32_dev.)
    c->exact_enqueued = false;
    if (!mfma && !c->on_caller) {
        // flags to the pinned mirror (and cleared for the next search) by the scan's own kernel when it is the one-kernel form
        // (a few queries), else by a one-wave kernel behind it, as on the MFMA path: no copy-engine launch behind the scan, no
        // memset in front of the next one
        c->h_flags[F_DONE] = 0;  // (no kernel of this context is in flight: the host may write its mirror)
        bool publish = true;
        if ((rc = exact_search(h, c, nullptr, nq, k, out_idx, out_score, s, -1, false, &publish, nq))) return rc;   // (no query list: all of them, in order)
        c->exact_enqueued = true;
        c->published = true;
        if (publish) {
            hipLaunchKernelGGL(publish_flags_kernel, dim3(1), dim3(64), 0, s, c->flags, c->h_flags_dev, (uint32_t)F_COUNT,
                               (uint32_t)F_DONE, nq, (uint32_t*)nullptr, 0u);
            HIPCHK(hipGetLastError());
        }
    }
    if (h->profiling > 1) HIPCHK(hipEventRecord(c->ev[3], s));
    if (!c->published) HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, F_COUNT * 4, hipMemcpyDeviceToHost, s));
    return CGV_OK;
}

// Wait for the batch enqueued on `c`, run the exact path for the queries whose guarantee check
// failed (or for all of them on an f32 / forced-exact index), fold the statistics in.
// Called WITHOUT h->mu (the context is owned by the caller); takes it for the statistics.
// Wait for a stream: poll for up to cgv_index::spin_us microseconds (cgv_set_spin_us; default 3000; 0 = never) before blocking.
// A batch takes ~1.5 ms, and the wake-up of a blocked hipStreamSynchronize costs tens of microseconds of it.
// Device-visible alias of a pinned / registered HOST pointer, or NULL (pageable memory, device memory, unknown).
// The WHOLE range [p, p + bytes) must be pinned / registered and map to one contiguous device range: a buffer that is only
// partly registered, or that starts inside a pinned allocation and runs past its end, is staged like pageable memory
// instead of letting a kernel fault on its tail.
void* device_alias(const void* p, size_t bytes) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    if (bytes > 1) {
        hipPointerAttribute_t last;
        memset(&last, 0, sizeof(last));
        if (hipPointerGetAttributes(&last, (const char*)p + (bytes - 1)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (last.type != hipMemoryTypeHost || (const char*)last.devicePointer != (const char*)at.devicePointer + (bytes - 1))
            return nullptr;
    }
    return at.devicePointer;
}

// ev != NULL: wait for that event (recorded on s) instead of the whole stream.
int wait_stream(hipStream_t s, long spin_us, hipEvent_t ev) {
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(s);
            if (e == hipSuccess) return CGV_OK;
            if (e != hipErrorNotReady) {
                (void)hipGetLastError();
                break;
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us)
                break;
        }
    }
    if (ev) HIPCHK(hipEventSynchronize(ev));
    else HIPCHK(hipStreamSynchronize(s));
    return CGV_OK;
}

int search_finish(cgv_index* h, SearchCtx* c) {
    hipStream_t s = c->cur();
    const uint32_t nq = c->nq, k = c->k;
    int rc;
    if ((rc = wait_stream(s, h->spin_us, c->wait_packed ? c->packed_done : nullptr))) return rc;
    c->wait_packed = false;
    c->rewrote = false;
    if (c->published && c->h_flags[F_DONE] != nq)
        return fail(CGV_ERR_INTERNAL, "search pipeline finished without publishing its flags");
    if (c->h_flags[F_NONFINITE_Q] & 1u)
        return fail(CGV_ERR_NONFINITE, "query contains NaN/Inf (the reference panics at simd_ops.rs:379)");
    if (c->h_flags[F_NONFINITE_Q] & 2u)
        return fail(CGV_ERR_INVALID_ARG, "fp8 index: a query's largest magnitude is outside [2^-48, 2^48]");
    uint32_t nfb = 0, nrepaired = 0, nsfix = 0;
    float me = 0.0f;
    if (!c->mfma && c->exact_enqueued) {
        // the scan ran behind the query conversion on the same stream: nothing left to do
    } else if (!c->mfma) {
        c->rewrote = true;
        if ((rc = exact_search(h, c, nullptr, nq, k, c->out_idx, c->out_score, s))) return rc;   // (all queries, in order)
        if (h->profiling > 1) HIPCHK(hipEventRecord(c->ev[3], s));
        HIPCHK(hipStreamSynchronize(s));
    } else {
        memcpy(&me, &c->h_flags[F_MAXERR], 4);
        if (h->shadow) memcpy(&c->eps, &c->h_flags[F_MAXEPS], 4);  // largest per-query bound of this batch
        nfb = c->h_flags[F_FB_COUNT];
        if (c->published && !c->top2) nsfix = c->h_flags[F_COMPACT];   // queries put right inside the final kernel (emitting sample)
        uint32_t nscan = nfb;   // queries left for the exact scan
        if (nfb > 0 && c->top2 && c->repair) {
            // COARSE_TOP2: a query that failed on its floor alone (flag 2) gets the offending cells re-scanned (kernels_repair.h);
            // what is still flagged afterwards is compacted for the exact scan
            c->rewrote = true;
            Top2RepairArgs ra;
            ra.rows = h->rows;
            ra.qrows = c->qrows.as<char>();
            ra.nq = nq;
            ra.n = (uint32_t)h->n;
            ra.D = h->D;
            ra.ld = h->ld;
            ra.k = k;
            ra.metric = (uint32_t)h->metric;
            ra.cellb = c->cellb.as<float>();
            ra.nsplit = c->t2_nsplit;
            ra.cnt = c->t2_cnt;
            ra.R = c->t2_R;
            ra.P = c->t2_P;
            ra.theta = c->reptheta.as<float>();
            ra.keys = c->repkeys.as<uint64_t>();
            ra.nkeys = c->repn.as<uint32_t>();
            ra.idmap = h->idmap;
            ra.out_idx = c->out_idx;
            ra.out_score = c->out_score;
            ra.fb_flag = c->fbflag.as<uint32_t>();
            ra.repaired = c->flags + F_MAXEPS;   // (a free word here: the flag words were published and cleared; read back below)
            ra.newkeys = c->repnew.as<uint64_t>();
            ra.nnew = c->repcnt.as<uint32_t>();
            ra.bad = c->repcnt.as<uint32_t>() + TOP2_MAX_NQ_C;
            {
                const size_t rowb = (size_t)h->ld * h->esize, pitch = rowb + 16;
                ra.rows_pp = (uint32_t)std::max<size_t>(1, std::min<size_t>(REPAIR_ROWS, (REPAIR_LDS_BYTES - rowb) / pitch));
                ra.qoff = (uint32_t)(ra.rows_pp * pitch);
            }
            const size_t rlds = (size_t)ra.qoff + (size_t)h->ld * h->esize;
            HIPCHK(hipMemsetAsync(c->flags + F_MAXEPS, 0, 4, s));
            const dim3 rgrid(nq, REPAIR_GRID);
            switch (h->dtype) {
                case CGV_DTYPE_F32: hipLaunchKernelGGL(top2_repair_scan_kernel<DT_F32>, rgrid, dim3(256), rlds, s, ra); break;
                case CGV_DTYPE_BF16: hipLaunchKernelGGL(top2_repair_scan_kernel<DT_BF16>, rgrid, dim3(256), rlds, s, ra); break;
                case CGV_DTYPE_FP16: hipLaunchKernelGGL(top2_repair_scan_kernel<DT_FP16>, rgrid, dim3(256), rlds, s, ra); break;
                default: hipLaunchKernelGGL(top2_repair_scan_kernel<DT_FP8>, rgrid, dim3(256), rlds, s, ra); break;
            }
            hipLaunchKernelGGL(top2_repair_merge_kernel, dim3(nq), dim3(256), 0, s, ra);
            hipLaunchKernelGGL(compact_flags_kernel, dim3((nq + 255) / 256), dim3(256), 0, s,
                               c->fbflag.as<uint32_t>(), nq, c->qlist.as<uint32_t>(), c->flags + F_COMPACT);
            // the two counters reach the pinned mirror by plain stores of a one-wave kernel (two copy-engine launches cost ~20 us)
            hipLaunchKernelGGL(mirror_words_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)c->flags, c->h_flags_dev,
                               (uint32_t)F_COMPACT, (uint32_t)F_MAXEPS);
            HIPCHK(hipGetLastError());
            if (int wrc = wait_stream(s, h->spin_us)) return wrc;
            nscan = c->h_flags[F_COMPACT];
            nrepaired = c->h_flags[F_MAXEPS];
            if (nscan > 0) {
                if ((rc = exact_search(h, c, c->qlist.as<uint32_t>(), nscan, k, c->out_idx, c->out_score, s))) return rc;
                if (h->profiling > 1) HIPCHK(hipEventRecord(c->ev[3], s));
                HIPCHK(hipStreamSynchronize(s));
            }
        } else if (nfb > 0) {
            c->rewrote = true;
            hipLaunchKernelGGL(compact_flags_kernel, dim3((nq + 255) / 256), dim3(256), 0, s,
                               c->fbflag.as<uint32_t>(), nq, c->qlist.as<uint32_t>(), c->flags + F_COMPACT);
            if ((rc = exact_search(h, c, c->qlist.as<uint32_t>(), nfb, k, c->out_idx, c->out_score, s))) return rc;
            if (h->profiling > 1) HIPCHK(hipEventRecord(c->ev[3], s));
            HIPCHK(hipStreamSynchronize(s));
        }
        (void)nscan;
    }
    float coarse_ms = 0.0f, total_ms = 0.0f;
    if (h->profiling) {
        float ms = 0.0f;
        if (c->timed_coarse && hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) coarse_ms = ms;
        if (h->profiling > 1 && hipEventElapsedTime(&ms, c->ev[0], c->ev[3]) == hipSuccess) total_ms = ms;
    }
    float phase_us[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // level 3, MFMA path: conversion | first threshold | emitting launches | final + publish
    if (h->profiling > 2 && c->mfma && h->n && !c->rewrote) {
        hipEvent_t seq[5] = {c->ev[0], c->pev[0], c->pev[1], c->pev[2], c->ev[3]};
        for (int i = 0; i < 4; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, seq[i], seq[i + 1]) == hipSuccess) phase_us[i] = 1e3f * ms;
            else (void)hipGetLastError();
        }
    }
    // the last kernel reset the flag words - unless the exact scan ran afterwards (its kernels use them too)
    c->flags_clean = c->published && !c->rewrote;
    if (c->mfma && c->top2) c->floor_clean = true;   // final_kernel ran for every query and cleared the floor words it read
    std::lock_guard<std::mutex> lk(h->mu);
    h->st.max_observed_err = std::max(h->st.max_observed_err, me);
    h->st.fallback_queries += nfb;
    h->top2_stats[0] += (c->mfma && c->top2) ? 1u : 0u;
    h->top2_stats[1] += (c->mfma && c->top2) ? nfb : 0u;
    h->top2_stats[2] += nrepaired;
    h->sample_repairs += nsfix;
    h->st.last_path = (c->mfma && h->n) ? 1u : 0u;
    h->last_top2 = c->mfma && c->top2;
    h->st.last_kprime = c->kprime;
    h->st.last_eps = c->eps;
    h->st.last_coarse_ms = coarse_ms;
    h->st.last_total_ms = total_ms;
    for (int i = 0; i < 4; ++i) h->last_phase_us[i] = phase_us[i];
    h->last_coarse_rows = c->coarse_rows;
    return CGV_OK;
}

// ---- context pool ---------------------------------------------------------------
// A search holds one context from acquire to release; writers (add / update / reserve / load /
// set_id_map) wait until every context is free and keep h->mu while they work. cgv_set_stream does
// not wait: it only changes where later calls record their ordering event.
// split = taken by cgv_search_begin_f32_dev (released by a later cgv_search_end): if every context
// is held that way by the CALLING thread, waiting would wait for this thread itself -> nullptr
// (the caller reports CGV_ERR_BUSY) instead of a deadlock.
SearchCtx* acquire_ctx(cgv_index* h, std::unique_lock<std::mutex>& lk, bool split) {
    SearchCtx* got = nullptr;
    const std::thread::id me = std::this_thread::get_id();
    bool self_deadlock = false;
    h->cv.wait(lk, [&] {
        int mine = 0;
        for (SearchCtx& c : h->ctx) {
            if (!c.busy) {
                got = &c;
                return true;
            }
            if (c.split && c.owner == me) ++mine;
        }
        if (mine == N_CTX) {
            self_deadlock = true;
            return true;
        }
        return false;
    });
    if (self_deadlock) return nullptr;
    got->busy = true;
    got->split = split;
    got->owner = me;
    got->gen++;
    got->on_caller = false;  // (a packed ticket ended through cgv_search_end leaves these set)
    got->rec_out = nullptr;
    got->wait_packed = false;
    dev_inflight_add(h, +1);
    return got;
}

void release_ctx(cgv_index* h, SearchCtx* c) {
    {
        std::lock_guard<std::mutex> lk(h->mu);
        c->busy = false;
    }
    dev_inflight_add(h, -1);
    h->cv.notify_all();
}

// Writers wait until no search is in flight. A context held by the CALLING thread through cgv_search_begin_f32_dev
// (released only by its own later cgv_search_end) would make that wait a wait for itself: CGV_ERR_BUSY instead,
// mirroring acquire_ctx.
int wait_all_idle(cgv_index* h, std::unique_lock<std::mutex>& lk) {
    const std::thread::id me = std::this_thread::get_id();
    bool self = false;
    h->cv.wait(lk, [&] {
        bool idle = true;
        for (SearchCtx& c : h->ctx) {
            if (!c.busy) continue;
            idle = false;
            if (c.split && c.owner == me) self = true;
        }
        return idle || self;
    });
    if (self)
        return fail(CGV_ERR_BUSY, "this thread holds a search ticket of the handle (cgv_search_begin_f32_dev): call "
                                  "cgv_search_end before changing or reading the index");
    return CGV_OK;
}

// order the context's stream after everything the caller queued on the handle's stream
int order_after_caller(cgv_index* h, SearchCtx* c) {
    HIPCHK(hipEventRecord(c->dep, h->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->dep, 0));
    return CGV_OK;
}

// Batches in flight (cgv_search_begin_f32_dev / cgv_search_packed_begin_f32_dev) whose queries sit in pinned HOST memory (the
// caller passed the device alias): the conversion kernel reading them in place holds its waves on the CUs for the length of
// the PCIe transfer (3 MB: 63 us) - fine for a serial call, where nothing else wants the device, but with batches in flight that
// conversion runs beside the other batches' kernels or, worse, while nothing computes (profiles/r05_batches_in_flight_traces.txt).
// Here the copy engine fetches the batch on the handle's copy stream, issued at once (it runs under whatever the device is
// computing), and the batch's stream waits for it: the conversion then reads HBM (5 us). Returns the pointer to convert from
// (the staging copy, or q itself: device memory, small batches, no other batch in flight, A/B knob off).
constexpr size_t FETCH_MIN_BYTES = 256u << 10;
const float* fetch_host_queries(cgv_index* h, SearchCtx* c, const float* q, uint32_t nq, hipStream_t s, int* rc) {
    *rc = CGV_OK;
    const size_t bytes = (size_t)nq * h->D * 4;
    if (bytes < FETCH_MIN_BYTES || !tun().fetch_queries) return q;
    // only while another batch of the handle is in flight (caller holds mu): a lone batch has nothing to run under, and the copy's
    // launch + the event hop cost it 4-10 us (serial begin / end pairs on the 125 k-row shard: 0.3135 -> 0.320 ms with the fetch)
    int busy = 0;
    for (const SearchCtx& o : h->ctx) busy += o.busy ? 1 : 0;
    if (busy <= 1) return q;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, q) != hipSuccess) {
        (void)hipGetLastError();
        return q;
    }
    if (at.type != hipMemoryTypeHost || !at.hostPointer) return q;
    if ((*rc = c->qstage.ensure(bytes))) return q;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && *rc == CGV_OK) *rc = fail(CGV_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    // the copy is ordered like the conversion kernel it replaces: after what the caller queued on the handle's stream (c->dep,
    // recorded by order_after_caller / the packed begin) and after earlier work on the stream the batch runs on - a query batch
    // produced asynchronously (a non-blocking D2H into the pinned buffer, say) must not be fetched stale
    chk(hipStreamWaitEvent(h->copy_stream, c->dep, 0), "hipStreamWaitEvent(copy, dep)");
    if (c->on_caller) {
        chk(hipEventRecord(c->dep_run, s), "hipEventRecord(dep_run)");
        chk(hipStreamWaitEvent(h->copy_stream, c->dep_run, 0), "hipStreamWaitEvent(copy, dep_run)");
    }
    chk(hipMemcpyAsync(c->qstage.p, at.hostPointer, bytes, hipMemcpyHostToDevice, h->copy_stream), "hipMemcpyAsync(queries)");
    chk(hipEventRecord(c->copied, h->copy_stream), "hipEventRecord");
    chk(hipStreamWaitEvent(s, c->copied, 0), "hipStreamWaitEvent");
    return *rc ? q : c->qstage.as<float>();
}

}  // namespace cgvi
