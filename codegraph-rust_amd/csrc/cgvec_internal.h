// cgvec_internal.h - what the two halves of the kNN library share: the handle and search-context structures, the error
// convention and the declarations of the pipeline functions (search.hip) the C ABI (abi.hip) is built on. Round 6: cgvec.hip
// (2946 lines) was split into plan.cpp (host-only planner), search.hip (pipeline + kernel launches) and abi.hip (entry points).
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cgvec.h"
#include "coalesce.h"
#include "common.h"
#include "coarse_launch.h"
#include "kernels_coarse.h"
#include "kernels_exact.h"
#include "kernels_exact_small.h"
#include "kernels_prep.h"
#include "plan.h"
#include "kernels_repair.h"
#include "kernels_select.h"

using namespace cgv;

extern "C" int cgv_set_error_(int code, const char* msg);   // abi.hip: the library's thread-local error message

namespace {

inline int fail(int code, const std::string& msg) { return cgv_set_error_(code, msg.c_str()); }


#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(_e == hipErrorOutOfMemory ? CGV_ERR_OOM : CGV_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return CGV_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e != hipSuccess) return fail(CGV_ERR_OOM, "hipMalloc scratch: " + std::string(hipGetErrorString(e)));
        bytes = want;
        return CGV_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const {
        return (T*)p;
    }
};

enum Flag { F_NONFINITE_C = 0, F_NONFINITE_Q, F_FB_COUNT, F_MAXERR, F_NAN, F_COMPACT, F_MAXEPS, F_DONE, F_COUNT = 8 };
// F_DONE: workgroups of the last kernel that have finished (rescore_body publishes the flags to the pinned mirror)
// a search context's flag words are followed by the coarse kernels' pacing words (kernels_coarse.h: Pace), one
// per workgroup, cleared together with the flags at the start of every search
constexpr uint32_t PACE_WORDS = 1024;
// ... and by the rendezvous words of the fused sample + emit launch (kernels_coarse.h: BootSync), 4 per query tile; zero at
// the start of every search: cleared with the flags, and again by publish_flags_kernel behind a search that used them
constexpr uint32_t BOOT_WORDS = 4 * 64;
constexpr uint32_t CTX_FLAG_WORDS = F_COUNT + PACE_WORDS + BOOT_WORDS;


}  // namespace


// Per-search scratch + stream. A handle owns N_CTX of them so that consecutive batches (or
// concurrent callers, SURVEY.md §8(b) "internal stream pool") overlap on the device: the short
// latency-bound kernels of one batch (boot, select, re-score) run beside the coarse kernel of
// the next one. Everything a search writes lives here; the corpus is shared and read-only.
struct SearchCtx {
    hipStream_t stream = nullptr;  // owned, non-blocking
    hipEvent_t dep = nullptr;      // ordering after the caller's stream (ingest, query producer)
    hipEvent_t dep_run = nullptr;  // packed searches: ordering of the copy-engine fetch after earlier work on the consumer's stream
    uint32_t* rec_out = nullptr;   // cgv_search_packed_begin_f32_dev: the caller's record buffer of the batch in flight (NULL: not a packed search)
    // cgv_search_packed_begin_f32_dev runs the WHOLE batch on the consumer's stream (no hop onto `stream` and back: two
    // cross-stream event waits, ~7 us each on this part, per batch of the N > 1 step): `run` is that stream while `on_caller`
    bool on_caller = false;
    hipStream_t run = nullptr;
    hipStream_t cur() const { return on_caller ? run : stream; }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pev[3] = {nullptr, nullptr, nullptr};   // profiling level 3: behind the query conversion, the first threshold, the last emitting launch
    // cgv_search_packed_begin_f32_dev: recorded on the consumer's stream right behind the pack kernel. The consumer's stream is
    // SHARED with whatever the caller enqueues next (the collective, the merge, the next batch of the same stream), so the
    // search's end waits for THIS, not for the stream (end(A) used to be serialised behind batch B's device work)
    hipEvent_t packed_done = nullptr;
    bool wait_packed = false;
    uint32_t* flags = nullptr;    // device, F_COUNT words
    uint32_t* h_flags = nullptr;  // pinned host mirror
    uint32_t* h_flags_dev = nullptr;  // ... as the device sees it (the last kernel of a search publishes the flags there)
    // Small pageable batches (the trait-level call: ONE query in a Rust Vec<f32>, results into a Vec): the query goes through
    // this pinned, device-mapped staging area with a host memcpy and the conversion kernel reads it in place; the last kernel
    // writes the results into its second half and the host copies them out after the batch's one synchronisation - no
    // copy-engine operation (H2D + 2 x D2H, ~10-20 us each for a few KB) on the path of a call that takes ~100 us in all
    char* h_stage = nullptr;      // pinned host: [0, SMALL_Q_BYTES) queries | [SMALL_Q_BYTES, + SMALL_OUT_BYTES) ids, then scores
    char* h_stage_dev = nullptr;  // ... as the device sees it
    bool flags_clean = false;     // device flag + pacing words are known to be zero (the last kernel resets them)
    bool published = false;       // the search in flight publishes its flags itself (no D2H copy enqueued)
    DevBuf qstage, qrows, qnorm, qinvn, qrexp, tau, nbest, best, overflow, fbflag, qlist, cand, candcnt, scores,
        keysA, keysB, outidx, outscore, dump, qshadow, qres, trace, floor, lad, ladc, qstat, xdone, qspread, cellb, reptheta, repkeys, repn, scand, repnew, repcnt;
    double host_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // diagnostics: host timeline of the last cgv_search_f32 (CGV_TRACE=1)
    bool busy = false, split = false;
    hipEvent_t copied = nullptr;   // batches in flight: the copy engine has fetched this batch's host queries (fetch_host_queries)
    std::thread::id owner;
    // state of the search in flight (between begin and end)
    uint32_t gen = 0, nq = 0, k = 0;
    uint64_t* out_idx = nullptr;
    float* out_score = nullptr;
    bool mfma = false, timed_coarse = false;
    bool boot_used = false;  // the search in flight used the fused sample + emit launch (its rendezvous words need clearing)
    bool top2 = false;       // the search in flight took the small-batch form (COARSE_TOP2: one launch, no thresholds)
    bool repair = false;     // ... and can repair a floor violation by re-scanning the offending cells (kernels_repair.h)
    uint32_t t2_nsplit = 0, t2_cnt = 0, t2_R = 0, t2_P = 0;   // its walk (what top2_repair_kernel maps a cell to rows with)
    bool floor_clean = false;  // the TOP2 floor words are known to be zero (final_kernel clears the ones it read)
    bool rewrote = false;  // search_finish ran the exact scan and rewrote (some of) the outputs after its first sync
    bool exact_enqueued = false;  // exact-scan-only batch (f32 index, forced exact, large k): the scan was enqueued by
                                  // search_enqueue itself - ONE host synchronisation per call instead of three
    uint64_t coarse_rows = 0;
    float eps = 0.0f;
    uint32_t kprime = 0;
    size_t bytes() const {
        const DevBuf* bufs[] = {&qstage, &qrows, &qnorm, &qinvn, &qrexp, &tau, &nbest, &best, &overflow, &fbflag,
                                &qlist, &cand, &candcnt, &scores, &keysA, &keysB, &outidx, &outscore, &dump,
                                &qshadow, &qres, &trace, &floor, &lad, &ladc, &qstat, &xdone, &qspread, &cellb, &reptheta, &repkeys, &repn, &scand, &repnew, &repcnt};
        size_t b = 0;
        for (const DevBuf* d : bufs) b += d->bytes;
        return b;
    }
    void release_all() {
        DevBuf* bufs[] = {&qstage, &qrows, &qnorm, &qinvn, &qrexp, &tau, &nbest, &best, &overflow, &fbflag,
                          &qlist, &cand, &candcnt, &scores, &keysA, &keysB, &outidx, &outscore, &dump,
                          &qshadow, &qres, &trace, &floor, &lad, &ladc, &qstat, &xdone, &qspread, &cellb, &reptheta, &repkeys, &repn, &scand, &repnew, &repcnt};
        for (DevBuf* d : bufs) d->release();
    }
};

constexpr uint32_t TOP2_MAX_NQ_C = 64;                 // queries of a small batch (COARSE_TOP2)
constexpr size_t REPAIR_LDS_BYTES = 60u << 10;         // staged rows + the query row of top2_repair_scan_kernel (within the 64 KiB default)
constexpr int N_CTX = 3;
constexpr size_t SMALL_Q_BYTES = 256u << 10, SMALL_OUT_BYTES = 128u << 10;   // SearchCtx::h_stage

struct cgv_index {
    int device = 0;
    uint32_t D = 0, ld = 0;
    int metric = 0, dtype = 0;
    uint32_t esize = 2;
    uint64_t n = 0, cap = 0;
    IdMap idmap = {0, 0, 1, 0, 0};  // local row -> reported id (cgv_set_index_base / cgv_set_id_map)
    char* rows = nullptr;
    float* norm = nullptr;
    float* invn = nullptr;
    float* blk_min = nullptr;
    float* blk_max = nullptr;
    int8_t* rexp = nullptr;       // fp8 only: per-row scale exponent
    // CGV_DTYPE_F32_SHADOW: dtype == F32 (rows, exact paths) + a bf16 blocked copy for the coarse pass
    bool shadow = false;
    char* srows = nullptr;
    uint32_t lds = 0;             // leading dimension (elements) of the shadow
    uint32_t* resmax_dev = nullptr;  // [2] max rounding residual over the corpus: relative, absolute (float bits)
    float res_rel_c = 0.0f, res_abs_c = 0.0f;
    uint32_t* flags = nullptr;    // device, F_COUNT words (ingest side)
    float* max_norm_dev = nullptr;
    uint32_t* h_flags = nullptr;  // pinned host mirror (F_COUNT words + 1 float)
    hipStream_t own_stream = nullptr, stream = nullptr;  // ingest / caller-ordering stream
    hipStream_t copy_stream = nullptr;   // H2D of the queries of batches in flight (fetch_host_queries)
    int n_cu = 256;
    float max_norm_c = 0.0f;
    DevBuf addstage;
    SearchCtx ctx[N_CTX];
    std::mutex mu;
    std::condition_variable cv;
    int profiling = 0;  // 0 off; 1 = HIP events around the dominant coarse launch; 2 = also around the whole pipeline
    long spin_us = 3000;  // cgv_set_spin_us: how long a search's end polls its stream before it blocks
    bool force_exact = false;
    bool wide_range = false;  // a stored row's magnitude is outside [2^-40, 2^40]: searches take the exact scan (kernels_prep.h)
    bool last_top2 = false;   // the last finished search took the small-batch form (cgv_debug_last_top2_)
    uint64_t sample_repairs = 0;          // queries whose check failed on a cell of the emitting sample and were put right inside the final kernel
    uint64_t top2_stats[3] = {0, 0, 0};   // COARSE_TOP2 searches, their queries that failed the check, of those repaired by a cell re-scan
    cgv_stats st;
    uint64_t last_coarse_rows = 0;
    float last_phase_us[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // cgv_get_phase_times
    Coalescer co;   // group commit of concurrent small cgv_search_f32 calls (coalesce.h)
    cgv_index() {
        memset(&st, 0, sizeof(st));
        co.max_q_bytes = SMALL_Q_BYTES;
        co.max_out_bytes = SMALL_OUT_BYTES;
    }
};


// ---- the pipeline and storage functions of search.hip that the entry points of abi.hip are built on ------------------------------
namespace cgvi {

struct IngestSnapshot {
    uint64_t n;
    float max_norm, res_rel, res_abs;
};

inline IngestSnapshot snapshot_of(const cgv_index* h) { return {h->n, h->max_norm_c, h->res_rel_c, h->res_abs_c}; }

size_t storage_bytes(const cgv_index* h, uint64_t nrows);
size_t shadow_bytes(const cgv_index* h, uint64_t nrows);
size_t device_bytes(const cgv_index* h);
int prep_dispatch(int dtype, const float* in, uint64_t n, uint32_t D, uint32_t ld, uint64_t row0, char* out,
                  float* norm, float* invn, int8_t* rexp, uint32_t* nonfinite, hipStream_t s,
                  uint32_t* z0 = nullptr, uint32_t* z1 = nullptr);
int grow(cgv_index* h, uint64_t need);
int ingest_enqueue(cgv_index* h, const float* rows_dev, uint64_t cnt, uint64_t row0, uint64_t n_valid = 0);
int ingest_rollback(cgv_index* h, const IngestSnapshot& snap);
int ingest_finish(cgv_index* h, uint64_t n_new);
int add_dev_locked(cgv_index* h, const float* rows_dev, uint64_t cnt);
int ensure_kernel_attrs(int device);
int launch_coarse(int dtype, int mode, const CoarseArgs& a, uint32_t W, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
SelectArgs make_select_args(SearchCtx* c, uint32_t nq, uint32_t nqt, uint32_t nsplit, uint32_t kprime,
                            const float* dense, uint32_t n_dense, uint64_t expected, size_t* lds_out, uint32_t extra_keys = 0);
int launch_select(SearchCtx* c, uint32_t nq, uint32_t nqt, uint32_t nsplit, uint32_t kprime,
                  const float* dense, uint32_t n_dense, hipStream_t s, uint64_t expected = 0,
                  bool tau_only = false, uint32_t extra_keys = 0);
int ensure_xdone(SearchCtx* c, hipStream_t s);
int exact_search(cgv_index* h, SearchCtx* c, const uint32_t* qlist_dev, uint32_t nql, uint32_t k, uint64_t* out_idx,
                 float* out_score, hipStream_t s, int op = -1, bool local_ids = false, bool* publish = nullptr,
                 uint32_t marker = 0u);
int search_enqueue(cgv_index* h, SearchCtx* c, const float* qdev, uint32_t nq, uint32_t k, uint64_t* out_idx,
                   float* out_score);
void* device_alias(const void* p, size_t bytes);
int wait_stream(hipStream_t s, long spin_us, hipEvent_t ev = nullptr);
int search_finish(cgv_index* h, SearchCtx* c);
SearchCtx* acquire_ctx(cgv_index* h, std::unique_lock<std::mutex>& lk, bool split = false);
void release_ctx(cgv_index* h, SearchCtx* c);
int wait_all_idle(cgv_index* h, std::unique_lock<std::mutex>& lk);
int order_after_caller(cgv_index* h, SearchCtx* c);
const float* fetch_host_queries(cgv_index* h, SearchCtx* c, const float* q, uint32_t nq, hipStream_t s, int* rc);

template <class F>
int atomic_ingest(cgv_index* h, F body) {
    const IngestSnapshot snap = snapshot_of(h);
    const int rc = body();
    if (rc != CGV_OK) {
        const std::string msg = cgv_last_error();  // keep the first error's message
        (void)ingest_rollback(h, snap);
        return fail(rc, msg);
    }
    return CGV_OK;
}

template <int DT>
void launch_exact_scores(cgv_index* h, SearchCtx* c, const uint32_t* qlist, uint32_t nql, float* scores, int op, hipStream_t s) {
    uint64_t gx = ((uint64_t)h->n + 31) / 32;
    if (gx > 16384) gx = 16384;
    hipLaunchKernelGGL(exact_scores_kernel<DT>, dim3((unsigned)gx, nql), dim3(256), 0, s, h->rows,
                       c->qrows.as<char>(), qlist, nql, (uint32_t)h->n, h->D, h->ld, op, scores);
}

#ifdef CGV_ABLATE_BUILD
void dev_inflight_add(const cgv_index* h, int d);
int dev_inflight(const cgv_index* h);
#else
inline void dev_inflight_add(const cgv_index*, int) {}
#endif

}  // namespace cgvi
using namespace cgvi;
