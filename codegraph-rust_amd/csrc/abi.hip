// abi.hip - the C ABI of include/cgvec.h: every extern "C" entry point of the kNN library (index life cycle, ingest, corpus files,
// searches - blocking, begin / end, packed, coalesced -, building blocks, statistics). The pipeline underneath is search.hip.
// There is NO CPU fallback: without a HIP device every entry point that needs one fails with CGV_ERR_HIP.
#include "cgvec_internal.h"

static thread_local std::string g_err;   // the library's thread-local error message (cgv_last_error)

static __global__ void f64_to_f32_kernel(const double* __restrict__ in, uint64_t total, float* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];  // `as f32`: round to nearest even
}

extern "C" {

uint32_t cgv_version(void) { return (0u << 16) | 9u; }  // 0.9: + sample-repair stats (0.8: scalar-arm normalise, small-batch stats, phase times, synthetic inputs)

// internal: lets the host mirror (host/store.cpp) share this library's thread-local error message
int cgv_set_error_(int code, const char* msg) {
    g_err = msg ? msg : "";
    return code;
}

const char* cgv_last_error(void) { return g_err.c_str(); }

#ifdef CGV_ABLATE_BUILD   // measurement flavour only (make ABLATE=1)
// internal (scripts): diagnostics of the last search on context `ctx` (CGV_TRACE=1): host timeline of cgv_search_f32
// in microseconds since entry {order, H2D enqueued, pipeline enqueued, D2H enqueued, stream done} and the final
// kernel's per-query phase stamps (100 MHz ticks: start, keys gathered, top-k' extracted, rows staged, scored, sorted, end)
int cgv_debug_trace_(cgv_index* h, uint32_t ctx, double* host_us8, uint64_t* stamps, uint32_t nq) {
    if (!h || ctx >= (uint32_t)N_CTX) return fail(CGV_ERR_INVALID_ARG, "bad argument");
    SearchCtx* c = &h->ctx[ctx];
    if (host_us8) memcpy(host_us8, c->host_us, sizeof(c->host_us));
    if (stamps && nq) {
        if (c->trace.bytes < (size_t)nq * 64) return fail(CGV_ERR_INVALID_ARG, "no trace recorded (CGV_TRACE unset?)");
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpy(stamps, c->trace.p, (size_t)nq * 64, hipMemcpyDeviceToHost));
    }
    return CGV_OK;
}

// internal (scripts/ab.py): set a planner knob at run time. Returns 0, or -1 for an unknown key.
int cgv_debug_set_(const char* key, double v) {
    if (!key) return -1;
    Tunables& t = tun();
    if (!strcmp(key, "plan_legacy")) t.plan_legacy = (int)v;
    else if (!strcmp(key, "sample_tiles")) t.sample_tiles = (int)v;
    else if (!strcmp(key, "plan_launches")) t.plan_launches = (int)v;
    else if (!strcmp(key, "hit_us")) t.hit_us = v;
    else if (!strcmp(key, "launch_us")) t.launch_us = v;
    else if (!strcmp(key, "zero_copy")) t.zero_copy = (int)v;
    else if (!strcmp(key, "pace")) t.pace = (int)v;
    else if (!strcmp(key, "epi")) t.epi = (int)v;
    else if (!strcmp(key, "fuse_sample")) t.fuse_sample = (int)v;
    else if (!strcmp(key, "top2")) t.top2 = (int)v;
    else if (!strcmp(key, "sample_emit")) t.sample_emit = (int)v;
    else if (!strcmp(key, "top2_repair")) t.top2_repair = (int)v;
    else if (!strcmp(key, "sample_repair")) t.sample_repair = (int)v;
    else if (!strcmp(key, "launch_events")) t.launch_events = (int)v;
    else if (!strcmp(key, "exact_small")) t.exact_small = (int)v;
    else if (!strcmp(key, "self_publish")) t.self_publish = (int)v;
    else if (!strcmp(key, "fetch_queries")) t.fetch_queries = (int)v;
    else if (!strcmp(key, "ladder")) t.ladder = (int)v;
    else return -1;
    return 0;
}
#endif  // CGV_ABLATE_BUILD

// internal (tests, scripts): the per-query fallback words of context `ctx` as the last search left them (0 = proven or repaired;
// 1 = exact scan, upper bits: why the cell re-scan gave up - kernels_repair.h)
int cgv_debug_fbflags_(cgv_index* h, uint32_t ctx, uint32_t* out, uint32_t nq) {
    if (!h || !out || ctx >= (uint32_t)N_CTX) return fail(CGV_ERR_INVALID_ARG, "bad argument");
    SearchCtx* c = &h->ctx[ctx];
    if (c->fbflag.bytes < (size_t)nq * 4) return fail(CGV_ERR_INVALID_ARG, "no such search");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(out, c->fbflag.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    return CGV_OK;
}

// internal (tests, scripts): 1 when the last finished search on the handle took the small-batch form (COARSE_TOP2)
int cgv_debug_last_top2_(cgv_index* h) { return (h && h->last_top2) ? 1 : 0; }

// internal (tests, scripts): the launch plan of a search over n rows with nq queries and k results on a device with
// n_cu compute units. out[0] = tiles of the sample launch (0: dense boot stage), out[1] = number of emitting
// launches m, out[2 .. 2+m) = tiles per launch. Returns the number of words written (0 if cap is too small).
uint32_t cgv_debug_plan_(uint64_t n, uint32_t k, uint32_t nq, uint32_t n_cu, int shadow, uint32_t* out, uint32_t cap) {
    const uint32_t kprime = shadow ? std::min<uint32_t>(((4 * k + 16 + 7) / 8) * 8, 256u) : kprime_of(k);
    const uint32_t nqt = (nq + BN - 1) / BN;
    const uint32_t nsplit_max = std::max<uint32_t>(1u, n_cu / std::max<uint32_t>(nqt, 1u));
    StagePlan p = plan_stages(n, kprime, nqt, n_cu, nsplit_max, 0, nq > 64);
    if (p.sample_emits && (p.counts.empty() || std::min<uint32_t>(p.counts[0], nsplit_max) < std::min<uint32_t>(p.sample_tiles, nsplit_max)))
        p = plan_stages(n, kprime, nqt, n_cu, nsplit_max, 0, false);
    if (!out || cap < 2 + p.counts.size()) return 0;
    out[0] = p.sample_tiles;
    out[1] = (uint32_t)p.counts.size() | (p.sample_emits ? 0x10000u : 0u);   // bit 16: the sample emits, `counts` start behind it
    for (size_t i = 0; i < p.counts.size(); ++i) out[2 + i] = p.counts[i];
    return (uint32_t)(2 + p.counts.size());
}

int cgv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int cgv_create(uint32_t dim, int metric, int dtype, int device_id, cgv_index** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dim == 0 || dim > 8192) return fail(CGV_ERR_INVALID_ARG, "dim must be in 1..=8192");
    if (metric != CGV_METRIC_COSINE && metric != CGV_METRIC_DOT && metric != CGV_METRIC_COSINE_SEQ &&
        metric != CGV_METRIC_COSINE_SCALAR)
        return fail(CGV_ERR_INVALID_ARG, "bad metric");
    const bool shadow = dtype == CGV_DTYPE_F32_SHADOW;
    if (shadow) dtype = CGV_DTYPE_F32;  // rows, exact paths and get_row are the f32 index; + a bf16 copy for the coarse pass
    if (dtype != CGV_DTYPE_F32 && dtype != CGV_DTYPE_BF16 && dtype != CGV_DTYPE_FP16 && dtype != CGV_DTYPE_FP8E4M3)
        return fail(CGV_ERR_INVALID_ARG, "unknown dtype (f32, bf16, fp16, fp8e4m3, f32+shadow)");
    if (dtype == CGV_DTYPE_FP8E4M3 && metric == CGV_METRIC_DOT)
        return fail(CGV_ERR_INVALID_ARG, "fp8 storage keeps a per-row scale: cosine only in this build");
    int ndev = cgv_device_count();
    if (ndev == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(CGV_ERR_INVALID_ARG, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    if (int arc = ensure_kernel_attrs(device_id)) return arc;
    cgv_index* h = new cgv_index();
    h->device = device_id;
    h->D = dim;
    const uint32_t kch = kchunk_of(dtype);  // elements per 64-byte row chunk
    h->ld = (dim + kch - 1) / kch * kch;
    h->metric = metric;
    h->dtype = dtype;
    h->esize = esize_of(dtype);
    h->shadow = shadow;
    h->lds = (dim + 31) / 32 * 32;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        h->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->flags, F_COUNT * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->max_norm_dev, 4);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_flags, (F_COUNT + 3) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->resmax_dev, 8);
    if (e == hipSuccess) e = hipMemset(h->resmax_dev, 0, 8);
    for (SearchCtx& c : h->ctx) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.dep, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.dep_run, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.packed_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.copied, hipEventDisableTiming);
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c.ev[i]);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&c.pev[i]);
        if (e == hipSuccess) e = hipMalloc((void**)&c.flags, CTX_FLAG_WORDS * 4);
        if (e == hipSuccess) e = hipHostMalloc((void**)&c.h_flags, F_COUNT * 4, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c.h_flags_dev, c.h_flags, 0);
        if (e == hipSuccess) e = hipHostMalloc((void**)&c.h_stage, SMALL_Q_BYTES + SMALL_OUT_BYTES, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c.h_stage_dev, c.h_stage, 0);
        if (e == hipSuccess) e = hipMemset(c.flags, 0, CTX_FLAG_WORDS * 4);
        if (e == hipSuccess) c.flags_clean = true;
    }
    if (e == hipSuccess) e = hipMemset(h->flags, 0, F_COUNT * 4);
    if (e == hipSuccess) e = hipMemset(h->max_norm_dev, 0, 4);
    if (e != hipSuccess) {
        std::string m = hipGetErrorString(e);
        cgv_destroy(h);
        return fail(CGV_ERR_HIP, "cgv_create: " + m);
    }
    h->stream = h->own_stream;
    *out = h;
    return CGV_OK;
}

int cgv_destroy(cgv_index* h) {
    if (!h) return CGV_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->rows) {
        (void)hipFree(h->rows);
        (void)hipFree(h->norm);
        (void)hipFree(h->invn);
        (void)hipFree(h->blk_min);
        (void)hipFree(h->blk_max);
        (void)hipFree(h->rexp);
        if (h->srows) (void)hipFree(h->srows);
    }
    if (h->resmax_dev) (void)hipFree(h->resmax_dev);
    h->addstage.release();
    for (SearchCtx& c : h->ctx) {
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        c.release_all();
        if (c.flags) (void)hipFree(c.flags);
        if (c.h_flags) (void)hipHostFree(c.h_flags);
        if (c.h_stage) (void)hipHostFree(c.h_stage);
        if (c.dep) (void)hipEventDestroy(c.dep);
        if (c.dep_run) (void)hipEventDestroy(c.dep_run);
        if (c.packed_done) (void)hipEventDestroy(c.packed_done);
        if (c.copied) (void)hipEventDestroy(c.copied);
        for (int i = 0; i < 4; ++i)
            if (c.ev[i]) (void)hipEventDestroy(c.ev[i]);
        for (int i = 0; i < 3; ++i)
            if (c.pev[i]) (void)hipEventDestroy(c.pev[i]);
        if (c.stream) (void)hipStreamDestroy(c.stream);
    }
    if (h->flags) (void)hipFree(h->flags);
    if (h->max_norm_dev) (void)hipFree(h->max_norm_dev);
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    delete h;
    return CGV_OK;
}

int cgv_reserve(cgv_index* h, uint64_t n_rows) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    HIPCHK(hipSetDevice(h->device));
    return grow(h, n_rows);
}

int cgv_add_f32_dev(cgv_index* h, const float* rows_dev, uint64_t n) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n && !rows_dev) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    HIPCHK(hipSetDevice(h->device));
    return atomic_ingest(h, [&] { return add_dev_locked(h, rows_dev, n); });
}

int cgv_add_f32(cgv_index* h, const float* rows_host, uint64_t n) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n && !rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    HIPCHK(hipSetDevice(h->device));
    int rc = grow(h, h->n + n);
    if (rc) return rc;
    return atomic_ingest(h, [&]() -> int {
        const uint64_t chunk_rows = std::max<uint64_t>(1, (256ull << 20) / ((uint64_t)h->D * 4));
        for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
            const uint64_t c = std::min<uint64_t>(chunk_rows, n - r0);
            int r;
            if ((r = h->addstage.ensure((size_t)c * h->D * 4))) return r;
            HIPCHK(hipMemcpyAsync(h->addstage.p, rows_host + r0 * h->D, (size_t)c * h->D * 4, hipMemcpyHostToDevice,
                                  h->stream));
            if ((r = add_dev_locked(h, h->addstage.as<float>(), c))) return r;
        }
        return CGV_OK;
    });
}

int cgv_add_f64(cgv_index* h, const double* rows_host, uint64_t n) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n && !rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    HIPCHK(hipSetDevice(h->device));
    int rc = grow(h, h->n + n);
    if (rc) return rc;
    return atomic_ingest(h, [&]() -> int {
        const uint64_t chunk_rows = std::max<uint64_t>(1, (128ull << 20) / ((uint64_t)h->D * 8));
        for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
            const uint64_t c = std::min<uint64_t>(chunk_rows, n - r0);
            const uint64_t total = c * h->D;
            int r;
            // staging: [c*D doubles][c*D floats]
            if ((r = h->addstage.ensure((size_t)total * 12))) return r;
            double* d64 = h->addstage.as<double>();
            float* d32 = (float*)(h->addstage.as<char>() + (size_t)total * 8);
            HIPCHK(hipMemcpyAsync(d64, rows_host + r0 * h->D, (size_t)total * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)std::min<uint64_t>(4096, (total + 255) / 256)), dim3(256),
                               0, h->stream, (const double*)d64, total, d32);
            HIPCHK(hipGetLastError());
            if ((r = add_dev_locked(h, d32, c))) return r;
        }
        return CGV_OK;
    });
}

// ---- corpus files in the reference's mmap format (memory.rs:242-374) -------------------------
// 16-byte header {u64 vector_count, u64 dimension} (native endian) + row-major f32.
namespace {
struct MappedFile {
    int fd = -1;
    void* p = MAP_FAILED;
    size_t len = 0;
    ~MappedFile() {
        if (p != MAP_FAILED) munmap(p, len);
        if (fd >= 0) close(fd);
    }
};
}  // namespace

int cgv_load_mmap(cgv_index* h, const char* path, uint64_t* out_rows) {
    if (out_rows) *out_rows = 0;
    if (!h || !path) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    MappedFile mf;
    mf.fd = open(path, O_RDONLY);
    if (mf.fd < 0) return fail(CGV_ERR_IO, std::string("Failed to open mmap file: ") + strerror(errno));
    struct stat st;
    if (fstat(mf.fd, &st) != 0) return fail(CGV_ERR_IO, std::string("Failed to map file: ") + strerror(errno));
    mf.len = (size_t)st.st_size;
    if (mf.len < 16) return fail(CGV_ERR_IO, "Invalid mmap file: too small");  // memory.rs:318-322
    mf.p = mmap(nullptr, mf.len, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (mf.p == MAP_FAILED) return fail(CGV_ERR_IO, std::string("Failed to map file: ") + strerror(errno));
    (void)madvise(mf.p, mf.len, MADV_SEQUENTIAL);
    uint64_t hdr[2];
    memcpy(hdr, mf.p, 16);
    const uint64_t count = hdr[0], dim = hdr[1];
    if (dim != h->D)  // memory.rs:329-334
        return fail(CGV_ERR_DIM_MISMATCH,
                    "Dimension mismatch: expected " + std::to_string(h->D) + ", found " + std::to_string(dim));
    const unsigned __int128 want = (unsigned __int128)16 + (unsigned __int128)count * dim * 4;
    if (want != (unsigned __int128)mf.len)  // memory.rs:337-348
        return fail(CGV_ERR_IO, "Invalid mmap file size: expected " + std::to_string((uint64_t)want) + ", got " +
                                    std::to_string(mf.len));
    if (count == 0) return CGV_OK;
    const float* src = (const float*)((const char*)mf.p + 16);

    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    HIPCHK(hipSetDevice(h->device));
    int rc = grow(h, h->n + count);
    if (rc) return rc;
    // page cache -> pinned staging -> device, double buffered: the CPU copy of chunk i+1 overlaps the
    // H2D copy + conversion kernels of chunk i (all on the handle's stream, no host sync per chunk).
    const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / (dim * 4));
    const size_t chunk_bytes = (size_t)chunk_rows * dim * 4;
    float* pin[2] = {nullptr, nullptr};
    float* dev[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (dev[i]) (void)hipFree(dev[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    };
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc((void**)&pin[i], chunk_bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&dev[i], chunk_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        cleanup();
        return fail(CGV_ERR_OOM, std::string("cgv_load_mmap staging: ") + hipGetErrorString(e));
    }
    const uint64_t base = h->n;
    const IngestSnapshot snap = snapshot_of(h);
    uint64_t ci = 0;
    for (uint64_t r0 = 0; r0 < count && rc == CGV_OK; r0 += chunk_rows, ++ci) {
        const int b = (int)(ci & 1);
        const uint64_t c = std::min<uint64_t>(chunk_rows, count - r0);
        if (ci >= 2 && hipEventSynchronize(done[b]) != hipSuccess) rc = fail(CGV_ERR_HIP, "cgv_load_mmap: event wait");
        if (rc) break;
        memcpy(pin[b], src + r0 * dim, (size_t)c * dim * 4);
        if (hipMemcpyAsync(dev[b], pin[b], (size_t)c * dim * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            rc = fail(CGV_ERR_HIP, "cgv_load_mmap: H2D copy");
            break;
        }
        rc = ingest_enqueue(h, dev[b], c, base + r0);
        if (rc == CGV_OK && hipEventRecord(done[b], h->stream) != hipSuccess) rc = fail(CGV_ERR_HIP, "cgv_load_mmap: event");
    }
    if (rc == CGV_OK) rc = ingest_finish(h, base + count);
    if (rc != CGV_OK) {  // the file is applied whole or not at all
        const std::string msg = g_err;   // (rolling back may overwrite it)
        (void)ingest_rollback(h, snap);
        (void)fail(rc, msg);
    }
    cleanup();
    if (rc == CGV_OK && out_rows) *out_rows = count;
    return rc;
}

int cgv_write_mmap_f32(const char* path, const float* rows_host, uint64_t n, uint32_t dim) {
    if (!path) return fail(CGV_ERR_INVALID_ARG, "path is NULL");
    if (n == 0) return CGV_OK;  // memory.rs:243-245: empty input writes no file
    if (!rows_host || dim == 0) return fail(CGV_ERR_INVALID_ARG, "rows is NULL or dim is 0");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(CGV_ERR_IO, std::string("Failed to create mmap file: ") + strerror(errno));
    const uint64_t hdr[2] = {n, dim};
    bool ok = fwrite(hdr, 8, 2, f) == 2;
    const size_t total = (size_t)n * dim;
    ok = ok && fwrite(rows_host, 4, total, f) == total;
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(CGV_ERR_IO, std::string("Failed to write to file: ") + strerror(errno));
    return CGV_OK;
}

int cgv_save_mmap(cgv_index* h, const char* path) {
    if (!h || !path) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    if (h->n == 0) return CGV_OK;
    HIPCHK(hipSetDevice(h->device));
    SearchCtx* c = &h->ctx[0];
    hipStream_t s = c->stream;
    if (int orc = order_after_caller(h, c)) return orc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(CGV_ERR_IO, std::string("Failed to create mmap file: ") + strerror(errno));
    const uint64_t hdr[2] = {h->n, h->D};
    bool ok = fwrite(hdr, 8, 2, f) == 2;
    const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)h->D * 4));
    std::vector<float> host((size_t)std::min<uint64_t>(chunk_rows, h->n) * h->D);
    int rc = CGV_OK;
    for (uint64_t r0 = 0; r0 < h->n && ok && rc == CGV_OK; r0 += chunk_rows) {
        const uint64_t cnt = std::min<uint64_t>(chunk_rows, h->n - r0);
        if ((rc = c->qstage.ensure((size_t)cnt * h->D * 4))) break;
        float* tmp = c->qstage.as<float>();
        const dim3 grid((unsigned)cnt), blk(256);
        switch (h->dtype) {
            case CGV_DTYPE_F32: hipLaunchKernelGGL(gather_row_kernel<DT_F32>, grid, blk, 0, s, (const char*)h->rows, r0, h->D, h->ld, (const int8_t*)h->rexp, tmp); break;
            case CGV_DTYPE_BF16: hipLaunchKernelGGL(gather_row_kernel<DT_BF16>, grid, blk, 0, s, (const char*)h->rows, r0, h->D, h->ld, (const int8_t*)h->rexp, tmp); break;
            case CGV_DTYPE_FP16: hipLaunchKernelGGL(gather_row_kernel<DT_FP16>, grid, blk, 0, s, (const char*)h->rows, r0, h->D, h->ld, (const int8_t*)h->rexp, tmp); break;
            default: hipLaunchKernelGGL(gather_row_kernel<DT_FP8>, grid, blk, 0, s, (const char*)h->rows, r0, h->D, h->ld, (const int8_t*)h->rexp, tmp); break;
        }
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(host.data(), tmp, (size_t)cnt * h->D * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(CGV_ERR_HIP, "cgv_save_mmap: device gather failed");
            break;
        }
        ok = fwrite(host.data(), 4, (size_t)cnt * h->D, f) == (size_t)cnt * h->D;
    }
    ok = (fclose(f) == 0) && ok;
    if (rc) return rc;
    if (!ok) return fail(CGV_ERR_IO, std::string("Failed to write to file: ") + strerror(errno));
    return CGV_OK;
}

int cgv_update_row_f32(cgv_index* h, uint64_t id, const float* row_host) {
    if (!h || !row_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    if (id >= h->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    for (uint32_t i = 0; i < h->D; ++i)  // checked BEFORE the stored row is overwritten: a rejected update changes nothing
        if (!(fabsf(row_host[i]) <= 3.402823466e38f))
            return fail(CGV_ERR_NONFINITE, "row contains NaN/Inf (the reference panics on NaN at simd_ops.rs:379); not applied");
    if (h->dtype == CGV_DTYPE_FP8E4M3) {
        float amax = 0.0f;
        for (uint32_t i = 0; i < h->D; ++i) amax = std::max(amax, fabsf(row_host[i]));
        const int e = fp8_row_exponent(amax);
        if (e < FP8_EXP_MIN || e > FP8_EXP_MAX)
            return fail(CGV_ERR_INVALID_ARG, "fp8 storage: the row's largest magnitude is outside [2^-48, 2^48]; not applied");
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = h->addstage.ensure((size_t)h->D * 4))) return rc;
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->addstage.p, row_host, (size_t)h->D * 4, hipMemcpyHostToDevice, s));
    if ((rc = ingest_enqueue(h, h->addstage.as<float>(), 1, id, h->n))) return rc;
    if ((rc = ingest_finish(h, h->n))) return rc;
    return CGV_OK;
}

uint64_t cgv_count(const cgv_index* h) { return h ? h->n : 0; }
uint32_t cgv_dim(const cgv_index* h) { return h ? h->D : 0; }

int cgv_set_index_base(cgv_index* h, uint64_t base) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    h->idmap.base = base;
    return CGV_OK;
}

int cgv_set_id_map(cgv_index* h, uint32_t chunk_rows, uint32_t n_shards, uint32_t shard) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n_shards == 0 || shard >= n_shards || (n_shards > 1 && chunk_rows == 0))
        return fail(CGV_ERR_INVALID_ARG, "cgv_set_id_map: need chunk_rows > 0 and shard < n_shards");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    h->idmap.chunk = n_shards > 1 ? chunk_rows : 0;
    h->idmap.nshards = n_shards;
    h->idmap.shard = shard;
    return CGV_OK;
}

static int check_search_args(cgv_index* h, const void* q, uint32_t k, const void* oi, const void* os) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (k > CGV_MAX_K) return fail(CGV_ERR_INVALID_ARG, "k exceeds CGV_MAX_K");
    if (!q || !oi || !os) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    return CGV_OK;
}

int cgv_search_begin_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k,
                             uint64_t* out_idx_dev, float* out_score_dev, uint64_t* ticket) {
    if (!ticket) return fail(CGV_ERR_INVALID_ARG, "ticket is NULL");
    *ticket = 0;
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || k == 0) return CGV_OK;  // surreal_store.rs:62-64; ticket 0 = nothing to wait for
    int rc = check_search_args(h, queries_dev, k, out_idx_dev, out_score_dev);
    if (rc) return rc;
    std::unique_lock<std::mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    SearchCtx* c = acquire_ctx(h, lk, /*split=*/true);
    if (!c)
        return fail(CGV_ERR_BUSY, "this thread already holds all " + std::to_string(N_CTX) +
                                      " search contexts of the handle: call cgv_search_end on one of its tickets first");
    if ((rc = order_after_caller(h, c)) == CGV_OK) {
        const float* qsrc = fetch_host_queries(h, c, queries_dev, nq, c->stream, &rc);
        if (rc == CGV_OK) rc = search_enqueue(h, c, qsrc, nq, k, out_idx_dev, out_score_dev);
    }
    if (rc) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        c->busy = false;
        dev_inflight_add(h, -1);
        lk.unlock();
        h->cv.notify_all();
        return rc;
    }
    *ticket = ((uint64_t)c->gen << 8) | (uint64_t)((c - h->ctx) + 1);
    return CGV_OK;
}

int cgv_search_end(cgv_index* h, uint64_t ticket) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (ticket == 0) return CGV_OK;
    const uint64_t slot = (ticket & 0xff);
    if (slot == 0 || slot > (uint64_t)N_CTX) return fail(CGV_ERR_INVALID_ARG, "bad ticket");
    SearchCtx* c = &h->ctx[slot - 1];
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!c->busy || c->gen != (uint32_t)(ticket >> 8)) return fail(CGV_ERR_INVALID_ARG, "stale ticket");
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = search_finish(h, c);
    if (rc) (void)hipStreamSynchronize(c->cur());
    c->on_caller = false;  // (a packed ticket ended here: its provisional records are simply not refreshed)
    c->rec_out = nullptr;
    release_ctx(h, c);
    return rc;
}

uint32_t cgv_max_batches_in_flight(const cgv_index* h) { return h ? (uint32_t)N_CTX : 0u; }

// ---- join-free exchange (one process per GPU) -------------------------------------------------------
// Round 3 joined the host on the shard search, THEN launched pack / all-gather / merge onto an idle GPU (0.09-0.14 ms per
// batch at C2, profiles/r03_c2_force_dist_bench.json). Here the records are packed on the search's own stream right behind
// its last kernel and the consumer stream is made to wait for them with an event: the caller enqueues the collective and
// the merge while the coarse kernel is still running and synchronises once. Queries the device could not prove travel as
// PROVISIONAL records (pack_topk_kernel); cgv_search_packed_end re-runs them through the exact scan and re-packs.
static void launch_pack(const uint64_t* idx, const float* score, uint32_t nq, uint32_t k, uint32_t* rec, const uint32_t* prov,
                        uint32_t prov_all, hipStream_t s) {
    const uint64_t total = (uint64_t)nq * packed_width(k);
    hipLaunchKernelGGL(pack_topk_kernel, dim3((unsigned)std::min<uint64_t>(1024, (total + 255) / 256)), dim3(256), 0, s, idx, score,
                       nq, k, rec, prov, prov_all);
}

int cgv_search_packed_begin_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k, uint32_t* rec_out_dev,
                                    void* consumer_stream, uint64_t* ticket) {
    if (!ticket) return fail(CGV_ERR_INVALID_ARG, "ticket is NULL");
    *ticket = 0;
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || k == 0) return CGV_OK;
    int rc = check_search_args(h, queries_dev, k, rec_out_dev, rec_out_dev);
    if (rc) return rc;
    std::unique_lock<std::mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    SearchCtx* c = acquire_ctx(h, lk, /*split=*/true);
    if (!c)
        return fail(CGV_ERR_BUSY, "this thread already holds all " + std::to_string(N_CTX) +
                                      " search contexts of the handle: call cgv_search_packed_end on one of its tickets first");
    c->on_caller = true;   // the batch runs on the consumer's stream, in line with the collective and the merge behind it
    c->run = (hipStream_t)consumer_stream;
    auto body = [&]() -> int {
        int r;
        if ((r = c->outidx.ensure((size_t)nq * k * 8))) return r;
        if ((r = c->outscore.ensure((size_t)nq * k * 4))) return r;
        HIPCHK(hipEventRecord(c->dep, h->stream));   // (also what a copy-engine fetch of the queries waits for: fetch_host_queries)
        if (h->stream != c->run)   // what the caller queued on the handle's stream (ingest, a query producer) comes first
            HIPCHK(hipStreamWaitEvent(c->run, c->dep, 0));
        const float* qsrc = fetch_host_queries(h, c, queries_dev, nq, c->run, &r);
        if (r) return r;
        if ((r = search_enqueue(h, c, qsrc, nq, k, c->outidx.as<uint64_t>(), c->outscore.as<float>()))) return r;
        // exact-scan-only batches (f32 index, forced exact, k beyond the fast path) are produced by search_finish: every
        // record is provisional. An empty index pads its results at enqueue time: final.
        const bool all_prov = !c->mfma;
        launch_pack(c->outidx.as<uint64_t>(), c->outscore.as<float>(), nq, k, rec_out_dev,
                    c->mfma && h->n ? c->fbflag.as<uint32_t>() : nullptr, all_prov ? 1u : 0u, c->run);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->packed_done, c->run));  // what this search's end waits for (not the consumer's whole stream)
        c->wait_packed = true;
        return CGV_OK;
    };
    rc = body();
    if (rc) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamSynchronize(c->run);
        c->on_caller = false;
        c->busy = false;
        dev_inflight_add(h, -1);
        lk.unlock();
        h->cv.notify_all();
        return rc;
    }
    c->rec_out = rec_out_dev;
    *ticket = ((uint64_t)c->gen << 8) | (uint64_t)((c - h->ctx) + 1);
    return CGV_OK;
}

int cgv_search_packed_end(cgv_index* h, uint64_t ticket, int* repacked) {
    if (repacked) *repacked = 0;
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (ticket == 0) return CGV_OK;
    const uint64_t slot = (ticket & 0xff);
    if (slot == 0 || slot > (uint64_t)N_CTX) return fail(CGV_ERR_INVALID_ARG, "bad ticket");
    SearchCtx* c = &h->ctx[slot - 1];
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!c->busy || c->gen != (uint32_t)(ticket >> 8) || !c->rec_out) return fail(CGV_ERR_INVALID_ARG, "stale ticket");
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = search_finish(h, c);
    if (rc == CGV_OK && c->rewrote) {  // the exact scan replaced (some of) the results: final records now
        launch_pack(c->outidx.as<uint64_t>(), c->outscore.as<float>(), c->nq, c->k, c->rec_out, nullptr, 0u, c->cur());
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->cur());
        if (e != hipSuccess) rc = fail(CGV_ERR_HIP, std::string("re-pack: ") + hipGetErrorString(e));
        if (repacked) *repacked = 1;
    }
    if (rc) (void)hipStreamSynchronize(c->cur());
    c->rec_out = nullptr;
    c->on_caller = false;
    release_ctx(h, c);
    return rc;
}

int cgv_search_f32_dev(cgv_index* h, const float* queries_dev, uint32_t nq, uint32_t k, uint64_t* out_idx_dev,
                       float* out_score_dev) {
    uint64_t t = 0;
    int rc = cgv_search_begin_f32_dev(h, queries_dev, nq, k, out_idx_dev, out_score_dev, &t);
    if (rc) return rc;
    return cgv_search_end(h, t);
}

// One cgv_search_f32 call by itself: host queries in, host results out, one search context.
static int search_host_plain(cgv_index* h, const float* queries_host, uint32_t nq, uint32_t k, uint64_t* out_idx_host,
                             float* out_score_host) {
    int rc;
    std::unique_lock<std::mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    SearchCtx* c = acquire_ctx(h, lk);
    if (!c) return fail(CGV_ERR_BUSY, "this thread holds every search context of the handle (cgv_search_begin without cgv_search_end)");
    hipStream_t s = c->stream;
#ifdef CGV_ABLATE_BUILD
    static const bool tracing = getenv("CGV_TRACE") != nullptr;
#else
    constexpr bool tracing = false;
#endif
    const auto t_in = std::chrono::steady_clock::now();
    auto stamp = [&](int i) {
        if (tracing) c->host_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count();
    };
    auto body = [&]() -> int {
        int r;
        // Pinned (hipHostMalloc / hipHostRegister) caller buffers are used IN PLACE: the query conversion kernel reads
        // the f32 batch over PCIe while it converts, the last kernel writes ids and scores straight into the caller's
        // arrays - no staging copies, no copy-engine launches on the critical path (r03a: -60 us per C2 step).
        // Pageable buffers go through the context's staging buffers as before.
        const float* qsrc = (tun().zero_copy & 1) ? (const float*)device_alias(queries_host, (size_t)nq * h->D * 4) : nullptr;
        uint64_t* oi = (tun().zero_copy & 2) ? (uint64_t*)device_alias(out_idx_host, (size_t)nq * k * 8) : nullptr;
        float* os = oi ? (float*)device_alias(out_score_host, (size_t)nq * k * 4) : nullptr;
        bool direct_out = oi && os;
        const size_t qbytes = (size_t)nq * h->D * 4, ibytes = (size_t)nq * k * 8, sbytes = (size_t)nq * k * 4;
        // small pageable buffers: through the context's pinned staging area by host memcpy (SearchCtx::h_stage)
        const bool small_q = !qsrc && qbytes <= SMALL_Q_BYTES && tun().zero_copy != 0;
        const bool small_out = !direct_out && ibytes + sbytes <= SMALL_OUT_BYTES && tun().zero_copy != 0;
        if (small_q) {
            memcpy(c->h_stage, queries_host, qbytes);
            qsrc = (const float*)c->h_stage_dev;
        }
        if (small_out) {
            oi = (uint64_t*)(c->h_stage_dev + SMALL_Q_BYTES);
            os = (float*)(c->h_stage_dev + SMALL_Q_BYTES + ibytes);
            direct_out = true;   // (for the device: the last kernel writes host memory; the copy to the caller's arrays is below)
        }
        if (!qsrc && (r = c->qstage.ensure(qbytes))) return r;
        if (!direct_out) {
            if ((r = c->outidx.ensure(ibytes))) return r;
            if ((r = c->outscore.ensure(sbytes))) return r;
            oi = c->outidx.as<uint64_t>();
            os = c->outscore.as<float>();
        }
        if ((r = order_after_caller(h, c))) return r;
        stamp(0);
        if (!qsrc) {
            HIPCHK(hipMemcpyAsync(c->qstage.p, queries_host, qbytes, hipMemcpyHostToDevice, s));
            qsrc = c->qstage.as<float>();
        } else if (!small_q) {
            // several threads in cgv_search_f32 at once (a Send + Sync store called from a multi-thread runtime): other batches are
            // computing, so this one's pinned queries come by the copy engine instead of holding conversion waves on the CUs
            // for the length of the PCIe transfer (fetch_host_queries); a lone call reads them in place
            qsrc = fetch_host_queries(h, c, qsrc, nq, s, &r);
            if (r) return r;
        }
        stamp(1);
        if ((r = search_enqueue(h, c, qsrc, nq, k, oi, os))) return r;
        stamp(2);
        lk.unlock();
        // MFMA path: the results exist once the enqueued pipeline has run, so their D2H copies ride the same
        // stream and ONE host synchronisation (inside search_finish) covers flags and results; only when the
        // exact scan then rewrote some queries (fallbacks, f32 index) are they copied again.
        const bool early = (c->mfma || c->exact_enqueued) && !direct_out;
        auto copy_out = [&]() -> int {
            HIPCHK(hipMemcpyAsync(out_idx_host, c->outidx.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(out_score_host, c->outscore.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
            return CGV_OK;
        };
        if (early && (r = copy_out())) return r;
        stamp(3);
        if ((r = search_finish(h, c))) return r;
        stamp(4);
        if (!direct_out && (!early || c->rewrote)) {
            if ((r = copy_out())) return r;
            HIPCHK(hipStreamSynchronize(s));
        }
        if (small_out) {   // (every path through search_finish ends with the stream idle: the staged results are complete)
            memcpy(out_idx_host, c->h_stage + SMALL_Q_BYTES, ibytes);
            memcpy(out_score_host, c->h_stage + SMALL_Q_BYTES + ibytes, sbytes);
        }
        return CGV_OK;
    };
    rc = body();
    if (lk.owns_lock()) lk.unlock();
    if (rc) {   // nothing of this call may still be in flight when the context goes back to the pool (the copy stream too)
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamSynchronize(s);
    }
    release_ctx(h, c);
    return rc;
}

// ---- group commit of concurrent small calls (coalesce.h) ---------------------------------------------------------------
// k class of a request: batches carry one class, so that a caller asking for a few neighbours is never dragged onto the path
// a large-k neighbour of the queue needs (0: one COARSE_TOP2 launch; 1: the staged MFMA path; 2: the exact scan).
static uint32_t coalesce_kclass(const cgv_index* h, uint32_t k) {
    const uint32_t kp = h->shadow ? std::min<uint32_t>(((4 * k + 16 + 7) / 8) * 8, 256u) : kprime_of(k, h->dtype == CGV_DTYPE_FP8E4M3);
    if (kp <= 64) return 0u;
    return (k <= CGV_FAST_MAX_K && kp <= CAND_CAPS && (!h->shadow || k <= 60)) ? 1u : 2u;
}

// ONE device batch for the requests of several callers: queries gathered into the context's pinned staging area, one search
// with k = kmax, every caller's first k results scattered to its own buffers. Sets DONE / ALONE on every request.
static void search_coalesced(cgv_index* h, std::vector<CoReq*>& batch, uint32_t kmax) {
    const size_t row_bytes = (size_t)h->D * 4;
    for (CoReq* r : batch) r->outcome = CoReq::PENDING;
    auto all_alone = [&]() {
        for (CoReq* r : batch) r->outcome = CoReq::ALONE;
    };
    uint32_t nq_total = 0;   // (every caller has checked its own queries before it joined: cgv_search_f32)
    for (CoReq* r : batch) {
        r->off = nq_total;
        nq_total += r->nq;
    }
    if (nq_total == 0) return;
    std::unique_lock<std::mutex> lk(h->mu);
    if (hipSetDevice(h->device) != hipSuccess) {
        (void)hipGetLastError();
        return all_alone();
    }
    SearchCtx* c = acquire_ctx(h, lk);
    if (!c) return all_alone();   // (this thread holds every context as begin tickets: each caller's own call reports it)
    hipStream_t s = c->stream;
    const size_t ibytes = (size_t)nq_total * kmax * 8, sbytes = (size_t)nq_total * kmax * 4;
    for (CoReq* r : batch)
        if (r->outcome == CoReq::PENDING) memcpy(c->h_stage + (size_t)r->off * row_bytes, r->q, (size_t)r->nq * row_bytes);
    uint64_t* oi = (uint64_t*)(c->h_stage_dev + SMALL_Q_BYTES);
    float* os = (float*)(c->h_stage_dev + SMALL_Q_BYTES + ibytes);
    int rc = order_after_caller(h, c);
    if (rc == CGV_OK) rc = search_enqueue(h, c, (const float*)c->h_stage_dev, nq_total, kmax, oi, os);
    lk.unlock();
    if (rc == CGV_OK) rc = search_finish(h, c);   // (every path through it ends with the stream idle: the staged results are complete)
    if (rc != CGV_OK) {   // whatever it was, nobody inherits another caller's failure: every request runs again on its own
        (void)hipStreamSynchronize(s);
        release_ctx(h, c);
        return all_alone();
    }
    const uint64_t* ri = (const uint64_t*)(c->h_stage + SMALL_Q_BYTES);
    const float* rs = (const float*)(c->h_stage + SMALL_Q_BYTES + ibytes);
    (void)sbytes;
    for (CoReq* r : batch) {
        if (r->outcome != CoReq::PENDING) continue;
        for (uint32_t j = 0; j < r->nq; ++j) {
            memcpy(r->out_idx + (size_t)j * r->k, ri + (size_t)(r->off + j) * kmax, (size_t)r->k * 8);
            memcpy(r->out_score + (size_t)j * r->k, rs + (size_t)(r->off + j) * kmax, (size_t)r->k * 4);
        }
        r->rc = CGV_OK;
        r->outcome = CoReq::DONE;
    }
    release_ctx(h, c);
}

int cgv_search_f32(cgv_index* h, const float* queries_host, uint32_t nq, uint32_t k, uint64_t* out_idx_host,
                   float* out_score_host) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || k == 0) return CGV_OK;
    int rc = check_search_args(h, queries_host, k, out_idx_host, out_score_host);
    if (rc) return rc;
    // A few queries per call - the reference's trait-level call is ONE (traits.rs:14; surreal_store.rs:61-85), its multi-query
    // caller is B concurrent single-query searches (search.rs:358-361): concurrent callers share one device batch (coalesce.h).
    if (!h->co.eligible(nq, k, h->D)) return search_host_plain(h, queries_host, nq, k, out_idx_host, out_score_host);
    // isolation: a query the device would reject for a whole batch (NaN / Inf: the reference panics, simd_ops.rs:379; fp8: largest
    // magnitude outside [2^-48, 2^48]) never joins one - the plain path reports it to its own caller exactly as a lone call
    for (uint32_t j = 0; j < nq; ++j) {
        const float* v = queries_host + (size_t)j * h->D;
        float amax = 0.0f;
        bool finite = true;
        for (uint32_t i = 0; i < h->D; ++i) {
            const float a = fabsf(v[i]);
            finite = finite && (a <= 3.402823466e38f);
            amax = a > amax ? a : amax;
        }
        bool ok = finite;
        if (ok && h->dtype == CGV_DTYPE_FP8E4M3 && amax > 0.0f) {
            const int e = fp8_row_exponent(amax);
            ok = e >= FP8_EXP_MIN && e <= FP8_EXP_MAX;
        }
        if (!ok) return search_host_plain(h, queries_host, nq, k, out_idx_host, out_score_host);
    }
    CoReq r;
    r.q = queries_host;
    r.nq = nq;
    r.k = k;
    r.kclass = coalesce_kclass(h, k);
    r.out_idx = out_idx_host;
    r.out_score = out_score_host;
    return h->co.submit(
        r, h->D, [&]() { return search_host_plain(h, queries_host, nq, k, out_idx_host, out_score_host); },
        [&](std::vector<CoReq*>& batch, uint32_t, uint32_t kmax) { search_coalesced(h, batch, kmax); },
        [](int code, const std::string& msg) { return fail(code, msg); });
}

int cgv_set_coalesce(cgv_index* h, uint32_t max_batch_queries, uint32_t max_batches_in_flight, uint32_t window_us) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (max_batches_in_flight > (uint32_t)N_CTX) return fail(CGV_ERR_INVALID_ARG, "at most cgv_max_batches_in_flight() batches");
    // (applies to calls that arrive from now on; requests already queued are served under whichever values their leader reads)
    h->co.configure(max_batch_queries != 0 && max_batches_in_flight != 0, max_batch_queries, (int)max_batches_in_flight, window_us);
    return CGV_OK;
}

int cgv_get_sample_repair_stats(cgv_index* h, uint64_t* out1) {
    if (!h || !out1) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    out1[0] = h->sample_repairs;
    return CGV_OK;
}

int cgv_get_small_batch_stats(cgv_index* h, uint64_t* out4) {
    if (!h || !out4) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    out4[0] = h->top2_stats[0];
    out4[1] = h->top2_stats[1];
    out4[2] = h->top2_stats[2];
    out4[3] = h->top2_stats[1] - h->top2_stats[2];
    return CGV_OK;
}

int cgv_get_coalesce_stats(cgv_index* h, uint64_t* out8) {
    if (!h || !out8) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    const CoStats st = h->co.stats();
    out8[0] = st.batches;
    out8[1] = st.batched_requests;
    out8[2] = st.batched_queries;
    out8[3] = st.lone_calls;
    out8[4] = st.retried_alone;
    out8[5] = st.max_batch_queries;
    out8[6] = st.window_waits;
    out8[7] = 0;
    return CGV_OK;
}

int cgv_get_row_f32(cgv_index* h, uint64_t id, float* out_host) {
    if (!h || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    SearchCtx* c = &h->ctx[0];
    if (id >= h->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = c->qstage.ensure((size_t)h->D * 4))) return rc;
    hipStream_t s = c->stream;
    if (int orc = order_after_caller(h, c)) return orc;
    float* tmp = c->qstage.as<float>();
    if (h->dtype == CGV_DTYPE_F32)
        hipLaunchKernelGGL(gather_row_kernel<DT_F32>, dim3(1), dim3(256), 0, s, (const char*)h->rows, id, h->D, h->ld, (const int8_t*)h->rexp, tmp);
    else if (h->dtype == CGV_DTYPE_BF16)
        hipLaunchKernelGGL(gather_row_kernel<DT_BF16>, dim3(1), dim3(256), 0, s, (const char*)h->rows, id, h->D, h->ld, (const int8_t*)h->rexp, tmp);
    else if (h->dtype == CGV_DTYPE_FP16)
        hipLaunchKernelGGL(gather_row_kernel<DT_FP16>, dim3(1), dim3(256), 0, s, (const char*)h->rows, id, h->D, h->ld, (const int8_t*)h->rexp, tmp);
    else
        hipLaunchKernelGGL(gather_row_kernel<DT_FP8>, dim3(1), dim3(256), 0, s, (const char*)h->rows, id, h->D, h->ld, (const int8_t*)h->rexp, tmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_host, tmp, (size_t)h->D * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CGV_OK;
}

// One query against the first `limit_rows` stored rows (0 = all): the reference's building blocks
// evaluated on device with their exact f32 operation order.
static int prep_single_query(cgv_index* h, SearchCtx* c, const float* query_host, hipStream_t s) {
    int rc;
    if ((rc = c->qstage.ensure((size_t)h->D * 4))) return rc;
    if ((rc = c->qrows.ensure(storage_bytes(h, 1)))) return rc;
    if ((rc = c->qnorm.ensure(4))) return rc;
    if ((rc = c->qinvn.ensure(4))) return rc;
    if ((rc = c->qrexp.ensure(16))) return rc;
    if ((rc = c->qlist.ensure(4))) return rc;
    HIPCHK(hipMemcpyAsync(c->qstage.p, query_host, (size_t)h->D * 4, hipMemcpyHostToDevice, s));
    c->flags_clean = false;
    HIPCHK(hipMemsetAsync(c->flags + F_NONFINITE_Q, 0, (F_COUNT - F_NONFINITE_Q) * 4, s));
    HIPCHK(hipMemsetAsync(c->qlist.p, 0, 4, s));
    return prep_dispatch(h->dtype, c->qstage.as<float>(), 1, h->D, h->ld, 0, c->qrows.as<char>(),
                         c->qnorm.as<float>(), c->qinvn.as<float>(), c->qrexp.as<int8_t>(), c->flags + F_NONFINITE_Q, s);
}

int cgv_batch_similarity_f32(cgv_index* h, const float* query_host, int op, uint64_t limit_rows, float* out_host) {
    if (!h || !query_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    if (op < 0 || op > OP_COSINE_SCALAR || op == OP_NEG_COSINE_DISTANCE_SEQ) return fail(CGV_ERR_INVALID_ARG, "unknown op");
    if (h->dtype == CGV_DTYPE_FP8E4M3 && (op == OP_DOT || op == OP_L2))
        return fail(CGV_ERR_INVALID_ARG, "fp8 storage is per-row scaled: only the (scale-invariant) cosine ops");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    SearchCtx* c = &h->ctx[0];
    HIPCHK(hipSetDevice(h->device));
    const uint64_t n = limit_rows ? std::min<uint64_t>(limit_rows, h->n) : h->n;
    if (n == 0) return CGV_OK;
    hipStream_t s = c->stream;
    if (int orc = order_after_caller(h, c)) return orc;
    int rc;
    if ((rc = prep_single_query(h, c, query_host, s))) return rc;
    if ((rc = c->scores.ensure((size_t)h->n * 4))) return rc;
    switch (h->dtype) {
        case CGV_DTYPE_F32: launch_exact_scores<DT_F32>(h, c, c->qlist.as<uint32_t>(), 1, c->scores.as<float>(), op, s); break;
        case CGV_DTYPE_BF16: launch_exact_scores<DT_BF16>(h, c, c->qlist.as<uint32_t>(), 1, c->scores.as<float>(), op, s); break;
        case CGV_DTYPE_FP8E4M3: launch_exact_scores<DT_FP8>(h, c, c->qlist.as<uint32_t>(), 1, c->scores.as<float>(), op, s); break;
        default: launch_exact_scores<DT_FP16>(h, c, c->qlist.as<uint32_t>(), 1, c->scores.as<float>(), op, s); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_host, c->scores.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CGV_OK;
}

// Shared body of cgv_score_ids_f32 (dense [nq][m] ids, qsel_host == NULL) and cgv_score_pairs_f32_ (pair list).
static int score_pairs(cgv_index* h, const float* queries_host, uint32_t nq, int op, const uint32_t* qsel_host,
                       const uint64_t* ids_host, uint64_t npairs, uint32_t m, float* out_host) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || npairs == 0) return CGV_OK;
    if (!queries_host || !ids_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    if (op < 0 || op > OP_COSINE_SCALAR || op == OP_NEG_COSINE_DISTANCE_SEQ) return fail(CGV_ERR_INVALID_ARG, "unknown op");
    if (h->dtype == CGV_DTYPE_FP8E4M3 && (op == OP_DOT || op == OP_L2))
        return fail(CGV_ERR_INVALID_ARG, "fp8 storage is per-row scaled: only the (scale-invariant) cosine ops");
    if (qsel_host)
        for (uint64_t i = 0; i < npairs; ++i)
            if (qsel_host[i] >= nq) return fail(CGV_ERR_INVALID_ARG, "pair list: query index out of range");
    std::unique_lock<std::mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    SearchCtx* c = acquire_ctx(h, lk);
    if (!c) return fail(CGV_ERR_BUSY, "this thread holds every search context of the handle");
    hipStream_t s = c->stream;
    const uint64_t n = h->n;
    auto body = [&]() -> int {
        int r;
        const size_t qb = (size_t)nq * h->D * 4, ib = (size_t)npairs * 8, ob = (size_t)npairs * 4, sb = qsel_host ? (size_t)npairs * 4 : 0;
        if ((r = c->qstage.ensure(qb))) return r;
        if ((r = c->outidx.ensure(ib))) return r;
        if ((r = c->outscore.ensure(ob))) return r;
        if (sb && (r = c->qlist.ensure(sb))) return r;
        if ((r = order_after_caller(h, c))) return r;
        lk.unlock();
        HIPCHK(hipMemcpyAsync(c->qstage.p, queries_host, qb, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->outidx.p, ids_host, ib, hipMemcpyHostToDevice, s));
        if (sb) HIPCHK(hipMemcpyAsync(c->qlist.p, qsel_host, sb, hipMemcpyHostToDevice, s));
        const dim3 grid((unsigned)((npairs + 31) / 32)), blk(256);
        const float* qd = c->qstage.as<float>();
        const uint64_t* idd = c->outidx.as<uint64_t>();
        const uint32_t* qs = sb ? c->qlist.as<uint32_t>() : nullptr;
        float* od = c->outscore.as<float>();
        switch (h->dtype) {
            case CGV_DTYPE_F32: hipLaunchKernelGGL(score_ids_kernel<DT_F32>, grid, blk, 0, s, (const char*)h->rows, qd, idd, qs, npairs, m, n, h->D, h->ld, op, od); break;
            case CGV_DTYPE_BF16: hipLaunchKernelGGL(score_ids_kernel<DT_BF16>, grid, blk, 0, s, (const char*)h->rows, qd, idd, qs, npairs, m, n, h->D, h->ld, op, od); break;
            case CGV_DTYPE_FP16: hipLaunchKernelGGL(score_ids_kernel<DT_FP16>, grid, blk, 0, s, (const char*)h->rows, qd, idd, qs, npairs, m, n, h->D, h->ld, op, od); break;
            default: hipLaunchKernelGGL(score_ids_kernel<DT_FP8>, grid, blk, 0, s, (const char*)h->rows, qd, idd, qs, npairs, m, n, h->D, h->ld, op, od); break;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out_host, od, ob, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return CGV_OK;
    };
    const int rc = body();
    if (lk.owns_lock()) lk.unlock();
    if (rc) (void)hipStreamSynchronize(s);
    release_ctx(h, c);
    return rc;
}

int cgv_score_ids_f32(cgv_index* h, const float* queries_host, uint32_t nq, int op, const uint64_t* ids_host, uint32_t m,
                      float* out_host) {
    if (m == 0) return h ? CGV_OK : fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    return score_pairs(h, queries_host, nq, op, nullptr, ids_host, (uint64_t)nq * m, m, out_host);
}

// internal (sharded.hip): the pairs (query qsel[p], row ids[p]) of one shard, p < npairs; out[p] = the score.
int cgv_score_pairs_f32_(cgv_index* h, const float* queries_host, uint32_t nq, int op, const uint32_t* qsel_host,
                         const uint64_t* ids_host, uint64_t npairs, float* out_host) {
    if (npairs && !qsel_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    return score_pairs(h, queries_host, nq, op, qsel_host, ids_host, npairs, 1u, out_host);
}

int cgv_truncate(cgv_index* h, uint64_t n_rows) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    if (n_rows >= h->n) return CGV_OK;
    HIPCHK(hipSetDevice(h->device));
    // the corpus-wide maxima (largest norm, shadow residuals) stay as they are: over-estimates only widen the
    // error bound of the exactness check, they never invalidate it
    return ingest_rollback(h, IngestSnapshot{n_rows, h->max_norm_c, h->res_rel_c, h->res_abs_c});
}

int cgv_search_baseline_f32(cgv_index* h, const float* query_host, uint32_t limit, uint64_t* out_idx_host,
                            float* out_dist_host, uint32_t* out_n) {
    if (!h || !query_host || !out_idx_host || !out_n) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    *out_n = 0;
    if (limit > CGV_MAX_K) return fail(CGV_ERR_INVALID_ARG, "limit exceeds CGV_MAX_K");
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    SearchCtx* c = &h->ctx[0];
    if (h->n == 0 || limit == 0) return CGV_OK;  // optimization.rs:382-384
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = c->stream;
    if (int orc = order_after_caller(h, c)) return orc;
    int rc;
    if ((rc = prep_single_query(h, c, query_host, s))) return rc;
    if ((rc = c->outidx.ensure((size_t)limit * 8))) return rc;
    if ((rc = c->outscore.ensure((size_t)limit * 4))) return rc;
    // ascending distance, stable (ties keep index order) == descending (-distance, index asc)
    if ((rc = exact_search(h, c, c->qlist.as<uint32_t>(), 1, limit, c->outidx.as<uint64_t>(), c->outscore.as<float>(), s,
                           OP_NEG_COSINE_DISTANCE_SEQ, /*local_ids=*/true)))
        return rc;
    std::vector<float> sc(limit);
    HIPCHK(hipMemcpyAsync(out_idx_host, c->outidx.p, (size_t)limit * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sc.data(), c->outscore.p, (size_t)limit * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint32_t m = 0;
    while (m < limit && out_idx_host[m] != UINT64_MAX) ++m;
    for (uint32_t i = 0; i < m; ++i) {
        if (out_dist_host) out_dist_host[i] = -sc[i];
    }
    *out_n = m;
    return CGV_OK;
}

static int normalize_rows_host(int device_id, float* rows_host, uint64_t n, uint32_t dim, bool scalar_arm) {
    if (n == 0 || dim == 0) return CGV_OK;  // simd_ops.rs:190-192
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    if (cgv_device_count() == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    HIPCHK(hipSetDevice(device_id));
    float* d = nullptr;
    const size_t bytes = (size_t)n * dim * 4;
    HIPCHK(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemcpy(d, rows_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (scalar_arm) hipLaunchKernelGGL(normalize_rows_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n, dim);
        else hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, 0, d, n, dim);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(rows_host, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CGV_ERR_HIP, std::string("cgv_normalize_rows: ") + hipGetErrorString(e));
    return CGV_OK;
}

int cgv_normalize_rows_f32(int device_id, float* rows_host, uint64_t n, uint32_t dim) {
    return normalize_rows_host(device_id, rows_host, n, dim, false);
}

int cgv_normalize_rows_scalar_f32(int device_id, float* rows_host, uint64_t n, uint32_t dim) {
    return normalize_rows_host(device_id, rows_host, n, dim, true);
}

int cgv_merge_topk_dev(int device_id, const uint64_t* idx_dev, const float* score_dev, uint32_t g, uint32_t nq,
                       uint32_t k, uint64_t* out_idx_dev, float* out_score_dev, void* stream) {
    if (nq == 0 || k == 0) return CGV_OK;
    if (!idx_dev || !score_dev || !out_idx_dev || !out_score_dev || g == 0) return fail(CGV_ERR_INVALID_ARG, "bad argument");
    if (g > 64) return fail(CGV_ERR_INVALID_ARG, "more than 64 partial lists per query");
    HIPCHK(hipSetDevice(device_id));
    if ((uint64_t)g * k > 4096) {  // beyond the LDS merge: G-way wave merge (any k)
        hipLaunchKernelGGL(merge_topk_wave_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const char*)idx_dev,
                           (uint64_t)k * 8, (const char*)score_dev, (uint64_t)k * 4, g, nq, k, out_idx_dev, out_score_dev);
        HIPCHK(hipGetLastError());
        return CGV_OK;
    }
    const uint32_t P = next_pow2(std::max<uint32_t>(g * k, 2));
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), (size_t)P * 16, (hipStream_t)stream,
                       (const char*)idx_dev, (uint64_t)k * 8, (const char*)score_dev, (uint64_t)k * 4, g, nq, k,
                       out_idx_dev, out_score_dev);
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

uint32_t cgv_packed_width(uint32_t k) { return packed_width(k); }

void* cgv_alloc_pinned(size_t bytes) {
    if (cgv_device_count() == 0) {
        (void)fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
        return nullptr;
    }
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)fail(e == hipErrorOutOfMemory ? CGV_ERR_OOM : CGV_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

int cgv_free_pinned(void* p) {
    if (!p) return CGV_OK;
    HIPCHK(hipHostFree(p));
    return CGV_OK;
}

void* cgv_host_device_alias(int device_id, const void* host_ptr, size_t bytes) {
    if (!host_ptr || hipSetDevice(device_id) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return device_alias(host_ptr, bytes);
}

int cgv_pack_topk_dev(int device_id, const uint64_t* idx_dev, const float* score_dev, uint32_t nq, uint32_t k,
                      uint32_t* out_rec_dev, void* stream) {
    if (nq == 0 || k == 0) return CGV_OK;
    if (!idx_dev || !score_dev || !out_rec_dev) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    HIPCHK(hipSetDevice(device_id));
    const uint64_t total = (uint64_t)nq * packed_width(k);
    hipLaunchKernelGGL(pack_topk_kernel, dim3((unsigned)std::min<uint64_t>(1024, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, idx_dev, score_dev, nq, k, out_rec_dev);
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

int cgv_merge_packed_flag_dev(int device_id, const uint32_t* rec_dev, uint32_t g, uint32_t nq, uint32_t k,
                              uint64_t* out_idx_dev, float* out_score_dev, uint32_t* redo_flag_dev, void* stream) {
    if (nq == 0 || k == 0) return CGV_OK;
    if (!rec_dev || !out_idx_dev || !out_score_dev || g == 0) return fail(CGV_ERR_INVALID_ARG, "bad argument");
    if (g > 64) return fail(CGV_ERR_INVALID_ARG, "more than 64 partial lists per query");
    HIPCHK(hipSetDevice(device_id));
    // pinned HOST result arrays (and redo word) are written in place by the merge kernel (no D2H copies for the caller to enqueue)
    if (void* al = device_alias(out_idx_dev, (size_t)nq * k * 8)) out_idx_dev = (uint64_t*)al;
    if (void* al = device_alias(out_score_dev, (size_t)nq * k * 4)) out_score_dev = (float*)al;
    if (redo_flag_dev)
        if (void* al = device_alias(redo_flag_dev, 4)) redo_flag_dev = (uint32_t*)al;
    const uint64_t stride = (uint64_t)packed_width(k) * 4;
    if ((uint64_t)g * k > 4096) {  // beyond the LDS merge: G-way wave merge (any k)
        hipLaunchKernelGGL(merge_topk_wave_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const char*)rec_dev, stride,
                           (const char*)rec_dev + (uint64_t)k * 8, stride, g, nq, k, out_idx_dev, out_score_dev, redo_flag_dev);
        HIPCHK(hipGetLastError());
        return CGV_OK;
    }
    const uint32_t P = next_pow2(std::max<uint32_t>(g * k, 2));
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), (size_t)P * 16, (hipStream_t)stream,
                       (const char*)rec_dev, stride, (const char*)rec_dev + (uint64_t)k * 8, stride, g, nq, k,
                       out_idx_dev, out_score_dev, redo_flag_dev);
    HIPCHK(hipGetLastError());
    return CGV_OK;
}

int cgv_merge_packed_dev(int device_id, const uint32_t* rec_dev, uint32_t g, uint32_t nq, uint32_t k,
                         uint64_t* out_idx_dev, float* out_score_dev, void* stream) {
    return cgv_merge_packed_flag_dev(device_id, rec_dev, g, nq, k, out_idx_dev, out_score_dev, nullptr, stream);
}

int cgv_set_stream(cgv_index* h, void* stream) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    // No wait for searches in flight: they already recorded their ordering event on the old stream;
    // the new value only decides where LATER calls order themselves (a pipelining caller re-points
    // the stream between cgv_search_begin calls).
    std::lock_guard<std::mutex> lk(h->mu);
    h->stream = (hipStream_t)stream;  // NULL == HIP's legacy default stream
    return CGV_OK;
}

int cgv_use_own_stream(cgv_index* h) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    h->stream = h->own_stream;
    return CGV_OK;
}

int cgv_synchronize(cgv_index* h) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (SearchCtx& c : h->ctx) HIPCHK(hipStreamSynchronize(c.stream));
    return CGV_OK;
}

int cgv_get_stats(cgv_index* h, cgv_stats* out) {
    if (!h || !out) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    h->st.n_rows = h->n;
    h->st.device_bytes = device_bytes(h);
    *out = h->st;
    out->coarse_rows = h->last_coarse_rows;
    return CGV_OK;
}

int cgv_get_phase_times(cgv_index* h, float* out_us4) {
    if (!h || !out_us4) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < 4; ++i) out_us4[i] = h->last_phase_us[i];
    return CGV_OK;
}

int cgv_set_profiling(cgv_index* h, int enabled) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    h->profiling = enabled < 0 ? 0 : enabled;
    return CGV_OK;
}

int cgv_set_spin_us(cgv_index* h, uint32_t spin_us) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    h->spin_us = (long)std::min<uint32_t>(spin_us, 1000000u);
    return CGV_OK;
}

int cgv_set_force_exact(cgv_index* h, int enabled) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    h->force_exact = enabled != 0;
    return CGV_OK;
}

int cgv_debug_coarse_scores_dev(cgv_index* h, const float* queries_dev, uint32_t nq, float* out_dev) {
    if (!h || !queries_dev || !out_dev) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    if (h->dtype == CGV_DTYPE_F32 && !h->shadow)
        return fail(CGV_ERR_INVALID_ARG, "coarse path needs a bf16/fp16/fp8 index or an f32 index with a shadow");
    if (nq == 0 || h->n == 0) return CGV_OK;
    std::unique_lock<std::mutex> lk(h->mu);
    if (int brc = wait_all_idle(h, lk)) return brc;
    SearchCtx* c = &h->ctx[0];
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = c->stream;
    if (int orc = order_after_caller(h, c)) return orc;
    int rc;
    if ((rc = c->qrows.ensure(storage_bytes(h, nq)))) return rc;
    if ((rc = c->qnorm.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->qinvn.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->qrexp.ensure((size_t)nq + 16))) return rc;
    if ((rc = c->tau.ensure((size_t)nq * 4))) return rc;
    if ((rc = c->overflow.ensure((size_t)nq * 4))) return rc;
    c->flags_clean = false;
    rc = prep_dispatch(h->dtype, queries_dev, nq, h->D, h->ld, 0, c->qrows.as<char>(), c->qnorm.as<float>(),
                       c->qinvn.as<float>(), c->qrexp.as<int8_t>(), c->flags + F_NONFINITE_Q, s);
    if (rc) return rc;
    if (h->shadow) {
        if ((rc = c->qshadow.ensure(shadow_bytes(h, nq)))) return rc;
        if ((rc = c->qres.ensure((size_t)nq * 8))) return rc;
        hipLaunchKernelGGL(shadow_rows_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, queries_dev, (uint64_t)nq, h->D, h->lds,
                           (uint64_t)0, c->qshadow.as<char>(), c->qnorm.as<float>(), c->qinvn.as<float>(),
                           c->qres.as<float>(), (uint32_t*)nullptr);
        HIPCHK(hipGetLastError());
    }
    const int cdt = h->shadow ? CGV_DTYPE_BF16 : h->dtype;
    const uint32_t nqt = (nq + BN - 1) / BN;
    const uint32_t nsplit_max = std::max<uint32_t>(1u, (uint32_t)h->n_cu / nqt);
    const uint32_t ntiles = (uint32_t)((h->n + BM - 1) / BM);
    const uint32_t nsplit = std::min(ntiles, nsplit_max);
    if ((rc = c->cand.ensure((size_t)nqt * nsplit * BN * CAND_CAPS * 8))) return rc;
    if ((rc = c->candcnt.ensure((size_t)nqt * nsplit * BN * 4))) return rc;
    hipLaunchKernelGGL(fill_f32_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, c->tau.as<float>(), -INFINITY, nq);
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
    a.dump = out_dev;
    a.n = (uint32_t)h->n;
    a.nq = nq;
    a.ld = h->shadow ? h->lds : h->ld;
    a.kc = a.ld / kchunk_of(cdt);
    a.T1 = 0;
    a.R = ntiles;
    a.P = 1;
    a.j0 = 0;
    a.cnt = ntiles;
    a.nsplit = nsplit;
    a.nqt = nqt;
    a.metric = h->metric;
    a.qgroup = query_group(nqt, a.ld, cdt);
    a.rexp_c = h->rexp;
    a.rexp_q = c->qrexp.as<int8_t>();
    a.pace = nullptr;
    a.floor_ord = nullptr;
    a.cellb = nullptr;
    a.sample_emit = 0;
    a.sample_floor = 0;
    a.scand = nullptr;
    a.append_splits = 0;
    a.lad = nullptr;
    a.ladc = nullptr;
    a.sample_ld = 0;
    a.sample_vals = 16;
    a.epi = 1;
    a.tau_out = nullptr;
    a.boot_sync = nullptr;
    a.kprime = 0;
    if ((rc = launch_coarse(cdt, COARSE_DUMP, a, nqt * nsplit, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    return CGV_OK;
}

}  // extern "C"
