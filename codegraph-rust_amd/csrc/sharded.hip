// sharded.hip — ONE index handle over several MI355X devices of a node (include/cgvec.h, cgv_sharded_*).
//
// The seam the reference injects a backend through holds a single object
// (`Arc<dyn SurrealVectorBackend>`, crates/codegraph-vector/src/surreal_store.rs:11-22,32-34;
// `Box<dyn VectorStore + Send>`, crates/codegraph-core/src/integration/graph_vector.rs:241-253), so that
// object owns every shard: one process, one worker thread per device (HIP's current device is per
// thread; the shards' pipelines are enqueued concurrently), the single-device library underneath.
//
//   rows     global row id = insertion index, dealt block-cyclically in chunks of CGV_SHARD_CHUNK_ROWS:
//            row r -> shard (r / C) % G, local row (r / C / G) * C + r % C; the shard reports global ids
//            through its id map (cgv_set_id_map), so incremental upserts keep the shards balanced.
//   search   queries: pinned staging -> every device (async H2D on each device's own PCIe link);
//            per shard the single-device pipeline (MFMA coarse + exact re-score), then
//            pack_topk -> ONE exchange of nq * k 12-byte records per shard:
//              CGV_EXCHANGE_RCCL  ncclAllGather over xGMI (one communicator per device, one thread each;
//                                 librccl is dlopen'ed on first use - the library has no link-time
//                                 dependency on it);
//              CGV_EXCHANGE_COPY  hipMemcpyPeerAsync (device-to-device copy when a device is listed
//                                 twice) into the root's gather buffer;
//            the root (shard 0's device) merges G * k records per query with (score desc, id asc)
//            (merge_topk_kernel) and the result goes back to the host.
//            The message is nq * k * 12 B per shard (120 KiB at C2, 960 KiB at C5): latency-bound, the
//            xGMI links never saturate.
// No CPU search path: every shard is a cgv_index, which needs a device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cgvec.h"

extern "C" int cgv_set_error_(int code, const char* msg);  // cgvec.hip: the library's thread-local error message

namespace {

int fail(int code, const std::string& m) { return cgv_set_error_(code, m.c_str()); }

#define SHIP(expr)                                                                                      \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(_e == hipErrorOutOfMemory ? CGV_ERR_OOM : CGV_ERR_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                             \
    } while (0)

// ---- RCCL, loaded on demand -------------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
constexpr int NCCL_INT32 = 2;  // ncclInt32 (rccl.h ncclDataType_t)
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GetErrorString; }
};
std::mutex g_rccl_mu;
Rccl g_rccl;
bool g_rccl_tried = false;
std::string g_rccl_err;  // why the library is unusable (dlerror() text is consumed by the call that reads it)

// why: receives the reason when NULL is returned. CGV_RCCL_LIB overrides the library name (tests: a name that
// cannot be loaded exercises the fall-back to the copy exchange).
const Rccl* load_rccl(std::string* why = nullptr) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl_tried) {
        g_rccl_tried = true;
        // a copy already mapped into the process (e.g. PyTorch's) is found first by its soname
        std::vector<std::string> names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char* forced = getenv("CGV_RCCL_LIB")) names = {forced};
        for (const std::string& name : names) {
            g_rccl.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.lib) break;
            const char* de = dlerror();  // read ONCE: the call clears the message
            g_rccl_err = de ? de : ("dlopen(" + name + ") failed");
        }
        if (g_rccl.lib) {
            g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(g_rccl.lib, "ncclCommInitAll");
            g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
            g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
            g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
            if (!g_rccl.ok()) g_rccl_err = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
        }
    }
    if (g_rccl.ok()) return &g_rccl;
    if (why) *why = g_rccl_err.empty() ? "librccl not loadable" : g_rccl_err;
    return nullptr;
}

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {  // the owning device must be current
        if (need <= bytes) return CGV_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        SHIP(hipMalloc(&p, need));
        bytes = need;
        return CGV_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// One worker thread per shard: jobs are posted by the calling thread, which waits for all of them.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = CGV_OK;
    std::string err;
};

struct Shard {
    int device = 0;
    uint32_t index = 0;
    cgv_index* ix = nullptr;
    hipStream_t xs = nullptr;  // H2D of the queries, pack, exchange
    ncclComm_t comm = nullptr;
    Buf qdev, oidx, osc, rec, gathered, stage;
    std::chrono::steady_clock::time_point t_search_done;
    Worker w;
};

using Clock = std::chrono::steady_clock;

}  // namespace

struct cgv_sharded {
    uint32_t D = 0;
    int metric = 0, dtype = 0;
    uint32_t G = 0;
    std::vector<Shard*> sh;
    uint64_t n = 0;
    int exchange = CGV_EXCHANGE_NONE;
    bool distinct = true;
    std::mutex mu;
    float* pin_q = nullptr;
    size_t pin_q_bytes = 0;
    Buf moidx, mosc;  // merged results, on the root device
    uint64_t searches = 0, queries = 0;
    float last_search_ms = 0.0f, last_exchange_ms = 0.0f;
};

namespace {

constexpr uint64_t C = CGV_SHARD_CHUNK_ROWS;

void worker_main(Shard* s) {
    (void)hipSetDevice(s->device);
    Worker& w = s->w;
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.cv.wait(lk, [&] { return w.has_job || w.quit; });
        if (w.quit) return;
        std::function<int()> job = std::move(w.job);
        w.has_job = false;
        lk.unlock();
        const int rc = job();
        const std::string err = rc ? cgv_last_error() : "";
        lk.lock();
        w.rc = rc;
        w.err = err;
        w.done = true;
        w.cv.notify_all();
    }
}

// Post one job per shard (jobs[i] may be empty = nothing to do), wait for all; first failure wins.
int run_all(cgv_sharded* s, std::vector<std::function<int()>>& jobs) {
    for (uint32_t g = 0; g < s->G; ++g) {
        if (!jobs[g]) continue;
        Worker& w = s->sh[g]->w;
        std::lock_guard<std::mutex> lk(w.mu);
        w.job = std::move(jobs[g]);
        w.has_job = true;
        w.done = false;
        w.cv.notify_all();
    }
    int rc = CGV_OK;
    std::string err;
    for (uint32_t g = 0; g < s->G; ++g) {
        Worker& w = s->sh[g]->w;
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        if (w.rc && !rc) {
            rc = w.rc;
            err = "shard " + std::to_string(g) + " (device " + std::to_string(s->sh[g]->device) + "): " + w.err;
        }
        w.rc = CGV_OK;
    }
    return rc ? fail(rc, err) : CGV_OK;
}

void locate(const cgv_sharded* s, uint64_t id, uint32_t* shard, uint64_t* local) {
    const uint64_t chunk = id / C;
    *shard = (uint32_t)(chunk % s->G);
    *local = (chunk / s->G) * C + id % C;
}

// rows of the first n global rows that live on shard g
uint64_t shard_count(const cgv_sharded* s, uint64_t n, uint32_t g) {
    const uint64_t full = n / C, rem = n % C;  // full chunks, rows of the partial one
    uint64_t cnt = (full / s->G) * C;
    const uint64_t extra = full % s->G;  // chunks beyond the last complete round: shards 0..extra-1
    if (g < extra) cnt += C;
    if (g == extra) cnt += rem;
    return cnt;
}

int set_exchange_locked(cgv_sharded* s, int kind) {
    if (s->G <= 1) {
        s->exchange = CGV_EXCHANGE_NONE;
        return CGV_OK;
    }
    if (kind == CGV_EXCHANGE_COPY) {
        s->exchange = CGV_EXCHANGE_COPY;
        return CGV_OK;
    }
    if (kind != CGV_EXCHANGE_RCCL) return fail(CGV_ERR_INVALID_ARG, "exchange must be CGV_EXCHANGE_RCCL or CGV_EXCHANGE_COPY");
    if (!s->distinct) return fail(CGV_ERR_INVALID_ARG, "RCCL needs distinct devices (one communicator rank per GPU)");
    if (!s->sh[0]->comm) {
        std::string why;
        const Rccl* r = load_rccl(&why);
        if (!r) return fail(CGV_ERR_HIP, "librccl could not be loaded: " + why);
        std::vector<int> devs(s->G);
        std::vector<ncclComm_t> comms(s->G, nullptr);
        for (uint32_t g = 0; g < s->G; ++g) devs[g] = s->sh[g]->device;
        const int e = r->CommInitAll(comms.data(), (int)s->G, devs.data());
        if (e != 0) return fail(CGV_ERR_HIP, std::string("ncclCommInitAll: ") + r->GetErrorString(e));
        for (uint32_t g = 0; g < s->G; ++g) s->sh[g]->comm = comms[g];
    }
    s->exchange = CGV_EXCHANGE_RCCL;
    return CGV_OK;
}

}  // namespace

extern "C" {

// internal (tests): try to load RCCL the way cgv_sharded_create does. Returns 1 when usable, else 0 with the
// reason in msg (truncated to cap bytes). Needs no device.
int cgv_debug_rccl_probe_(char* msg, uint32_t cap) {
    std::string why;
    const Rccl* r = load_rccl(&why);
    if (msg && cap) {
        strncpy(msg, r ? "" : why.c_str(), cap - 1);
        msg[cap - 1] = 0;
    }
    return r ? 1 : 0;
}

int cgv_sharded_create(uint32_t dim, int metric, int dtype, uint32_t n_devices, const int* device_ids, cgv_sharded** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices == 0 || n_devices > 64 || !device_ids) return fail(CGV_ERR_INVALID_ARG, "n_devices must be 1..64 with a device list");
    const int ndev = cgv_device_count();
    if (ndev == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    for (uint32_t g = 0; g < n_devices; ++g)
        if (device_ids[g] < 0 || device_ids[g] >= ndev) return fail(CGV_ERR_INVALID_ARG, "device id out of range");
    cgv_sharded* s = new cgv_sharded();
    s->D = dim;
    s->metric = metric;
    s->dtype = dtype;
    s->G = n_devices;
    for (uint32_t g = 0; g < n_devices; ++g)
        for (uint32_t g2 = 0; g2 < g; ++g2)
            if (device_ids[g] == device_ids[g2]) s->distinct = false;
    int rc = CGV_OK;
    for (uint32_t g = 0; g < n_devices && rc == CGV_OK; ++g) {
        Shard* sh = new Shard();
        sh->device = device_ids[g];
        sh->index = g;
        s->sh.push_back(sh);
        if ((rc = cgv_create(dim, metric, dtype, sh->device, &sh->ix))) break;  // leaves sh->device current
        if ((rc = cgv_set_id_map(sh->ix, CGV_SHARD_CHUNK_ROWS, n_devices, g))) break;
        if (hipStreamCreateWithFlags(&sh->xs, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(CGV_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        if ((rc = cgv_set_stream(sh->ix, sh->xs))) break;  // the shard's searches / ingests order after xs
        sh->w.th = std::thread(worker_main, sh);
    }
    if (rc == CGV_OK && s->distinct && n_devices > 1) {  // direct xGMI copies into the root's gather buffer
        for (uint32_t g = 1; g < n_devices; ++g) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s->sh[g]->device, s->sh[0]->device) == hipSuccess && can) {
                (void)hipSetDevice(s->sh[g]->device);
                const hipError_t e = hipDeviceEnablePeerAccess(s->sh[0]->device, 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled / not supported: copies are staged
            }
        }
    }
    if (rc == CGV_OK) {
        int want = s->distinct ? CGV_EXCHANGE_RCCL : CGV_EXCHANGE_COPY;
        if (const char* e = getenv("CGV_SHARDED_EXCHANGE")) {
            if (!strcmp(e, "copy")) want = CGV_EXCHANGE_COPY;
            if (!strcmp(e, "rccl")) want = CGV_EXCHANGE_RCCL;
        }
        rc = set_exchange_locked(s, want);
        if (rc != CGV_OK && want == CGV_EXCHANGE_RCCL && !getenv("CGV_SHARDED_EXCHANGE")) {
            // RCCL not loadable on this box: the exchange still happens on the device side, by peer copies
            rc = set_exchange_locked(s, CGV_EXCHANGE_COPY);
        }
    }
    if (rc != CGV_OK) {
        const std::string msg = cgv_last_error();
        cgv_sharded_destroy(s);
        return fail(rc, msg);
    }
    *out = s;
    return CGV_OK;
}

int cgv_sharded_destroy(cgv_sharded* s) {
    if (!s) return CGV_OK;
    for (Shard* sh : s->sh) {
        if (sh->w.th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(sh->w.mu);
                sh->w.quit = true;
                sh->w.cv.notify_all();
            }
            sh->w.th.join();
        }
    }
    const Rccl* r = g_rccl.ok() ? &g_rccl : nullptr;
    for (Shard* sh : s->sh) {
        (void)hipSetDevice(sh->device);
        if (sh->xs) (void)hipStreamSynchronize(sh->xs);
        if (sh->comm && r) (void)r->CommDestroy(sh->comm);
        if (sh->ix) (void)cgv_destroy(sh->ix);
        for (Buf* b : {&sh->qdev, &sh->oidx, &sh->osc, &sh->rec, &sh->gathered, &sh->stage}) b->release();
        if (sh->xs) (void)hipStreamDestroy(sh->xs);
    }
    if (!s->sh.empty()) {
        (void)hipSetDevice(s->sh[0]->device);
        s->moidx.release();
        s->mosc.release();
    }
    if (s->pin_q) (void)hipHostFree(s->pin_q);
    for (Shard* sh : s->sh) delete sh;
    delete s;
    return CGV_OK;
}

int cgv_sharded_reserve(cgv_sharded* s, uint64_t total_rows) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    for (uint32_t g = 0; g < s->G; ++g) {
        const int rc = cgv_reserve(s->sh[g]->ix, shard_count(s, total_rows, g));
        if (rc) return rc;
    }
    return CGV_OK;
}

int cgv_sharded_add_f32(cgv_sharded* s, const float* rows_host, uint64_t n) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n == 0) return CGV_OK;
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    // pieces of [n0, n0 + n) per shard, in global (= local) order
    struct Piece {
        const float* src;
        uint64_t rows;
    };
    std::vector<std::vector<Piece>> pieces(s->G);
    const uint64_t n0 = s->n;
    for (uint64_t r = n0; r < n0 + n;) {
        const uint64_t take = std::min<uint64_t>(C - r % C, n0 + n - r);
        pieces[(r / C) % s->G].push_back({rows_host + (r - n0) * s->D, take});
        r += take;
    }
    std::vector<uint64_t> before(s->G);
    for (uint32_t g = 0; g < s->G; ++g) before[g] = cgv_count(s->sh[g]->ix);
    std::vector<std::function<int()>> jobs(s->G);
    const uint32_t D = s->D;
    for (uint32_t g = 0; g < s->G; ++g) {
        if (pieces[g].empty()) continue;
        Shard* sh = s->sh[g];
        const std::vector<Piece>* pl = &pieces[g];
        jobs[g] = [sh, pl, D]() -> int {
            // stage up to 256 MiB of this shard's pieces, then ONE device-side add per flush
            const uint64_t cap_rows = std::max<uint64_t>(C, (256ull << 20) / ((uint64_t)D * 4));
            uint64_t total = 0;
            for (const Piece& p : *pl) total += p.rows;
            int rc = sh->stage.ensure((size_t)std::min<uint64_t>(total, cap_rows) * D * 4);
            if (rc) return rc;
            uint64_t staged = 0;
            auto flush = [&]() -> int {
                if (!staged) return CGV_OK;
                const int r = cgv_add_f32_dev(sh->ix, (const float*)sh->stage.p, staged);  // ordered after xs
                staged = 0;
                return r;
            };
            for (const Piece& p : *pl) {
                uint64_t off = 0;
                while (off < p.rows) {
                    const uint64_t take = std::min<uint64_t>(p.rows - off, cap_rows - staged);
                    SHIP(hipMemcpyAsync((char*)sh->stage.p + (size_t)staged * D * 4, p.src + off * D, (size_t)take * D * 4,
                                        hipMemcpyHostToDevice, sh->xs));
                    staged += take;
                    off += take;
                    if (staged == cap_rows && (rc = flush())) return rc;
                }
            }
            return flush();
        };
    }
    int rc = run_all(s, jobs);
    if (rc != CGV_OK) {  // all shards or none
        const std::string msg = cgv_last_error();
        for (uint32_t g = 0; g < s->G; ++g) (void)cgv_truncate(s->sh[g]->ix, before[g]);
        return fail(rc, msg);
    }
    s->n = n0 + n;
    return CGV_OK;
}

int cgv_sharded_update_row_f32(cgv_sharded* s, uint64_t id, const float* row_host) {
    if (!s || !row_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    uint32_t g;
    uint64_t local;
    locate(s, id, &g, &local);
    return cgv_update_row_f32(s->sh[g]->ix, local, row_host);
}

int cgv_sharded_get_row_f32(cgv_sharded* s, uint64_t id, float* out_host) {
    if (!s || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    uint32_t g;
    uint64_t local;
    locate(s, id, &g, &local);
    return cgv_get_row_f32(s->sh[g]->ix, local, out_host);
}

uint64_t cgv_sharded_count(const cgv_sharded* s) { return s ? s->n : 0; }
uint32_t cgv_sharded_n_shards(const cgv_sharded* s) { return s ? s->G : 0; }
cgv_index* cgv_sharded_shard(cgv_sharded* s, uint32_t i) { return (s && i < s->G) ? s->sh[i]->ix : nullptr; }
int cgv_sharded_exchange(const cgv_sharded* s) { return s ? s->exchange : CGV_EXCHANGE_NONE; }

int cgv_sharded_set_exchange(cgv_sharded* s, int kind) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    return set_exchange_locked(s, kind);
}

int cgv_sharded_search_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, uint32_t k, uint64_t* out_idx_host,
                           float* out_score_host) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || k == 0) return CGV_OK;  // surreal_store.rs:62-64
    if (!queries_host || !out_idx_host || !out_score_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    if (k > CGV_MAX_K) return fail(CGV_ERR_INVALID_ARG, "k exceeds CGV_MAX_K");
    if ((uint64_t)s->G * k > 4096) return fail(CGV_ERR_INVALID_ARG, "n_shards * k exceeds 4096 (merge capacity)");
    std::lock_guard<std::mutex> lk(s->mu);
    const auto t0 = Clock::now();
    const uint32_t G = s->G, D = s->D;
    const uint32_t w = cgv_packed_width(k);
    const size_t qbytes = (size_t)nq * D * 4, rec_bytes = (size_t)nq * w * 4;
    Shard* root = s->sh[0];
    // buffers (only when the batch shape grows)
    if (s->pin_q_bytes < qbytes) {
        if (s->pin_q) (void)hipHostFree(s->pin_q);
        s->pin_q = nullptr;
        s->pin_q_bytes = 0;
        SHIP(hipSetDevice(root->device));
        SHIP(hipHostMalloc((void**)&s->pin_q, qbytes, hipHostMallocPortable));
        s->pin_q_bytes = qbytes;
    }
    for (Shard* sh : s->sh) {
        SHIP(hipSetDevice(sh->device));
        int rc;
        if ((rc = sh->qdev.ensure(qbytes))) return rc;
        if ((rc = sh->oidx.ensure((size_t)nq * k * 8))) return rc;
        if ((rc = sh->osc.ensure((size_t)nq * k * 4))) return rc;
        if (G > 1) {
            if ((rc = sh->rec.ensure(rec_bytes))) return rc;
            if ((s->exchange == CGV_EXCHANGE_RCCL || sh == root) && (rc = sh->gathered.ensure((size_t)G * rec_bytes))) return rc;
        }
    }
    SHIP(hipSetDevice(root->device));
    if (G > 1) {
        int rc;
        if ((rc = s->moidx.ensure((size_t)nq * k * 8))) return rc;
        if ((rc = s->mosc.ensure((size_t)nq * k * 4))) return rc;
    }
    memcpy(s->pin_q, queries_host, qbytes);  // one pass; every device then pulls it over its own link

    const int exchange = s->exchange;
    const Rccl* rccl = exchange == CGV_EXCHANGE_RCCL ? load_rccl() : nullptr;
    const float* pin_q = s->pin_q;
    std::vector<std::function<int()>> jobs(G);
    for (uint32_t g = 0; g < G; ++g) {
        Shard* sh = s->sh[g];
        jobs[g] = [=]() -> int {
            SHIP(hipMemcpyAsync(sh->qdev.p, pin_q, qbytes, hipMemcpyHostToDevice, sh->xs));
            int rc = cgv_search_f32_dev(sh->ix, (const float*)sh->qdev.p, nq, k, (uint64_t*)sh->oidx.p, (float*)sh->osc.p);
            sh->t_search_done = Clock::now();
            if (rc || G == 1) return rc;
            if ((rc = cgv_pack_topk_dev(sh->device, (const uint64_t*)sh->oidx.p, (const float*)sh->osc.p, nq, k,
                                        (uint32_t*)sh->rec.p, sh->xs)))
                return rc;
            if (exchange == CGV_EXCHANGE_RCCL) return CGV_OK;  // the collective is entered below, by ALL shards or none
            char* dst = (char*)root->gathered.p + (size_t)g * rec_bytes;
            if (sh->device == root->device)
                SHIP(hipMemcpyAsync(dst, sh->rec.p, rec_bytes, hipMemcpyDeviceToDevice, sh->xs));
            else
                SHIP(hipMemcpyPeerAsync(dst, root->device, sh->rec.p, sh->device, rec_bytes, sh->xs));
            SHIP(hipStreamSynchronize(sh->xs));
            return CGV_OK;
        };
    }
    int rc = run_all(s, jobs);
    if (rc) return rc;
    if (G > 1 && exchange == CGV_EXCHANGE_RCCL) {
        // A collective that one rank never enters hangs the others: it is only started once every shard's search
        // and pack were enqueued without error (a failed shard returned above, before anyone called into RCCL).
        for (uint32_t g = 0; g < G; ++g) {
            Shard* sh = s->sh[g];
            jobs[g] = [=]() -> int {
                const int e = rccl->AllGather(sh->rec.p, sh->gathered.p, (size_t)nq * w, NCCL_INT32, sh->comm, sh->xs);
                if (e != 0) return fail(CGV_ERR_HIP, std::string("ncclAllGather: ") + rccl->GetErrorString(e));
                SHIP(hipStreamSynchronize(sh->xs));
                return CGV_OK;
            };
        }
        if ((rc = run_all(s, jobs))) return rc;
    }
    auto t1 = s->sh[0]->t_search_done;
    for (Shard* sh : s->sh) t1 = std::max(t1, sh->t_search_done);
    SHIP(hipSetDevice(root->device));
    const uint64_t* ri = (const uint64_t*)root->oidx.p;
    const float* rs = (const float*)root->osc.p;
    if (G > 1) {
        if ((rc = cgv_merge_packed_dev(root->device, (const uint32_t*)root->gathered.p, G, nq, k, (uint64_t*)s->moidx.p,
                                       (float*)s->mosc.p, root->xs)))
            return rc;
        ri = (const uint64_t*)s->moidx.p;
        rs = (const float*)s->mosc.p;
    }
    SHIP(hipMemcpyAsync(out_idx_host, ri, (size_t)nq * k * 8, hipMemcpyDeviceToHost, root->xs));
    SHIP(hipMemcpyAsync(out_score_host, rs, (size_t)nq * k * 4, hipMemcpyDeviceToHost, root->xs));
    SHIP(hipStreamSynchronize(root->xs));
    const auto t2 = Clock::now();
    s->searches++;
    s->queries += nq;
    s->last_search_ms = std::chrono::duration<float, std::milli>(t2 - t0).count();
    s->last_exchange_ms = std::chrono::duration<float, std::milli>(t2 - t1).count();
    return CGV_OK;
}

int cgv_sharded_get_stats(cgv_sharded* s, cgv_sharded_stats* out) {
    if (!s || !out) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memset(out, 0, sizeof(*out));
    out->n_rows = s->n;
    out->n_shards = s->G;
    out->exchange = (uint32_t)s->exchange;
    out->searches = s->searches;
    out->queries = s->queries;
    out->last_search_ms = s->last_search_ms;
    out->last_exchange_ms = s->last_exchange_ms;
    for (Shard* sh : s->sh) {
        cgv_stats st;
        const int rc = cgv_get_stats(sh->ix, &st);
        if (rc) return rc;
        out->device_bytes += st.device_bytes + sh->qdev.bytes + sh->oidx.bytes + sh->osc.bytes + sh->rec.bytes +
                             sh->gathered.bytes + sh->stage.bytes;
        out->fallback_queries += st.fallback_queries;
    }
    return CGV_OK;
}

}  // extern "C"
