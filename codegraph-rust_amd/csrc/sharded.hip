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
#include <deque>
#include <functional>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cgvec.h"

extern "C" int cgv_set_error_(int code, const char* msg);  // abi.hip: the library's thread-local error message
extern "C" int cgv_score_pairs_f32_(cgv_index* h, const float* queries_host, uint32_t nq, int op, const uint32_t* qsel_host,
                                    const uint64_t* ids_host, uint64_t npairs, float* out_host);  // abi.hip

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
    int (*CommAbort)(ncclComm_t) = nullptr;  // optional: unblocks the other ranks when one fails to post
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;  // optional pair: the warm-up collective of a new communicator set (warm_up_comms)
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GetErrorString; }
};
std::mutex g_rccl_mu;
Rccl g_rccl;
bool g_rccl_tried = false;
std::string g_rccl_err;  // why the library is unusable (dlerror() text is consumed by the call that reads it)
std::string g_rccl_forced;  // cgv_debug_rccl_lib_(): the only name to try (tests: a name that cannot be loaded exercises the
                            // fall-back to the copy exchange). The library reads nothing from the environment.

// why: receives the reason when NULL is returned.
const Rccl* load_rccl(std::string* why = nullptr) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl_tried) {
        g_rccl_tried = true;
        // a copy already mapped into the process (e.g. PyTorch's) is found first by its soname
        std::vector<std::string> names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (!g_rccl_forced.empty()) names = {g_rccl_forced};
        for (const std::string& name : names) {
            g_rccl.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.lib) break;
            const char* de = dlerror();  // read ONCE: the call clears the message
            g_rccl_err = de ? de : ("dlopen(" + name + ") failed");
        }
        if (g_rccl.lib) {
            g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(g_rccl.lib, "ncclCommInitAll");
            g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
            g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(g_rccl.lib, "ncclCommAbort");
            g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
            g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.lib, "ncclGroupStart");
            g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.lib, "ncclGroupEnd");
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

// One worker thread per shard with a FIFO of jobs. A job's completion is awaited through its sequence number, so the
// calling thread can post the first half of a batch (queries in, search enqueued), return to its caller, and post /
// await the second half (search done, pack, exchange) later: batches overlap on the devices.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    struct Job {
        uint64_t seq;
        std::function<int()> fn;
    };
    std::deque<Job> q;
    uint64_t next_seq = 1, done_seq = 0;  // jobs complete in order
    std::deque<std::pair<uint64_t, std::pair<int, std::string>>> failed;  // (seq, (rc, message)) of failed jobs not yet collected
    bool quit = false;
    uint64_t post(std::function<int()> fn) {
        std::lock_guard<std::mutex> lk(mu);
        const uint64_t id = next_seq++;
        q.push_back(Job{id, std::move(fn)});
        cv.notify_all();
        return id;
    }
    // wait until job `id` has run; returns its status (message in *err)
    int wait(uint64_t id, std::string* err) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done_seq >= id; });
        for (auto it = failed.begin(); it != failed.end(); ++it)
            if (it->first == id) {
                const int rc = it->second.first;
                if (err) *err = it->second.second;
                failed.erase(it);
                return rc;
            }
        return CGV_OK;
    }
};

struct Shard {
    int device = 0;
    uint32_t index = 0;
    cgv_index* ix = nullptr;
    hipStream_t xs = nullptr;  // H2D of the queries, pack, exchange
    ncclComm_t comm = nullptr;
    Buf stage;                 // ingest staging
    struct SlotBufs {          // one set per batch in flight
        Buf qdev, oidx, osc, rec, gathered;
        uint64_t ticket = 0;   // the shard's own search ticket (cgv_search_begin_f32_dev / cgv_search_packed_begin_f32_dev)
        int rc_a = CGV_OK;     // status of the first half (queries in + search enqueued + records packed + exchange posted)
        std::string err_a;
        bool packed = false;   // the ticket is a packed one (cgv_search_packed_end)
        int xerr = 0;          // this shard's ncclAllGather CALL failed (the communicators were aborted by its job)
        int repacked = 0;      // cgv_search_packed_end replaced provisional records
        hipEvent_t x0 = nullptr, x1 = nullptr;  // root only: records ready -> merged results copied (last_exchange_ms)
        hipEvent_t done = nullptr;  // behind the last operation the first half enqueued for THIS batch on xs: what the second
        bool done_ok = false;       // half waits for (xs is shared by the batches in flight - ADVICE r4)
    } slot[3];
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
    // batches in flight (cgv_sharded_search_begin_f32 / _end): N_SLOTS sets of staging + result buffers
    struct Slot {
        bool busy = false;
        uint32_t gen = 0, nq = 0, k = 0;
        float* pin_q = nullptr;
        size_t pin_q_bytes = 0;
        Buf moidx, mosc;  // merged results, on the root device
        uint32_t* pin_redo = nullptr;  // pinned word the merge kernel raises when a record was PROVISIONAL (cgvec.h)
        bool merged_in_begin = false;  // RCCL exchange: the root's merge + result copy were enqueued by the first half
        int exchange = CGV_EXCHANGE_NONE;  // the exchange the first half used
        uint64_t* out_idx = nullptr;
        float* out_score = nullptr;
        std::vector<uint64_t> job_a;  // per shard: sequence number of the first-half job
        std::chrono::steady_clock::time_point t0;
    } slots[3];
    std::condition_variable slot_cv;
    bool rccl_broken = false;  // a rank failed to post a collective: communicators aborted, copy exchange from now on
    // abort_comms_now(): the failing rank's worker aborts every communicator, once - EXCLUSIVE; every ncclAllGather call is
    // made under a SHARED hold after checking comms_aborted, so no worker can enter a collective on a freed communicator
    // (ADVICE r4). A communicator's FIRST collective may block inside the call on lazy connection setup until every rank has
    // entered; that one is made by warm_up_comms() at creation, from one thread inside a group, so the per-batch calls only
    // enqueue and a shared hold does not wait for a peer. Should one block all the same, abort_comms_now() stops waiting for the
    // exclusive hold after ABORT_LOCK_MS and aborts anyway: the abort is what unblocks that call.
    std::shared_timed_mutex abort_mu;
    std::atomic<bool> comms_aborted{false};
    // a handle over ONE shard normally skips pack / exchange / merge; cgv_sharded_force_exchange() runs them anyway - a
    // one-rank ncclAllGather: the only way to execute the RCCL branch on a single-GPU box (tests)
    bool force_xch = false;
    uint64_t searches = 0, queries = 0, redo_batches = 0;
    float last_search_ms = 0.0f, last_exchange_ms = 0.0f;
};

namespace {

constexpr uint64_t C = CGV_SHARD_CHUNK_ROWS;

constexpr int N_SLOTS = 3;
constexpr int ABORT_LOCK_MS = 500;

void worker_main(Shard* s) {
    (void)hipSetDevice(s->device);
    Worker& w = s->w;
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.cv.wait(lk, [&] { return !w.q.empty() || w.quit; });
        if (w.q.empty()) return;  // quit, nothing pending
        Worker::Job job = std::move(w.q.front());
        w.q.pop_front();
        lk.unlock();
        const int rc = job.fn();
        const std::string err = rc ? cgv_last_error() : "";
        lk.lock();
        if (rc) w.failed.push_back({job.seq, {rc, err}});
        w.done_seq = job.seq;
        w.cv.notify_all();
    }
}

// Post one job per shard (jobs[i] may be empty = nothing to do), wait for all; first failure wins.
int run_all(cgv_sharded* s, std::vector<std::function<int()>>& jobs) {
    std::vector<uint64_t> ids(s->G, 0);
    for (uint32_t g = 0; g < s->G; ++g)
        if (jobs[g]) ids[g] = s->sh[g]->w.post(std::move(jobs[g]));
    int rc = CGV_OK;
    std::string err;
    for (uint32_t g = 0; g < s->G; ++g) {
        if (!ids[g]) continue;
        std::string e;
        const int r = s->sh[g]->w.wait(ids[g], &e);
        if (r && !rc) {
            rc = r;
            err = "shard " + std::to_string(g) + " (device " + std::to_string(s->sh[g]->device) + "): " + e;
        }
    }
    return rc ? fail(rc, err) : CGV_OK;
}

// The calling thread's current HIP device is restored when a cgv_sharded_* entry point returns.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() {
        if (hipGetDevice(&dev) != hipSuccess) {
            (void)hipGetLastError();
            dev = -1;
        }
    }
    ~DeviceGuard() {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

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

// Writers and readers of the shards (add, update_row, get_row, reserve, score_ids) run the shard's single-device API on the
// CALLING thread while holding s->mu. A batch begun with cgv_sharded_search_begin_f32 holds a search context of every shard on
// its WORKER thread until cgv_sharded_search_end - which needs s->mu: the shard call would wait for that context forever and
// nobody could release it. Same answer as the single-device handle: CGV_ERR_BUSY while a batch is in flight.
int busy_if_in_flight(const cgv_sharded* s, const char* what) {
    for (const auto& sl : s->slots)
        if (sl.busy)
            return fail(CGV_ERR_BUSY, std::string(what) + ": a batch begun with cgv_sharded_search_begin_f32 is in flight - call "
                                      "cgv_sharded_search_end first");
    return CGV_OK;
}

// The first collective of a communicator set, made here by ONE thread inside a group: lazy connection setup (which can block
// inside the call until every rank has entered) happens now, not under a worker's hold of abort_mu.
int warm_up_comms(cgv_sharded* s, const Rccl* r) {
    if (!r->GroupStart || !r->GroupEnd) return CGV_OK;
    std::vector<void*> src(s->G, nullptr), dst(s->G, nullptr);
    int rc = CGV_OK, e = 0;
    for (uint32_t g = 0; g < s->G && rc == CGV_OK; ++g) {
        if (hipSetDevice(s->sh[g]->device) != hipSuccess || hipMalloc(&src[g], 64) != hipSuccess ||
            hipMalloc(&dst[g], (size_t)64 * s->G) != hipSuccess)
            rc = fail(CGV_ERR_HIP, "warm-up buffers of the RCCL exchange");
    }
    if (rc == CGV_OK) {
        (void)r->GroupStart();
        for (uint32_t g = 0; g < s->G && e == 0; ++g) e = r->AllGather(src[g], dst[g], 16, NCCL_INT32, s->sh[g]->comm, s->sh[g]->xs);
        const int ge = r->GroupEnd();
        if (e == 0) e = ge;
        if (e != 0) rc = fail(CGV_ERR_HIP, std::string("ncclAllGather (warm-up): ") + r->GetErrorString(e));
    }
    for (uint32_t g = 0; g < s->G; ++g) {
        if (hipSetDevice(s->sh[g]->device) != hipSuccess) continue;
        if (rc == CGV_OK && hipStreamSynchronize(s->sh[g]->xs) != hipSuccess) rc = fail(CGV_ERR_HIP, "warm-up collective did not complete");
        if (src[g]) (void)hipFree(src[g]);
        if (dst[g]) (void)hipFree(dst[g]);
    }
    return rc;
}

int set_exchange_locked(cgv_sharded* s, int kind) {
    if (s->G <= 1 && !s->force_xch) {
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
        if (int rc = warm_up_comms(s, r)) {
            for (uint32_t g = 0; g < s->G; ++g) {
                (void)r->CommDestroy(s->sh[g]->comm);
                s->sh[g]->comm = nullptr;
            }
            return rc;
        }
    }
    s->exchange = CGV_EXCHANGE_RCCL;
    return CGV_OK;
}

}  // namespace

extern "C" {

// internal (tests): the only library name load_rccl() may try; must be called before the first use of RCCL in the process.
int cgv_debug_rccl_lib_(const char* name) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_tried) return 0;
    g_rccl_forced = name ? name : "";
    return 1;
}

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
    DeviceGuard guard;
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
        const int want = s->distinct ? CGV_EXCHANGE_RCCL : CGV_EXCHANGE_COPY;  // (cgv_sharded_set_exchange changes it)
        rc = set_exchange_locked(s, want);
        if (rc != CGV_OK && want == CGV_EXCHANGE_RCCL) {
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
    DeviceGuard guard;
    for (Shard* sh : s->sh) {
        if (sh->w.th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(sh->w.mu);
                sh->w.quit = true;  // (pending jobs still run: the worker leaves once its queue is empty)
                sh->w.cv.notify_all();
            }
            sh->w.th.join();
        }
    }
    const Rccl* r = g_rccl.ok() ? &g_rccl : nullptr;
    for (Shard* sh : s->sh) {
        (void)hipSetDevice(sh->device);
        if (sh->xs) (void)hipStreamSynchronize(sh->xs);
        if (sh->comm && r && !s->comms_aborted.load()) (void)r->CommDestroy(sh->comm);
        if (sh->ix) (void)cgv_destroy(sh->ix);
        sh->stage.release();
        for (auto& sb : sh->slot) {
            for (Buf* b : {&sb.qdev, &sb.oidx, &sb.osc, &sb.rec, &sb.gathered}) b->release();
            if (sb.x0) (void)hipEventDestroy(sb.x0);
            if (sb.x1) (void)hipEventDestroy(sb.x1);
            if (sb.done) (void)hipEventDestroy(sb.done);
        }
        if (sh->xs) (void)hipStreamDestroy(sh->xs);
    }
    if (!s->sh.empty()) {
        (void)hipSetDevice(s->sh[0]->device);
        for (auto& sl : s->slots) {
            sl.moidx.release();
            sl.mosc.release();
        }
    }
    for (auto& sl : s->slots) {
        if (sl.pin_q) (void)hipHostFree(sl.pin_q);
        if (sl.pin_redo) (void)hipHostFree(sl.pin_redo);
    }
    for (Shard* sh : s->sh) delete sh;
    delete s;
    return CGV_OK;
}

int cgv_sharded_reserve(cgv_sharded* s, uint64_t total_rows) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    if (int brc = busy_if_in_flight(s, "cgv_sharded_reserve")) return brc;
    for (uint32_t g = 0; g < s->G; ++g) {
        const int rc = cgv_reserve(s->sh[g]->ix, shard_count(s, total_rows, g));
        if (rc) return rc;
    }
    return CGV_OK;
}

int cgv_sharded_add_f32(cgv_sharded* s, const float* rows_host, uint64_t n) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    DeviceGuard guard;
    if (n == 0) return CGV_OK;
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    if (int brc = busy_if_in_flight(s, "cgv_sharded_add_f32")) return brc;
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
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    if (int brc = busy_if_in_flight(s, "cgv_sharded_update_row_f32")) return brc;
    if (id >= s->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    uint32_t g;
    uint64_t local;
    locate(s, id, &g, &local);
    return cgv_update_row_f32(s->sh[g]->ix, local, row_host);
}

int cgv_sharded_get_row_f32(cgv_sharded* s, uint64_t id, float* out_host) {
    if (!s || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    if (int brc = busy_if_in_flight(s, "cgv_sharded_get_row_f32")) return brc;
    if (id >= s->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    uint32_t g;
    uint64_t local;
    locate(s, id, &g, &local);
    return cgv_get_row_f32(s->sh[g]->ix, local, out_host);
}

// cgv_score_ids_f32 over the shards: every (query, id) pair is scored on the shard that owns the row (global id ->
// shard, local row: locate()), all shards in parallel, one device launch each. Each shard receives ONLY its own pairs as a
// compact list (query index, local row) and returns one score per pair, scattered back through the pair's position: host
// memory and device work are O(nq * m) in total, not per shard.
int cgv_sharded_score_ids_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, int op, const uint64_t* ids_host,
                              uint32_t m, float* out_host) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || m == 0) return CGV_OK;
    if (!queries_host || !ids_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    if (int brc = busy_if_in_flight(s, "cgv_sharded_score_ids_f32")) return brc;
    const size_t pairs = (size_t)nq * m;
    struct Part {
        std::vector<uint32_t> qsel;   // query of the pair
        std::vector<uint64_t> local;  // local row on the shard
        std::vector<size_t> pos;      // position of the pair in the caller's [nq][m] arrays
        std::vector<float> score;
    };
    std::vector<Part> part(s->G);
    for (size_t i = 0; i < pairs; ++i) {
        out_host[i] = 0.0f;  // UINT64_MAX or beyond the index: scores 0.0 (a missing embedding, search.rs:207-217)
        const uint64_t id = ids_host[i];
        if (id >= s->n) continue;
        uint32_t g;
        uint64_t l;
        locate(s, id, &g, &l);
        part[g].qsel.push_back((uint32_t)(i / m));
        part[g].local.push_back(l);
        part[g].pos.push_back(i);
    }
    std::vector<std::function<int()>> jobs(s->G);
    for (uint32_t g = 0; g < s->G; ++g) {
        Part* p = &part[g];
        if (p->pos.empty()) continue;
        p->score.assign(p->pos.size(), 0.0f);
        Shard* sh = s->sh[g];
        jobs[g] = [=]() -> int {
            return cgv_score_pairs_f32_(sh->ix, queries_host, nq, op, p->qsel.data(), p->local.data(), (uint64_t)p->pos.size(),
                                        p->score.data());
        };
    }
    const int rc = run_all(s, jobs);
    if (rc) return rc;
    for (const Part& p : part)
        for (size_t j = 0; j < p.pos.size(); ++j) out_host[p.pos[j]] = p.score[j];
    return CGV_OK;
}

uint64_t cgv_sharded_count(const cgv_sharded* s) { return s ? s->n : 0; }
uint32_t cgv_sharded_n_shards(const cgv_sharded* s) { return s ? s->G : 0; }
cgv_index* cgv_sharded_shard(cgv_sharded* s, uint32_t i) { return (s && i < s->G) ? s->sh[i]->ix : nullptr; }
int cgv_sharded_exchange(const cgv_sharded* s) { return s ? s->exchange : CGV_EXCHANGE_NONE; }

int cgv_sharded_set_exchange(cgv_sharded* s, int kind) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    for (const auto& sl : s->slots)  // a batch in flight sized its buffers for the current exchange
        if (sl.busy) return fail(CGV_ERR_BUSY, "a search is in flight: call cgv_sharded_search_end first");
    if (kind == CGV_EXCHANGE_RCCL && s->rccl_broken)
        return fail(CGV_ERR_HIP, "the handle's RCCL communicators were aborted after a failed collective");
    return set_exchange_locked(s, kind);
}

int cgv_sharded_force_exchange(cgv_sharded* s, int enabled) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(s->mu);
    for (const auto& sl : s->slots)
        if (sl.busy) return fail(CGV_ERR_BUSY, "a search is in flight: call cgv_sharded_search_end first");
    s->force_xch = enabled != 0;
    if (s->G > 1) return CGV_OK;  // several shards always exchange
    if (!s->force_xch) return set_exchange_locked(s, CGV_EXCHANGE_NONE);
    int rc = set_exchange_locked(s, s->rccl_broken ? CGV_EXCHANGE_COPY : CGV_EXCHANGE_RCCL);
    if (rc != CGV_OK) rc = set_exchange_locked(s, CGV_EXCHANGE_COPY);  // RCCL not loadable: the one-shard exchange by a copy
    return rc;
}

// ---- search: batches in flight (cgv_sharded_search_begin_f32 / _end, below) ----------------------------------------
namespace {

int slot_of_ticket(cgv_sharded* s, uint64_t ticket, cgv_sharded::Slot** out) {
    const uint64_t si = ticket & 0xff;
    if (si == 0 || si > (uint64_t)N_SLOTS) return fail(CGV_ERR_INVALID_ARG, "bad ticket");
    cgv_sharded::Slot* sl = &s->slots[si - 1];
    if (!sl->busy || sl->gen != (uint32_t)(ticket >> 8)) return fail(CGV_ERR_INVALID_ARG, "stale ticket");
    *out = sl;
    return CGV_OK;
}

// every shard's exchange stream idle (error paths: nothing of this batch may still be reading the staging buffers)
void drain_streams(cgv_sharded* s) {
    std::vector<std::function<int()>> jobs(s->G);
    for (uint32_t g = 0; g < s->G; ++g) {
        Shard* sh = s->sh[g];
        jobs[g] = [sh]() -> int {
            (void)hipStreamSynchronize(sh->xs);
            return CGV_OK;
        };
    }
    (void)run_all(s, jobs);
}

// A rank failed to post its part of a collective: the other ranks have posted theirs and sit in hipStreamSynchronize on a
// kernel that can never complete - on their WORKER threads, inside the jobs run_all() is waiting for. So the abort has to
// come from the failing rank's own job (abort_comms_now, once, from whichever worker gets there first): ncclCommAbort is
// made to be called from another thread and makes the stuck collectives return. abort_rccl() then runs on the caller's
// thread after the join: the communicators are gone and the handle continues with the copy exchange. Without
// ncclCommAbort in the library the communicators are destroyed instead.
void abort_comms_now(cgv_sharded* s, const Rccl* r) {
    // no ncclAllGather call should be in progress while the communicators go - but never wait for one forever: a rank stuck INSIDE
    // the call (holding abort_mu shared) can only be released by the abort itself
    std::unique_lock<std::shared_timed_mutex> lk(s->abort_mu, std::chrono::milliseconds(ABORT_LOCK_MS));
    if (s->comms_aborted.exchange(true)) return;
    for (Shard* sh : s->sh) {
        if (!sh->comm || !r) continue;
        if (r->CommAbort) (void)r->CommAbort(sh->comm);
        else (void)r->CommDestroy(sh->comm);
    }
}

// ncclAllGather of one shard's records, entered only while the communicators are alive. Returns 0, an RCCL error code, or -1
// when the communicators were aborted (by a failing rank of this or an earlier batch): the call is skipped.
int guarded_all_gather(cgv_sharded* s, const Rccl* r, Shard* sh, const void* src, void* dst, size_t count) {
    std::shared_lock<std::shared_timed_mutex> lk(s->abort_mu);
    if (s->comms_aborted.load() || !sh->comm) return -1;
    return r->AllGather(src, dst, count, NCCL_INT32, sh->comm, sh->xs);
}

void abort_rccl(cgv_sharded* s, const Rccl* r) {
    abort_comms_now(s, r);  // (no-op when the failing worker already did it)
    s->rccl_broken = true;
    for (Shard* sh : s->sh) sh->comm = nullptr;
    s->exchange = CGV_EXCHANGE_COPY;
}

}  // namespace

// Root side of the exchange: merge the G gathered record blocks (raising the slot's redo word on a provisional record) and
// copy the merged results to the caller's host arrays, all on the root's stream, no wait. The root device is current.
static int enqueue_root_merge(cgv_sharded* s, cgv_sharded::Slot& sl, int si) {
    Shard* root = s->sh[0];
    int rc = cgv_merge_packed_flag_dev(root->device, (const uint32_t*)root->slot[si].gathered.p, s->G, sl.nq, sl.k,
                                       (uint64_t*)sl.moidx.p, (float*)sl.mosc.p, sl.pin_redo, root->xs);
    if (rc) return rc;
    SHIP(hipMemcpyAsync(sl.out_idx, sl.moidx.p, (size_t)sl.nq * sl.k * 8, hipMemcpyDeviceToHost, root->xs));
    SHIP(hipMemcpyAsync(sl.out_score, sl.mosc.p, (size_t)sl.nq * sl.k * 4, hipMemcpyDeviceToHost, root->xs));
    return CGV_OK;
}

// Round 4: NO host join between a shard's search and the exchange. The first half (begin) enqueues, per shard and on its
// worker thread: queries in -> cgv_search_packed_begin_f32_dev (search + records packed on the search's own stream; the
// shard's exchange stream waits for them by an event) -> the exchange, straight behind:
//   RCCL  ncclAllGather posted by EVERY shard right here (also by one whose search failed to enqueue: a collective some rank
//         skips never completes on the others); the root also enqueues merge + result copy behind its all-gather;
//   COPY  the records go into the root's gather buffer; the root merges in the second half, once every copy has landed.
// The second half (end) is ONE join: every shard waits for its stream and ends its search (cgv_search_packed_end: status,
// exact scan + re-pack of provisional records). The merge raised the slot's pinned redo word if any record was provisional
// (a query whose top-k a shard could not prove on the device): only then is the exchange repeated with the final records.
int cgv_sharded_search_begin_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, uint32_t k,
                                 uint64_t* out_idx_host, float* out_score_host, uint64_t* ticket) {
    if (!ticket) return fail(CGV_ERR_INVALID_ARG, "ticket is NULL");
    *ticket = 0;
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (nq == 0 || k == 0) return CGV_OK;  // surreal_store.rs:62-64; ticket 0 = nothing to wait for
    if (!queries_host || !out_idx_host || !out_score_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    if (k > CGV_MAX_K) return fail(CGV_ERR_INVALID_ARG, "k exceeds CGV_MAX_K");
    DeviceGuard guard;
    std::unique_lock<std::mutex> lk(s->mu);
    int si = -1;
    for (int i = 0; i < N_SLOTS; ++i)
        if (!s->slots[i].busy) {
            si = i;
            break;
        }
    if (si < 0)
        return fail(CGV_ERR_BUSY, "all " + std::to_string(N_SLOTS) + " batches in flight: call cgv_sharded_search_end first");
    cgv_sharded::Slot& sl = s->slots[si];
    const uint32_t G = s->G, D = s->D;
    const uint32_t w = cgv_packed_width(k);
    const size_t qbytes = (size_t)nq * D * 4, rec_bytes = (size_t)nq * w * 4;
    Shard* root = s->sh[0];
    const bool xch = G > 1 || s->force_xch;
    // (a rank of an earlier batch failed to post its collective and aborted the communicators, and that batch has not been
    // ended yet - abort_rccl() switches the handle over there: batches begun in between already use the copy exchange)
    const int exchange = (s->exchange == CGV_EXCHANGE_RCCL && s->comms_aborted.load()) ? CGV_EXCHANGE_COPY : s->exchange;
    if (sl.pin_q_bytes < qbytes) {  // (the slot is idle: nothing reads its staging)
        if (sl.pin_q) (void)hipHostFree(sl.pin_q);
        sl.pin_q = nullptr;
        sl.pin_q_bytes = 0;
        SHIP(hipSetDevice(root->device));
        SHIP(hipHostMalloc((void**)&sl.pin_q, qbytes, hipHostMallocPortable));
        sl.pin_q_bytes = qbytes;
    }
    if (!sl.pin_redo) {
        SHIP(hipSetDevice(root->device));
        SHIP(hipHostMalloc((void**)&sl.pin_redo, 64, hipHostMallocPortable | hipHostMallocMapped));
    }
    for (Shard* sh : s->sh) {
        SHIP(hipSetDevice(sh->device));
        Shard::SlotBufs& b = sh->slot[si];
        int rc;
        if ((rc = b.qdev.ensure(qbytes))) return rc;
        if (!xch) {
            if ((rc = b.oidx.ensure((size_t)nq * k * 8))) return rc;
            if ((rc = b.osc.ensure((size_t)nq * k * 4))) return rc;
        } else {
            if ((rc = b.rec.ensure(rec_bytes))) return rc;
            if ((exchange == CGV_EXCHANGE_RCCL || sh == root) && (rc = b.gathered.ensure((size_t)G * rec_bytes))) return rc;
        }
        if (sh == root && !b.x0) SHIP(hipEventCreate(&b.x0));  // (independently: a failed second create must not leave the
        if (sh == root && !b.x1) SHIP(hipEventCreate(&b.x1));  //  first one looking like "both exist" - ADVICE r4)
        if (!b.done) SHIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    }
    if (xch) {
        SHIP(hipSetDevice(root->device));
        int rc;
        if ((rc = sl.moidx.ensure((size_t)nq * k * 8))) return rc;
        if ((rc = sl.mosc.ensure((size_t)nq * k * 4))) return rc;
    }
    const Rccl* rccl = nullptr;
    if (xch && exchange == CGV_EXCHANGE_RCCL) {
        std::string why;
        if (!(rccl = load_rccl(&why))) return fail(CGV_ERR_HIP, "librccl could not be loaded: " + why);
    }
    memcpy(sl.pin_q, queries_host, qbytes);  // one pass; every device then pulls it over its own link
    *sl.pin_redo = 0u;                       // (the slot is idle: no kernel writes it)
    sl.busy = true;
    sl.gen++;
    sl.nq = nq;
    sl.k = k;
    sl.out_idx = out_idx_host;
    sl.out_score = out_score_host;
    sl.t0 = Clock::now();
    sl.merged_in_begin = xch && exchange == CGV_EXCHANGE_RCCL;
    sl.exchange = exchange;
    sl.job_a.assign(G, 0);
    const float* pin_q = sl.pin_q;
    cgv_sharded::Slot* slp = &sl;
    for (uint32_t g = 0; g < G; ++g) {
        Shard* sh = s->sh[g];
        sl.job_a[g] = sh->w.post([=]() -> int {
            Shard::SlotBufs& b = sh->slot[si];
            b.ticket = 0;
            b.rc_a = CGV_OK;
            b.packed = xch;
            b.xerr = 0;
            b.repacked = 0;
            auto search_half = [&]() -> int {
                SHIP(hipMemcpyAsync(b.qdev.p, pin_q, qbytes, hipMemcpyHostToDevice, sh->xs));
                if (!xch)
                    return cgv_search_begin_f32_dev(sh->ix, (const float*)b.qdev.p, nq, k, (uint64_t*)b.oidx.p, (float*)b.osc.p,
                                                    &b.ticket);
                return cgv_search_packed_begin_f32_dev(sh->ix, (const float*)b.qdev.p, nq, k, (uint32_t*)b.rec.p, (void*)sh->xs,
                                                       &b.ticket);
            };
            b.rc_a = search_half();
            if (b.rc_a) b.err_a = cgv_last_error();
            // whatever follows, the second half waits for THIS batch's last operation on xs, not for the stream
            struct DoneMark {
                Shard* sh;
                Shard::SlotBufs& b;
                ~DoneMark() { b.done_ok = hipEventRecord(b.done, sh->xs) == hipSuccess; }
            } mark{sh, b};
            if (!xch) return CGV_OK;  // the status travels with the slot: the second half reports it
            if (sh == root) (void)hipEventRecord(b.x0, sh->xs);   // the root's records are ready
            if (exchange == CGV_EXCHANGE_RCCL) {
                // entered by every shard, whatever its own search returned (the buffers exist; the batch's status discards
                // the result): a collective that one rank skips never completes on the others
                const int e = guarded_all_gather(s, rccl, sh, b.rec.p, b.gathered.p, (size_t)nq * w);
                if (e != 0) {
                    b.xerr = 1;
                    if (b.rc_a == CGV_OK) {
                        b.rc_a = CGV_ERR_HIP;
                        b.err_a = e < 0 ? std::string("ncclAllGather skipped: the communicators were aborted by a failing rank")
                                        : std::string("ncclAllGather: ") + rccl->GetErrorString(e);
                    }
                    abort_comms_now(s, rccl);  // the other ranks posted theirs and will wait on their streams: unblock them
                    return CGV_OK;
                }
                if (sh == root && b.rc_a == CGV_OK) {
                    const int mrc = enqueue_root_merge(s, *slp, si);
                    if (mrc) {
                        b.rc_a = mrc;
                        b.err_a = cgv_last_error();
                    }
                }
                if (sh == root) (void)hipEventRecord(b.x1, sh->xs);  // (always: the second half reads the pair)
            } else if (b.rc_a == CGV_OK) {
                char* dst = (char*)root->slot[si].gathered.p + (size_t)sh->index * rec_bytes;
                const hipError_t he = sh->device == root->device
                                          ? hipMemcpyAsync(dst, b.rec.p, rec_bytes, hipMemcpyDeviceToDevice, sh->xs)
                                          : hipMemcpyPeerAsync(dst, root->device, b.rec.p, sh->device, rec_bytes, sh->xs);
                if (he != hipSuccess) {
                    b.rc_a = CGV_ERR_HIP;
                    b.err_a = std::string("exchange copy: ") + hipGetErrorString(he);
                }
            }
            return CGV_OK;
        });
    }
    *ticket = ((uint64_t)sl.gen << 8) | (uint64_t)(si + 1);
    return CGV_OK;
}

int cgv_sharded_search_end(cgv_sharded* s, uint64_t ticket) {
    if (!s) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (ticket == 0) return CGV_OK;
    DeviceGuard guard;
    std::unique_lock<std::mutex> lk(s->mu);
    cgv_sharded::Slot* slp = nullptr;
    if (int rc = slot_of_ticket(s, ticket, &slp)) return rc;
    cgv_sharded::Slot& sl = *slp;
    const int si = (int)(slp - s->slots);
    const uint32_t G = s->G, nq = sl.nq, k = sl.k;
    const uint32_t w = cgv_packed_width(k);
    const size_t rec_bytes = (size_t)nq * w * 4;
    Shard* root = s->sh[0];
    const bool xch = G > 1 || s->force_xch;
    const bool rccl_mode = sl.merged_in_begin;  // the exchange the first half used
    std::string why;
    const Rccl* rccl = rccl_mode ? load_rccl(&why) : nullptr;
    auto finish = [&](int code) {
        sl.busy = false;
        s->slot_cv.notify_all();
        return code;
    };
    // ---- the ONE join: every shard's stream idle, every search ended ----
    std::vector<std::function<int()>> jobs(G);
    for (uint32_t g = 0; g < G; ++g) {
        Shard* sh = s->sh[g];
        jobs[g] = [=]() -> int {
            Shard::SlotBufs& b = sh->slot[si];
            int rc = b.rc_a;
            std::string err = b.err_a;
            // this batch's exchange (and, RCCL root, merge + result copy) done - its own event, not the stream the batches
            // in flight share
            const hipError_t se = b.done_ok ? hipEventSynchronize(b.done) : hipStreamSynchronize(sh->xs);
            if (se != hipSuccess && rc == CGV_OK) {
                rc = CGV_ERR_HIP;
                err = std::string("waiting for the batch: ") + hipGetErrorString(se);
            }
            if (b.ticket) {  // ends the search whatever happened around it (the context must be released)
                const int erc = b.packed ? cgv_search_packed_end(sh->ix, b.ticket, &b.repacked) : cgv_search_end(sh->ix, b.ticket);
                if (erc && rc == CGV_OK) {
                    rc = erc;
                    err = cgv_last_error();
                }
                b.ticket = 0;
            }
            return rc ? fail(rc, err) : CGV_OK;
        };
    }
    int rc = run_all(s, jobs);
    bool any_xerr = false;
    for (Shard* sh : s->sh) any_xerr = any_xerr || sh->slot[si].xerr;
    if (any_xerr) {
        const std::string msg = rc ? cgv_last_error() : "a rank failed to post its ncclAllGather";
        abort_rccl(s, rccl);
        drain_streams(s);
        rc = fail(rc ? rc : CGV_ERR_HIP, msg + " (RCCL communicators aborted; the handle continues with the copy exchange)");
    }
    if (rc) return finish(rc);
    // This batch posted every all-gather, but a rank of ANOTHER batch in flight then failed and aborted the communicators: the
    // stream waits above returned because of the abort, not because the gather completed, so what the root merged may be
    // incomplete. The shards' records are intact (the searches ended above): exchange them again by copies.
    bool rerun_by_copy = false;
    if (rccl_mode && s->comms_aborted.load()) {
        abort_rccl(s, rccl);
        drain_streams(s);
        rerun_by_copy = true;
    }
    hipError_t he = hipSetDevice(root->device);
    if (he != hipSuccess) return finish(fail(CGV_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he)));
    if (!xch) {  // one shard, no exchange: its results go back as they are
        he = hipMemcpyAsync(sl.out_idx, root->slot[si].oidx.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, root->xs);
        if (he == hipSuccess) he = hipMemcpyAsync(sl.out_score, root->slot[si].osc.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, root->xs);
        const hipError_t se = hipStreamSynchronize(root->xs);
        if (he == hipSuccess) he = se;
        if (he != hipSuccess) return finish(fail(CGV_ERR_HIP, std::string("sharded search, result copy: ") + hipGetErrorString(he)));
    } else {
        auto root_round = [&]() -> int {  // merge + result copy on the root, waited for
            int r = enqueue_root_merge(s, sl, si);
            (void)hipEventRecord(root->slot[si].x1, root->xs);
            const hipError_t se = hipStreamSynchronize(root->xs);  // also on the error path: nothing stays in flight
            if (r == CGV_OK && se != hipSuccess) r = fail(CGV_ERR_HIP, std::string("sharded search, merge: ") + hipGetErrorString(se));
            return r;
        };
        if (!rccl_mode && (rc = root_round())) return finish(rc);  // COPY: every shard's records have landed (the join above)
        if (*sl.pin_redo != 0u || rerun_by_copy) {
            // a shard could not prove a query on the device: its search has now run the exact scan and re-packed (the end
            // jobs above) - the exchange is repeated once with the final records (also: the rerun after a foreign abort)
            const bool was_redo = *sl.pin_redo != 0u;
            *sl.pin_redo = 0u;
            const bool by_rccl = rccl_mode && !rerun_by_copy;
            std::vector<std::function<int()>> again(G);
            std::vector<int> xerr2(G, 0);
            for (uint32_t g = 0; g < G; ++g) {
                Shard* sh = s->sh[g];
                int* xe = &xerr2[g];
                again[g] = [=]() -> int {
                    Shard::SlotBufs& b = sh->slot[si];
                    if (by_rccl) {
                        const int e = guarded_all_gather(s, rccl, sh, b.rec.p, b.gathered.p, (size_t)nq * w);
                        if (e != 0) {
                            *xe = 1;
                            abort_comms_now(s, rccl);
                            return fail(CGV_ERR_HIP, e < 0 ? std::string("ncclAllGather skipped: communicators aborted")
                                                           : std::string("ncclAllGather: ") + rccl->GetErrorString(e));
                        }
                    } else {
                        char* dst = (char*)root->slot[si].gathered.p + (size_t)sh->index * rec_bytes;
                        const hipError_t ce = sh->device == root->device
                                                  ? hipMemcpyAsync(dst, b.rec.p, rec_bytes, hipMemcpyDeviceToDevice, sh->xs)
                                                  : hipMemcpyPeerAsync(dst, root->device, b.rec.p, sh->device, rec_bytes, sh->xs);
                        if (ce != hipSuccess) return fail(CGV_ERR_HIP, std::string("exchange copy: ") + hipGetErrorString(ce));
                    }
                    const hipError_t se = hipStreamSynchronize(sh->xs);
                    return se == hipSuccess ? CGV_OK : fail(CGV_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(se));
                };
            }
            rc = run_all(s, again);
            bool x2 = false;
            for (int e : xerr2) x2 = x2 || e;
            if (x2) {
                const std::string msg = cgv_last_error();
                abort_rccl(s, rccl);
                drain_streams(s);
                rc = fail(CGV_ERR_HIP, msg + " (RCCL communicators aborted; the handle continues with the copy exchange)");
            }
            if (rc) return finish(rc);
            if (hipSetDevice(root->device) != hipSuccess) return finish(fail(CGV_ERR_HIP, "hipSetDevice(root)"));
            if ((rc = root_round())) return finish(rc);
            if (*sl.pin_redo != 0u) return finish(fail(CGV_ERR_INTERNAL, "records still provisional after the searches ended"));
            if (was_redo) s->redo_batches++;
        }
    }
    const auto t2 = Clock::now();
    s->searches++;
    s->queries += nq;
    s->last_search_ms = std::chrono::duration<float, std::milli>(t2 - sl.t0).count();
    if (xch) {  // the root's records ready -> merged results copied to the host (events on the root's stream)
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, root->slot[si].x0, root->slot[si].x1) == hipSuccess) s->last_exchange_ms = ms;
        else (void)hipGetLastError();
    } else {
        s->last_exchange_ms = 0.0f;
    }
    return finish(CGV_OK);
}

uint32_t cgv_sharded_max_batches_in_flight(const cgv_sharded* s) { return s ? (uint32_t)N_SLOTS : 0u; }

int cgv_sharded_search_f32(cgv_sharded* s, const float* queries_host, uint32_t nq, uint32_t k, uint64_t* out_idx_host,
                           float* out_score_host) {
    uint64_t t = 0;
    int rc = cgv_sharded_search_begin_f32(s, queries_host, nq, k, out_idx_host, out_score_host, &t);
    if (rc) return rc;
    return cgv_sharded_search_end(s, t);
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
        out->device_bytes += st.device_bytes + sh->stage.bytes;
        for (const auto& sb : sh->slot)
            out->device_bytes += sb.qdev.bytes + sb.oidx.bytes + sb.osc.bytes + sb.rec.bytes + sb.gathered.bytes;
        out->fallback_queries += st.fallback_queries;
    }
    return CGV_OK;
}

}  // extern "C"
