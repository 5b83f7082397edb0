// coalesce.h — group commit of concurrent SMALL searches on one index (host side only; no HIP in this file).
//
// The reference's trait-level call is ONE query: VectorStore::search_similar(&self, &[f32], limit) (traits.rs:14;
// surreal_store.rs:61-85; caller search.rs:114-117), and its only multi-query caller issues B independent concurrent
// single-query searches (search.rs:358-361, try_join_all). Through that unchanged surface every call streams the whole corpus
// for ONE query column (C2's corpus: 1.5 GB, 382 us, 2.6 k q/s per caller) while the device would serve 64 columns for the
// price of one. So callers of cgv_search_f32 with a few queries each are merged here into one device batch:
//
//   * a caller enqueues its request; whoever finds a free LEADER slot takes the requests at the head of the queue (FIFO, up to
//     max_batch_q queries, one k class, what fits the pinned staging area), runs them as ONE search with k = the largest k of
//     the batch and scatters every caller's first k results to its own buffers; the others sleep on their own condition
//     variable until their request is done (no thundering herd: a wake-up goes to exactly the thread it is for);
//   * while `max_leaders` batches are on the device, arrivals queue up behind them - that wait IS the batching window: the
//     busier the index, the larger the batches. A lone caller finds a free slot and an empty queue and runs the plain
//     single-call path at once (no staging, no extra copy, nothing to wait for);
//   * optionally (window_us > 0) a leader that follows a multi-caller batch lingers until arrivals stop (no new request for
//     gap_us) or the window is over - callers released by the previous batch come back within microseconds of each other;
//   * isolation: a request that cannot ride in a batch (NaN / Inf or out-of-range query, a batch whose search failed) is
//     handed back to its own thread to run ALONE through the plain path - so every caller gets exactly the status and message
//     a lone call would have given it, and one caller's bad query never fails another's call.
//
// Results are the exact top-k under (score desc, id asc) whatever the batch, and the top-k of a query is the
// prefix of its top-kmax: every caller's ids and scores are bit-equal to a lone call's.
#pragma once
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cgv {

struct CoReq {
    const float* q = nullptr;      // caller's buffers (host; pageable or pinned)
    uint32_t nq = 0, k = 0, kclass = 0;
    uint64_t* out_idx = nullptr;
    float* out_score = nullptr;
    enum State : int { WAITING = 0, TAKEN = 1, DONE = 2, ALONE = 3 };
    int state = WAITING;           // read and written under the coalescer's mutex only
    int outcome = ALONE;           // written by the batch runner (which holds the request exclusively while it is TAKEN): DONE or
                                   // ALONE; the leader publishes it as `state` under the mutex once the batch has run
    int rc = 0;                    // DONE: the status of this caller's call ...
    std::string err;               // ... and its message
    uint32_t off = 0;              // first query slot of this request in the batch (set by the batch runner)
    std::condition_variable cv;
};

struct CoStats {                   // cgv_get_coalesce_stats
    uint64_t batches = 0;          // device batches that carried more than one caller
    uint64_t batched_requests = 0; // callers served by those batches
    uint64_t batched_queries = 0;
    uint64_t lone_calls = 0;       // eligible calls that ran alone (free slot, empty queue)
    uint64_t retried_alone = 0;    // requests handed back to their own thread (isolation)
    uint64_t max_batch_queries = 0;
    uint64_t window_waits = 0;     // batches whose leader lingered for more arrivals
    uint64_t reserved = 0;
};

class Coalescer {
   public:
    // policy (cgv_set_coalesce). Changed under mu; eligible() reads its three words without it (relaxed atomics: a call that
    // races with a change is served under either setting)
    std::atomic<bool> enabled{true};
    std::atomic<uint32_t> max_req_nq{8};    // a call with more queries than this is its own batch already
    std::atomic<uint32_t> max_batch_q{64};  // one COARSE_TOP2 launch holds 64 query columns
    int max_leaders = 2;                    // batches on the device at once
    uint32_t window_us = 0, gap_us = 0;
    size_t max_q_bytes = 0, max_out_bytes = 0;   // pinned staging area of a search context (cgvec.hip: SearchCtx::h_stage); set once

    bool eligible(uint32_t nq, uint32_t k, uint32_t dim) const {
        return enabled.load(std::memory_order_relaxed) && nq >= 1 && nq <= max_req_nq.load(std::memory_order_relaxed) &&
               nq <= max_batch_q.load(std::memory_order_relaxed) && (size_t)nq * dim * 4 <= max_q_bytes &&
               (size_t)nq * k * 12 <= max_out_bytes;
    }

    void configure(bool on, uint32_t batch_q, int leaders, uint32_t window) {
        std::lock_guard<std::mutex> lk(mu_);
        enabled.store(on, std::memory_order_relaxed);
        if (on) {
            max_batch_q.store(batch_q, std::memory_order_relaxed);
            max_leaders = leaders;
        }
        window_us = window;
        // (a thread waiting for a slot is woken by whoever frees one; raising max_leaders takes effect at the next hand-over)
    }

    CoStats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return st_;
    }

    // run_alone(): the plain single-call path for r (returns its status; the thread-local message is already set).
    // run_batch(reqs, nq_total, kmax): ONE search for all of them; sets outcome = DONE (+ rc, err) or ALONE on every request
    // (never `state`: the owner reads that under the mutex, and may return - freeing the request - as soon as it is final).
    // set_error(rc, msg): installs a follower's status message in ITS thread.
    template <class RunAlone, class RunBatch, class SetError>
    int submit(CoReq& r, uint32_t dim, RunAlone&& run_alone, RunBatch&& run_batch, SetError&& set_error) {
        std::unique_lock<std::mutex> lk(mu_);
        queue_.push_back(&r);
        queued_q_ += r.nq;
        ++arrivals_;
        for (;;) {
            if (r.state == CoReq::DONE) return r.rc ? set_error(r.rc, r.err) : 0;
            if (r.state == CoReq::ALONE) {
                ++st_.retried_alone;
                lk.unlock();
                return run_alone();
            }
            if (r.state == CoReq::WAITING && leaders_ < max_leaders) {
                ++leaders_;
                if (window_us > 0 && last_batch_requests_ > 1 && queued_q_ < max_batch_q.load(std::memory_order_relaxed)) linger(lk);
                std::vector<CoReq*> batch;
                uint32_t nq_total = 0, kmax = 0;
                take(batch, nq_total, kmax, dim);
                if (!queue_.empty() && leaders_ < max_leaders) queue_.front()->cv.notify_one();
                const bool lone = batch.size() == 1 && batch[0] == &r;
                lk.unlock();
                int rc_lone = 0;
                if (lone) rc_lone = run_alone();
                else run_batch(batch, nq_total, kmax);
                lk.lock();
                --leaders_;
                last_batch_requests_ = (uint32_t)batch.size();
                if (lone) {
                    ++st_.lone_calls;
                } else {
                    ++st_.batches;
                    st_.batched_requests += batch.size();
                    st_.batched_queries += nq_total;
                    if (nq_total > st_.max_batch_queries) st_.max_batch_queries = nq_total;
                    for (CoReq* b : batch) {
                        b->state = b->outcome;
                        if (b != &r) b->cv.notify_one();   // (b is not touched after this: its owner may return and free it)
                    }
                }
                if (!queue_.empty()) queue_.front()->cv.notify_one();
                if (lone) return rc_lone;
                continue;   // my own request was in the batch (DONE / ALONE) - or is still queued behind a full batch
            }
            r.cv.wait(lk);
        }
    }

   private:
    // FIFO from the head: one k class per batch, at most max_batch_q queries, what fits the staging area
    void take(std::vector<CoReq*>& batch, uint32_t& nq_total, uint32_t& kmax, uint32_t dim) {
        while (!queue_.empty()) {
            CoReq* f = queue_.front();
            const uint32_t nq2 = nq_total + f->nq, k2 = f->k > kmax ? f->k : kmax;
            if (!batch.empty() && (f->kclass != batch[0]->kclass || nq2 > max_batch_q.load(std::memory_order_relaxed) || (size_t)nq2 * dim * 4 > max_q_bytes ||
                                   (size_t)nq2 * k2 * 12 > max_out_bytes))
                break;
            queue_.pop_front();
            queued_q_ -= f->nq;
            f->state = CoReq::TAKEN;
            batch.push_back(f);
            nq_total = nq2;
            kmax = k2;
        }
    }
    // bounded wait for the callers the previous batch has just released: until the batch is full, arrivals have stopped for
    // gap_us, or window_us are over. The lock is dropped while waiting (arrivals need it).
    void linger(std::unique_lock<std::mutex>& lk) {
        using clk = std::chrono::steady_clock;
        ++st_.window_waits;
        const auto t0 = clk::now();
        const auto deadline = t0 + std::chrono::microseconds(window_us);
        const auto gap = std::chrono::microseconds(gap_us ? gap_us : (window_us + 3) / 4);
        uint64_t seen = arrivals_;
        auto last = t0;
        while (queued_q_ < max_batch_q.load(std::memory_order_relaxed)) {
            lk.unlock();
            std::this_thread::yield();
            lk.lock();
            const auto now = clk::now();
            if (arrivals_ != seen) {
                seen = arrivals_;
                last = now;
            }
            if (now >= deadline || now - last >= gap) break;
        }
    }

    std::mutex mu_;
    std::deque<CoReq*> queue_;
    uint32_t queued_q_ = 0;
    uint64_t arrivals_ = 0;
    int leaders_ = 0;
    uint32_t last_batch_requests_ = 0;
    CoStats st_;
};

}  // namespace cgv
