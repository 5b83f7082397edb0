// coalesce.h — group commit of concurrent SMALL searches on one index (host side only; no HIP in this file).
//
// The reference's trait-level call is ONE query: VectorStore::search_similar(&self, &[f32], limit) (traits.rs:14;
// surreal_store.rs:61-85; caller search.rs:114-117), and its only multi-query caller issues B independent concurrent
// single-query searches (search.rs:358-361, try_join_all). Through that unchanged surface every call streams the whole corpus
// for ONE query column (C2's corpus: 1.5 GB, 382 us, 2.6 k q/s per caller) while the device would serve 64 columns for about
// the price of one. So callers of cgv_search_f32 with a few queries each are merged here into one device batch:
//
//   * a call joins the batch that is currently FORMING (FIFO; up to max_batch_q queries, one k class, what fits the pinned
//     staging area - else a new batch is opened behind it). Whoever finds a free LEADER slot claims the oldest unclaimed batch,
//     runs it as ONE search with k = the largest k in it and scatters every caller's first k results to its own buffers; the
//     other members sleep on the batch's futex word and are released by ONE wake-all when it is done (one condition variable
//     per caller, notified under the lock, cost ~100 us of a 64-caller cycle: profiles/r06_coalesce_sweep.txt);
//   * while `max_leaders` batches are on the device, arrivals collect in the forming batch - that wait IS the batching window:
//     the busier the index, the larger the batches. A lone caller finds a free slot and an empty queue and runs the plain
//     single-call path at once (no staging, no extra copy, nothing to wait for);
//   * a leader that claims a batch right behind a multi-caller batch lingers (at most window_us) until as many callers have
//     joined as that batch had, or arrivals stop, and the batch keeps taking members meanwhile: the callers a batch has just
//     released come back within tens of microseconds of each other, and a batch started by the first of them would carry one
//     query and make all the others wait for the next one;
//   * a leader whose own request still sits in an unclaimed batch when it is done leads the next batch itself (it is awake and
//     on a CPU: no wake-up latency on the device's critical path); otherwise it hands over with one wake-up;
//   * isolation: a request that cannot ride in a batch (NaN / Inf or out-of-range query, a batch whose search failed) is
//     handed back to its own thread to run ALONE through the plain path - so every caller gets exactly the status and message
//     a lone call would have given it, and one caller's bad query never fails another's call.
//
// Results are the exact top-k under (score desc, id asc) whatever the batch, and the top-k of a query is the
// prefix of its top-kmax: every caller's ids and scores are bit-equal to a lone call's.
#pragma once
#include <limits.h>
#include <linux/futex.h>
#include <stdint.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cgv {

struct CoBatch;

struct CoReq {
    const float* q = nullptr;      // caller's buffers (host; pageable or pinned)
    uint32_t nq = 0, k = 0, kclass = 0;
    uint64_t* out_idx = nullptr;
    float* out_score = nullptr;
    enum Outcome : int { PENDING = 0, DONE = 2, ALONE = 3 };
    int outcome = PENDING;         // written by the batch runner; published to the owner by the batch's word (release / acquire)
    int rc = 0;                    // DONE: the status of this caller's call ...
    std::string err;               // ... and its message
    uint32_t off = 0;              // first query slot of this request in the batch (set by the batch runner)
    std::shared_ptr<CoBatch> batch;
};

struct CoBatch {
    enum Word : uint32_t { WAITING = 0, NEED_LEADER = 1, CLAIMED = 2, FINISHED = 3 };
    // futex word the members sleep on: state in the low byte (>= CLAIMED: a leader has it, lingering or running), a hand-over
    // count above it - every hand-over CHANGES the word, so a member that has just looked at it and is about to sleep does not
    // sleep through the wake-up meant for it
    std::atomic<uint32_t> word{WAITING};
    static uint32_t state_of(uint32_t w) { return w & 0xffu; }
    uint32_t state(std::memory_order mo = std::memory_order_acquire) const { return state_of(word.load(mo)); }
    std::vector<CoReq*> reqs;              // members; touched under the coalescer's mutex until the batch leaves the queue, then by
                                           // its leader only, and by nobody once FINISHED is published (the owners may be gone)
    uint32_t nq = 0, kmax = 0, kclass = 0;
    bool sealed = false;                   // takes no more members (full, or its leader has started it)
};

struct CoStats {                   // cgv_get_coalesce_stats
    uint64_t batches = 0;          // device batches that carried more than one caller
    uint64_t batched_requests = 0; // callers served by those batches
    uint64_t batched_queries = 0;
    uint64_t lone_calls = 0;       // eligible calls that ran alone (free slot, nobody else around)
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
    int max_leaders = 1;                    // batches on the device at once (CGV_COALESCE_BATCHES_IN_FLIGHT)
    uint32_t window_us = 250, gap_us = 0;    // CGV_COALESCE_WINDOW_US; gap_us == 0: a third of the window
    size_t max_q_bytes = 0, max_out_bytes = 0;   // pinned staging area of a search context (cgvec_internal.h: SearchCtx::h_stage); set once

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
    }

    CoStats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        CoStats s = st_;
        s.retried_alone = retried_alone_.load(std::memory_order_relaxed);
        return s;
    }

    // run_alone(): the plain single-call path for r (returns its status; the thread-local message is already set).
    // run_batch(reqs, nq_total, kmax): ONE search for all of them; sets outcome = DONE (+ rc, err) or ALONE on every request.
    // set_error(rc, msg): installs a member's status message in ITS thread.
    template <class RunAlone, class RunBatch, class SetError>
    int submit(CoReq& r, uint32_t dim, RunAlone&& run_alone, RunBatch&& run_batch, SetError&& set_error) {
        std::unique_lock<std::mutex> lk(mu_, std::defer_lock);
        lock(lk);
        join(r, dim);
        ++arrivals_;
        const std::shared_ptr<CoBatch> mine = r.batch;
        uint32_t tried = 0xffffffffu;
        for (;;) {
            // (1) a free slot and an unclaimed batch, and my own request is not being served by somebody else already: lead it
            //     (this thread is awake and on a CPU; after it, the next one as long as my own batch is still unclaimed)
            while (leaders_ < max_leaders && mine->state() < CoBatch::CLAIMED) {
                std::shared_ptr<CoBatch> b = first_unclaimed();   // (never null here: `mine` itself is unclaimed and queued)
                ++leaders_;
                b->word.store(CoBatch::CLAIMED, std::memory_order_relaxed);   // (its sleepers stay asleep until FINISHED)
                if (window_us > 0 && last_batch_requests_ > 1 && !b->sealed && b == pending_.back()) linger(lk, b);
                remove(b);
                b->sealed = true;
                const bool lone = b->reqs.size() == 1 && b->reqs[0] == &r;
                lk.unlock();
                int rc_lone = 0;
                if (lone) rc_lone = run_alone();
                else run_batch(b->reqs, b->nq, b->kmax);
                const size_t members = b->reqs.size();
                const uint32_t nq_b = b->nq;
                b->word.store(CoBatch::FINISHED, std::memory_order_release);   // the members' requests are theirs again from here on
                lock(lk);
                --leaders_;
                last_batch_requests_ = (uint32_t)members;
                if (lone) {
                    ++st_.lone_calls;
                } else {
                    ++st_.batches;
                    st_.batched_requests += members;
                    st_.batched_queries += nq_b;
                    if (nq_b > st_.max_batch_queries) st_.max_batch_queries = nq_b;
                }
                // the slot is free: whoever leads the next batch is woken BEFORE this batch's members (its wake-up latency is on
                // the device's critical path, theirs is not) - unless this thread goes on to lead it itself (the loop condition)
                const bool go_on = mine->state(std::memory_order_relaxed) < CoBatch::CLAIMED;
                if (!go_on) handover();
                if (!lone) {   // (somebody else's request was in it - also when it is a batch of ONE that this thread led for its owner)
                    lk.unlock();
                    // the members are woken as a TREE: a few by this thread, two more by every member that wakes (step (2)) -
                    // one FUTEX_WAKE for all of them ran ~1.5 us per sleeper on this one CPU, and the last of 63 came back
                    // 100+ us after the first (profiles/r06_coalesce_sweep.txt); they leave without taking the lock
                    futex_wake(&b->word, WAKE_FANOUT);
                    if (!go_on) break;
                    lock(lk);
                } else {
                    lk.unlock();
                    return rc_lone;
                }
            }
            if (lk.owns_lock()) {
                handover();   // (a slot may be free while I do not lead: somebody must)
                lk.unlock();
            }
            // (2) sleep on my batch's word until it is FINISHED (no lock: the word publishes my request's outcome) - or until it
            //     says NEED_LEADER: a slot has become free, back to (1) under the lock
            //     (`tried`: the hand-over this thread has answered already - the slot was gone again by the time it held the lock;
            //     it sleeps on the word as it is, and the next hand-over, which changes the word, wakes it again)
            for (bool slept = false;; slept = true) {
                const uint32_t seen = mine->word.load(std::memory_order_acquire);
                if (CoBatch::state_of(seen) == CoBatch::FINISHED) {
                    // wake tree (`reqs` is frozen). More members than the leader wakes itself: all of them may be asleep - a batch
                    // can be led by a thread whose own request is in another one
                    if (slept && mine->reqs.size() > (size_t)WAKE_FANOUT) futex_wake(&mine->word, 2);
                    if (r.outcome == CoReq::DONE) return r.rc ? set_error(r.rc, r.err) : 0;
                    retried_alone_.fetch_add(1, std::memory_order_relaxed);
                    return run_alone();
                }
                if (CoBatch::state_of(seen) == CoBatch::NEED_LEADER && seen != tried) {
                    tried = seen;
                    break;
                }
                futex_wait(&mine->word, seen);
            }
            lock(lk);
        }
    }

   private:
    static constexpr int WAKE_FANOUT = 4;
    // the critical sections are tens of nanoseconds and 64 callers arrive within microseconds of each other: spin briefly before
    // sleeping on the mutex (a sleeping waiter costs two system calls and a wake-up latency per hand-over)
    void lock(std::unique_lock<std::mutex>& lk) {
        for (int i = 0; i < 400; ++i) {
            if (lk.try_lock()) return;
            __builtin_ia32_pause();
        }
        lk.lock();
    }
    static void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
        (void)syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
    }
    static void futex_wake(std::atomic<uint32_t>* w, int n) {
        (void)syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
    }

    // under mu: put r into the forming batch (the newest queued batch, if it still takes it) or open a new one behind it
    void join(CoReq& r, uint32_t dim) {
        std::shared_ptr<CoBatch> b = pending_.empty() ? nullptr : pending_.back();
        if (b) {
            const uint32_t nq2 = b->nq + r.nq, k2 = r.k > b->kmax ? r.k : b->kmax;
            if (b->sealed || b->kclass != r.kclass || nq2 > max_batch_q.load(std::memory_order_relaxed) ||
                (size_t)nq2 * dim * 4 > max_q_bytes || (size_t)nq2 * k2 * 12 > max_out_bytes)
                b = nullptr;
        }
        if (!b) {
            b = std::make_shared<CoBatch>();
            b->kclass = r.kclass;
            pending_.push_back(b);
        }
        b->reqs.push_back(&r);
        b->nq += r.nq;
        if (r.k > b->kmax) b->kmax = r.k;
        if (b->nq >= max_batch_q.load(std::memory_order_relaxed)) b->sealed = true;
        r.batch = b;
    }
    std::shared_ptr<CoBatch> first_unclaimed() const {
        for (const std::shared_ptr<CoBatch>& b : pending_)
            if (b->state(std::memory_order_relaxed) < CoBatch::CLAIMED) return b;
        return nullptr;
    }
    void remove(const std::shared_ptr<CoBatch>& b) {
        for (auto it = pending_.begin(); it != pending_.end(); ++it)
            if (*it == b) {
                pending_.erase(it);
                return;
            }
    }
    // under mu: if a slot is free and a batch waits unclaimed, make sure one of its sleepers comes to lead it (one wake-up)
    void handover() {
        if (leaders_ >= max_leaders) return;
        const std::shared_ptr<CoBatch> b = first_unclaimed();
        if (!b) return;
        const uint32_t w = b->word.load(std::memory_order_relaxed);
        b->word.store(CoBatch::NEED_LEADER | ((w & ~0xffu) + 0x100u), std::memory_order_release);
        futex_wake(&b->word, 1);
    }
    // Bounded wait for the callers the previous batch has just released. It ends when the batch has as many members as that one
    // had (everybody is back: the steady state of callers in serial loops pays no waiting time at all), when it is full, when
    // arrivals have stopped for `gap`, or after 3 gaps - gap grows with the number of callers expected back (N woken threads on
    // fewer cores return one after the other): window_us / 3 from 32 callers on, proportionally less below, at least 8 us.
    // The lock is dropped while waiting (arrivals need it); the batch is CLAIMED and stays in the queue, so it keeps taking
    // members and nobody else leads it.
    void linger(std::unique_lock<std::mutex>& lk, const std::shared_ptr<CoBatch>& b) {
        using clk = std::chrono::steady_clock;
        const uint32_t expected = last_batch_requests_;
        if (b->reqs.size() >= expected) return;
        ++st_.window_waits;
        const uint32_t full_gap = gap_us ? gap_us : (window_us + 2) / 3;
        const uint32_t g_us = std::max<uint32_t>(8u, std::min<uint32_t>(full_gap, (uint32_t)((uint64_t)full_gap * expected / 32u)));
        const auto t0 = clk::now();
        const auto deadline = t0 + std::chrono::microseconds(std::min<uint32_t>(window_us, 3u * g_us));
        const auto gap = std::chrono::microseconds(g_us);
        uint64_t seen = arrivals_;
        auto last = t0;
        while (!b->sealed && b == pending_.back() && b->reqs.size() < expected) {
            lk.unlock();
            std::this_thread::yield();
            lock(lk);
            const auto now = clk::now();
            if (arrivals_ != seen) {
                seen = arrivals_;
                last = now;
            }
            if (now >= deadline || now - last >= gap) break;
        }
    }

    std::mutex mu_;
    std::deque<std::shared_ptr<CoBatch>> pending_;   // batches not started yet, oldest first; the last one is the forming batch
    uint64_t arrivals_ = 0;
    int leaders_ = 0;
    uint32_t last_batch_requests_ = 0;
    CoStats st_;
    std::atomic<uint64_t> retried_alone_{0};
};

}  // namespace cgv
