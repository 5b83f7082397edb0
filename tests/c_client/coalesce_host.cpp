// Host-only test driver of the group-commit layer (codegraph-rust_amd/csrc/coalesce.h) with a FAKE device: T caller threads, each
// in a serial loop of single-request calls, a batch runner that sleeps like a device batch and answers from the query values,
// a poisoned request every now and then (handed back to run alone, must fail its own caller only). Built with g++ (and once more
// under -fsanitize=thread) by tests/test_coalesce_host.py. Prints one JSON line; exit code 0 = every caller got its own answers.
//   usage: coalesce_host <threads> <calls per thread> <max_leaders> <window_us> <batch_sleep_us> [k classes]
//   k classes > 1: the callers fall into that many classes that never share a batch - single-member batches of a rare class get led
//   by threads of another one (the case that once slept forever: its one member was never woken)
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../codegraph-rust_amd/csrc/coalesce.h"

using namespace cgv;

static const uint32_t D = 8;
static thread_local std::string t_err;

static void answer(const float* q, uint32_t k, uint64_t* oi, float* os) {   // the "search": ids / scores derived from the query
    for (uint32_t j = 0; j < k; ++j) {
        oi[j] = (uint64_t)q[0] * 1000u + j;
        os[j] = q[1] - (float)j;
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 200, leaders = argc > 3 ? atoi(argv[3]) : 2;
    const uint32_t window = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
    const int sleep_us = argc > 5 ? atoi(argv[5]) : 200;
    const int kclasses = argc > 6 ? atoi(argv[6]) : 1;
    Coalescer co;
    co.max_q_bytes = 64 * D * 4;
    co.max_out_bytes = 64 * 16 * 12;
    co.configure(true, 64, leaders, window);
    std::atomic<long> bad{0}, alone_runs{0}, batch_runs{0}, poisoned_ok{0};
    std::atomic<int> in_flight{0}, max_in_flight{0};
    auto device = [&](int us) {
        const int now = ++in_flight;
        int m = max_in_flight.load();
        while (now > m && !max_in_flight.compare_exchange_weak(m, now)) {}
        std::this_thread::sleep_for(std::chrono::microseconds(us));
        --in_flight;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            for (int i = 0; i < N; ++i) {
                const uint32_t nq = 1 + (uint32_t)((t + i) % 3), k = (t % 2) ? 10u : 5u;   // mixed nq and k in one batch
                std::vector<float> q(nq * D);
                for (uint32_t j = 0; j < nq; ++j) {
                    q[j * D] = (float)(t * 100000 + i * 10 + (int)j);
                    q[j * D + 1] = (float)(t + i);
                }
                const bool poison = (i % 37) == 5 && (t % 4) == 1;
                if (poison) q[2] = -1.0f;   // the batch runner refuses it (stands for a NaN query)
                std::vector<uint64_t> oi(nq * k, ~0ull);
                std::vector<float> os(nq * k, -1.0f);
                CoReq r;
                r.q = q.data();
                r.nq = nq;
                r.k = k;
                r.kclass = (kclasses > 1 && t % 7 == 0) ? (uint32_t)(1 + (t / 7) % (kclasses - 1)) : 0u;   // rare classes
                r.out_idx = oi.data();
                r.out_score = os.data();
                if (!co.eligible(nq, k, D)) {
                    ++bad;
                    continue;
                }
                const int rc = co.submit(
                    r, D,
                    [&]() -> int {   // plain path
                        ++alone_runs;
                        device(sleep_us);
                        if (q[2] < 0.0f) {
                            t_err = "poisoned query";
                            return 5;
                        }
                        for (uint32_t j = 0; j < nq; ++j) answer(q.data() + j * D, k, oi.data() + j * k, os.data() + j * k);
                        return 0;
                    },
                    [&](std::vector<CoReq*>& batch, uint32_t nq_total, uint32_t kmax) {
                        ++batch_runs;
                        uint32_t seen = 0;
                        for (CoReq* b : batch) {
                            b->outcome = CoReq::PENDING;
                            if (b->q[2] < 0.0f) {
                                b->outcome = CoReq::ALONE;
                                continue;
                            }
                            seen += b->nq;
                        }
                        if (seen > nq_total || nq_total > 64) ++bad;
                        device(sleep_us + (int)nq_total);
                        for (CoReq* b : batch) {
                            if (b->outcome != CoReq::PENDING) continue;
                            if (b->k > kmax) ++bad;
                            for (uint32_t j = 0; j < b->nq; ++j) answer(b->q + j * D, b->k, b->out_idx + j * b->k, b->out_score + j * b->k);
                            b->rc = 0;
                            b->outcome = CoReq::DONE;
                        }
                    },
                    [&](int code, const std::string& msg) {
                        t_err = msg;
                        return code;
                    });
                if (poison) {
                    if (rc == 5 && t_err == "poisoned query") ++poisoned_ok;
                    else ++bad;
                    continue;
                }
                if (rc != 0) ++bad;
                for (uint32_t j = 0; j < nq; ++j)
                    for (uint32_t e = 0; e < k; ++e)
                        if (oi[j * k + e] != (uint64_t)q[j * D] * 1000u + e || os[j * k + e] != q[j * D + 1] - (float)e) ++bad;
            }
        });
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const CoStats st = co.stats();
    printf("{\"threads\": %d, \"calls\": %ld, \"bad\": %ld, \"batches\": %llu, \"batched_requests\": %llu, \"lone_calls\": %llu, "
           "\"retried_alone\": %llu, \"max_batch_queries\": %llu, \"window_waits\": %llu, \"poisoned_ok\": %ld, \"max_in_flight\": %d, "
           "\"seconds\": %.4f}\n",
           T, (long)T * N, bad.load(), (unsigned long long)st.batches, (unsigned long long)st.batched_requests,
           (unsigned long long)st.lone_calls, (unsigned long long)st.retried_alone, (unsigned long long)st.max_batch_queries,
           (unsigned long long)st.window_waits, poisoned_ok.load(), max_in_flight.load(), secs);
    return bad.load() == 0 ? 0 : 1;
}
