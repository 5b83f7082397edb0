#!/usr/bin/env python3
"""T host threads, each issuing serial cgv_search_f32 calls on its own pinned buffers against ONE index (the reference's threading
model: a Send + Sync store called from a multi-thread tokio runtime through spawn_blocking): aggregate ms per batch.
   python scripts/threads_probe.py --workload c2shard8 --threads 3 --iters 100"""
import argparse
import importlib
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2shard8")
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    m = importlib.import_module("codegraph-rust_amd")
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(n_total)
    for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
    g = torch.Generator(device=dev).manual_seed(5)
    for T in (1, args.threads, 1, args.threads):
        bufs = []
        for t in range(T):
            q = torch.nn.functional.normalize(torch.randn((batch, dim), generator=g, device=dev), dim=1).cpu().pin_memory()
            bufs.append((q, torch.empty((batch, k), dtype=torch.int64).pin_memory(), torch.empty((batch, k), dtype=torch.float32).pin_memory()))
        ref = [ix.search(b[0].numpy(), k) for b in bufs]
        start = threading.Barrier(T + 1)

        def work(t):
            q, oi, os_ = bufs[t]
            for _ in range(5):
                ix.search_host_ptr(q.data_ptr(), batch, k, oi.data_ptr(), os_.data_ptr())
            start.wait()
            for _ in range(args.iters):
                ix.search_host_ptr(q.data_ptr(), batch, k, oi.data_ptr(), os_.data_ptr())

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        start.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        ok = all(bool((bufs[t][1].numpy().view("uint64") == ref[t][0]).all() and (bufs[t][2].numpy() == ref[t][1]).all()) for t in range(T))
        print(f"{args.workload}: {T} thread(s) x {args.iters} serial calls: {1e3 * dt / (T * args.iters):.4f} ms per batch "
              f"({batch * T * args.iters / dt / 1e3:.0f} k q/s); results equal to single calls: {ok}")
    ix.close()


if __name__ == "__main__":
    main()
