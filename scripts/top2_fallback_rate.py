#!/usr/bin/env python3
"""How often does a small batch (COARSE_TOP2) send a query to the exact scan on random data? Expected: three of a query's
top-(k + 1) rows in one of 1024 cells: C(11, 3) / 1024^2 = 1.6e-4 per query for k = 10.
   python scripts/top2_fallback_rate.py --workload c2 --queries 4000
--clustered: a code-embedding-like corpus - "files" of 8 ADJACENT rows around one centre (siblings at cosine ~0.9) - and
queries near a file's centre: a query's best rows are neighbours in memory, i.e. in ONE cell of the small-batch launch, so nearly
every query fails the cell check; what matters then is how it is answered (re-scan of the offending cells vs the exact scan of the
whole corpus) and what the call costs."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--queries", type=int, default=4000)
    ap.add_argument("--clustered", action="store_true")
    args = ap.parse_args()
    n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    m = importlib.import_module("codegraph-rust_amd")
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(n_total)
    g = torch.Generator(device=dev).manual_seed(77)
    if args.clustered:
        import time
        centres = None
        for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
            lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
            nf = (hi - lo + 7) // 8
            cen = torch.nn.functional.normalize(torch.randn((nf, dim), generator=g, device=dev), dim=1)
            rows = cen.repeat_interleave(8, dim=0)[: hi - lo] + 0.33 / dim ** 0.5 * torch.randn((hi - lo, dim), generator=g, device=dev)
            ix.add(torch.nn.functional.normalize(rows, dim=1))
            if c == 0:
                centres = cen[: args.queries].clone()
        q = torch.nn.functional.normalize(centres + 0.2 / dim ** 0.5 * torch.randn(centres.shape, generator=g, device=dev), dim=1).cpu().numpy()
        args.queries = q.shape[0]
        for nq in (1, 16):
            s0 = ix.small_batch_stats()
            ts = []
            for lo in range(0, args.queries - nq + 1, nq):
                t0 = time.perf_counter()
                ix.search(q[lo:lo + nq], k)
                ts.append(time.perf_counter() - t0)
            s1 = ix.small_batch_stats()
            d = {kk: s1[kk] - s0[kk] for kk in s1}
            print(f"{args.workload} CLUSTERED nq={nq}: {d} ; call us p50 {1e6 * np.median(ts):.1f} p99 {1e6 * np.percentile(ts, 99):.1f}")
        ix.set_force_exact(True)
        idx_e, sc_e = ix.search(q[:64], k)
        ix.set_force_exact(False)
        idx_f, sc_f = ix.search(q[:64], k)
        print("clustered: fast path equals the exact scan on 64 queries:", bool(np.array_equal(idx_e, idx_f) and np.array_equal(sc_e, sc_f)))
        ix.close()
        return
    for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
    q = torch.nn.functional.normalize(torch.randn((args.queries, dim), generator=g, device=dev), dim=1).cpu().numpy()
    for nq in (1, 32, 64):
        f0 = ix.stats()["fallback_queries"]
        done = 0
        for lo in range(0, args.queries - nq + 1, nq):
            ix.search(q[lo:lo + nq], k)
            done += nq
        f1 = ix.stats()["fallback_queries"]
        print(f"{args.workload} nq={nq}: {f1 - f0} fallbacks in {done} queries = {(f1 - f0) / max(done, 1):.2e} per query")
    # the staged path on the same queries (batches of 512)
    f0 = ix.stats()["fallback_queries"]
    for lo in range(0, args.queries - 511, 512):
        ix.search(q[lo:lo + 512], k)
    print(f"{args.workload} nq=512 (staged thresholds): {ix.stats()['fallback_queries'] - f0} fallbacks")
    ix.close()


if __name__ == "__main__":
    main()
