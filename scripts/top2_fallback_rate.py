#!/usr/bin/env python3
"""How often does a small batch (COARSE_TOP2) send a query to the exact scan on random data? Expected: three of a query's
top-(k + 1) rows in one of 1024 cells: C(11, 3) / 1024^2 = 1.6e-4 per query for k = 10.
   python scripts/top2_fallback_rate.py --workload c2 --queries 4000"""
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
    args = ap.parse_args()
    n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    m = importlib.import_module("codegraph-rust_amd")
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(n_total)
    for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
    g = torch.Generator(device=dev).manual_seed(77)
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
