#!/usr/bin/env python3
"""A loop of small cgv_search_f32 calls on one resident corpus (for rocprofv3 --kernel-trace: which kernels a single-query call
is made of and how long each takes):  python scripts/lat_loop.py --workload c2shard8 --nq 1 --iters 200"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2shard8")
    ap.add_argument("--nq", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    m = importlib.import_module("codegraph-rust_amd")
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(n_total)
    for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
    q = torch.nn.functional.normalize(torch.randn((args.nq, dim), device=dev), dim=1).cpu().numpy()
    oi, os_ = np.empty((args.nq, k), np.uint64), np.empty((args.nq, k), np.float32)
    ts = []
    for i in range(args.iters + 10):
        t0 = time.perf_counter()
        ix.search_host_ptr(q.ctypes.data, args.nq, k, oi.ctypes.data, os_.ctypes.data)
        ts.append(time.perf_counter() - t0)
    print("median us", round(1e6 * float(np.median(ts[10:])), 1), "p99", round(1e6 * float(np.percentile(ts[10:], 99)), 1))
    ix.close()


if __name__ == "__main__":
    main()
