#!/usr/bin/env python3
"""Sweep of the group-commit policy (cgv_set_coalesce: batches in flight, window) and of the caller count on one corpus:
native single-query caller threads (tests/c_client/callers.c) -> q/s, per-call p50 / p99, queries per device batch.
  python scripts/coalesce_sweep.py [rows] [dim] [dtype]            (GPU box; default 1_000_000 768 bf16 = C2's corpus)"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    k = 10
    m = importlib.import_module("codegraph-rust_amd")
    dev = torch.device("cuda", 0)
    ix = m.HipKnnIndex(d, dtype=dtype)
    for c in range((n + bench.CHUNK - 1) // bench.CHUNK):
        ix.add(bench.gen_chunk(c, min(bench.CHUNK, n - c * bench.CHUNK), d, dev))
    g = torch.Generator(device=dev).manual_seed(bench.SEED_QUERY)
    q = torch.nn.functional.normalize(torch.randn((4096, d), generator=g, device=dev), dim=1).cpu().numpy()
    CL = bench.load_callers_lib(m)
    _, _, lat1, wall1 = bench.run_native_callers(CL, ix, q, k, 1, 100)
    print(json.dumps({"lone_p50_us": round(float(np.median(lat1)), 1), "lone_qps": round(100 / wall1, 1)}), flush=True)
    if os.environ.get("SWEEP") == "short":
        configs = [(64, 1, 150)] * 5 + [(64, 1, 250), (16, 1, 150), (32, 1, 150), (48, 1, 150), (96, 1, 150), (128, 1, 150)]
    else:
        configs = [(T, L, w) for T in (64,) for L in (1, 2) for w in (0, 60, 100, 150, 250)]
        configs += [(T, 1, 150) for T in (2, 3, 4, 8, 16, 32, 48, 96, 128, 256)] + [(64, 1, 150)] * 3
    for T, L, w in configs:
        ix.set_coalesce(64, L, w)
        calls = max(20, min(600, int(20000 / T)))
        s0 = ix.coalesce_stats()
        _, _, lat, wall = bench.run_native_callers(CL, ix, q, k, T, calls)
        s1 = ix.coalesce_stats()
        nb = max(1, s1["batches"] - s0["batches"])
        print(json.dumps({"threads": T, "in_flight": L, "window_us": w, "qps": round(T * calls / wall, 1),
                          "p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
                          "q_per_batch": round((s1["batched_queries"] - s0["batched_queries"]) / nb, 1), "batches": nb,
                          "us_per_batch": round(1e6 * wall / nb, 1), "alone": s1["lone_calls"] - s0["lone_calls"],
                          "window_waits": s1["window_waits"] - s0["window_waits"]}), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
