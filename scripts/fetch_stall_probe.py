#!/usr/bin/env python3
"""The first host batches in flight after a stretch of HBM-resident work: begin time and completion interval of each of the first
batches of a 3-in-flight host-in / host-out run (HipKnnIndex.search_begin_pinned), after `--idle-ms` of device-resident searches.
With the measurement flavour (CGV_LIB_PATH=.../libcgvec_hip_ablate.so) `--fetch 0` makes the conversion kernel read the pinned
batch in place instead of the copy-engine fetch.   python scripts/fetch_stall_probe.py [--workload c2] [--fetch 0|1] [--idle-ms 500]"""
import argparse, collections, ctypes as C, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--fetch", type=int, default=-1)
ap.add_argument("--idle-ms", type=float, default=500.0)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
dev = torch.device("cuda", 0)
m = importlib.import_module("codegraph-rust_amd")
L = m.cgvec.lib()
if args.fetch >= 0:
    L.cgv_debug_set_.argtypes = [C.c_char_p, C.c_double]
    assert L.cgv_debug_set_(b"fetch_queries", float(args.fetch)) == 0
ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
ix.reserve(n_total)
for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
    lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
    ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
qpool = bench.gen_query_pool(4, batch, dim, dev)
qh = [q.cpu().pin_memory() for q in qpool]
outs = [(torch.empty((batch, k), dtype=torch.int64).pin_memory(), torch.empty((batch, k), dtype=torch.float32).pin_memory()) for _ in range(3)]
for rnd in range(args.rounds):
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < args.idle_ms:      # HBM-resident work: nothing crosses PCIe but 120 KB of results
        ix.search(qpool[0], k)
    pend, tb, ends = collections.deque(), [], []
    t00 = time.perf_counter()
    for i in range(16):
        t1 = time.perf_counter()
        pend.append(ix.search_begin_pinned(qh[i % 4], k, outs[i % 3]))
        tb.append(1e3 * (time.perf_counter() - t1))
        if len(pend) >= 3:
            pend.popleft().wait()
            ends.append(time.perf_counter())
    while pend:
        pend.popleft().wait()
        ends.append(time.perf_counter())
    iv = [1e3 * (b - a) for a, b in zip([t00] + ends[:-1], ends)]
    print(f"round {rnd} fetch={args.fetch}: begin ms " + " ".join(f"{x:.2f}" for x in tb) + " | completion intervals ms " + " ".join(f"{x:.2f}" for x in iv), flush=True)
ix.close()
