#!/usr/bin/env python3
"""Latency of the single-query call (C1: 10k x 384 f32, cgv_search_f32) under the things a bench does around it: profiling
levels, stats reads, stream / event / device-wide synchronisations. Finding (round 6): the first torch.cuda.synchronize() of a
process (hipDeviceSynchronize) is followed by a few milliseconds in which the call takes 60-70 us instead of 47 (first call 110-120
us, spikes of 200 us); stream and event synchronisations do not do that. bench.py therefore times 2000 steps on single-query
workloads.   python scripts/c1_probe.py [repo root]"""
import sys, time, importlib, ctypes as C
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
import torch, numpy as np
import bench
m = importlib.import_module("codegraph-rust_amd")
L = m.cgvec.lib()
n, d, k = 10000, 384, 10
dev = torch.device("cuda", 0)
ix = m.HipKnnIndex(d, metric="cosine", dtype="f32", device=0)
ix.add(bench.gen_chunk(0, n, d, dev))
q = [x.cpu().pin_memory() for x in bench.gen_query_pool(128, 1, d, dev)]
oi = torch.empty((1, k), dtype=torch.int64).pin_memory(); osc = torch.empty((1, k), dtype=torch.float32).pin_memory()
qp = [C.c_void_p(x.data_ptr()) for x in q]; oip, osp = C.c_void_p(oi.data_ptr()), C.c_void_p(osc.data_ptr())
def loop(nm, N=200):
    t0 = time.perf_counter()
    for i in range(N):
        L.cgv_search_f32(ix._h, qp[i % 128], 1, k, oip, osp)
    print(nm, round(1e6 * (time.perf_counter() - t0) / N, 1), "us/call", flush=True)
ix.use_own_stream()
for lvl in (0, 1, 1, 3, 1, 1, 0, 2, 1):
    ix.set_profiling(lvl); loop(f"profiling {lvl}")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    L.cgv_search_f32(ix._h, qp[0], 1, k, oip, osp)
loop("after 0.4 s settle, profiling 1")
st = ix.stats(); loop("after ix.stats()")
ix.phase_times_us(); loop("after phase_times")
torch.cuda.synchronize(); loop("after torch sync")
import gc; gc.collect(); loop("after gc.collect")

def series(nm, N=40):
    ts = []
    for i in range(N):
        t0 = time.perf_counter()
        L.cgv_search_f32(ix._h, qp[i % 128], 1, k, oip, osp)
        ts.append(1e6 * (time.perf_counter() - t0))
    print(nm, " ".join(f"{t:.0f}" for t in ts), flush=True)
loop("baseline again", 400)
series("baseline series")
torch.cuda.synchronize(); series("after torch.cuda.synchronize()")
loop("...", 400); series("later")
torch.cuda.current_stream().synchronize(); series("after torch current_stream sync")
loop("...", 400)
ix.synchronize(); series("after cgv_synchronize")
loop("...", 400)
e = torch.cuda.Event(); e.record(); e.synchronize(); series("after torch event record+sync")
loop("...", 400)
x = torch.zeros(16, device=dev); x += 1; torch.cuda.synchronize(); series("after a torch kernel + device sync")
loop("...", 1000); series("1000 calls later")
