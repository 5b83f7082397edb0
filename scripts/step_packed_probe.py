#!/usr/bin/env python3
"""Where the host time of ShardedKnn.step_packed goes (one rank, no process group: the all-gather of one rank is a view): time from
the start of step_packed_begin to the library call that enqueues the first kernel, length of that call, rest of begin, the wait,
end.   python scripts/step_packed_probe.py [repo root] [workload]"""
import sys, time, importlib, ctypes as C
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
import numpy as np, torch
import bench
wl = sys.argv[2] if len(sys.argv) > 2 else "c2shard8"
n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[wl]
m = importlib.import_module("codegraph-rust_amd")
L = m.cgvec.lib()
dev = torch.device("cuda", 0)
ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
    lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
    ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
qpool = bench.gen_query_pool(4, batch, dim, dev)
out = (torch.empty((batch, k), dtype=torch.int64).pin_memory(), torch.empty((batch, k), dtype=torch.float32).pin_memory())
sk = m.ShardedKnn(ix, rank=0, world=1)
marks = {}
real = L.cgv_search_packed_begin_f32_dev
class Wrap:
    def __call__(self, *a):
        marks["lib_in"] = time.perf_counter()
        r = real(*a)
        marks["lib_out"] = time.perf_counter()
        return r
L.cgv_search_packed_begin_f32_dev = Wrap()
for i in range(50):
    sk.step_packed(qpool[i % 4], k, out=out, device=dev)
rows = []
for i in range(300):
    t0 = time.perf_counter()
    b = sk.step_packed_begin(qpool[i % 4], k, out=out, device=dev)
    t1 = time.perf_counter()
    sk._wait(b.done)
    t2 = time.perf_counter()
    sk.step_packed_end(b)
    t3 = time.perf_counter()
    rows.append((marks["lib_in"] - t0, marks["lib_out"] - marks["lib_in"], t1 - marks["lib_out"], t2 - t1, t3 - t2, t3 - t0))
a = 1e6 * np.median(np.array(rows), axis=0)
print(f"{wl}: to the library call {a[0]:.1f} us | the call (enqueues the shard search + pack) {a[1]:.1f} | rest of begin (merge, event) {a[2]:.1f} | "
      f"wait {a[3]:.1f} | end {a[4]:.1f} | step {a[5]:.1f}")
