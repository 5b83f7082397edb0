"""Stress for the flaky abort of the full GPU suite (HISTORY.md §9.5): register a HEAP numpy buffer with
hipHostRegister, let the library read queries from it, unregister, free it, then hand fresh heap buffers (likely the same
addresses) to cgv_add_f32 / cgv_search_f32 as pageable memory. Prints one line per phase; an abort shows the runtime's message."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("codegraph-rust_amd")
rt = torch.cuda.cudart()
rng = np.random.default_rng(0)
d, nq, k = 256, 300, 10
rows = rng.standard_normal((30_000, d)).astype(np.float32)
ix = m.HipKnnIndex(d, dtype="bf16")
ix.add(rows)
oi = torch.empty((nq, k), dtype=torch.int64).pin_memory()
osc = torch.empty((nq, k), dtype=torch.float32).pin_memory()
big = np.empty(40 << 20, dtype=np.uint8)   # raise glibc's mmap threshold: later buffers come from the heap
del big
t0 = time.time()
it = 0
limit = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
while time.time() - t0 < limit:
    q = rng.standard_normal((nq, d)).astype(np.float32)
    reg = np.ascontiguousarray(q.copy())
    rc = int(rt.cudaHostRegister(reg.ctypes.data, reg.nbytes, 0))
    if rc == 0:
        ix.search_host_ptr(reg.ctypes.data, nq, k, oi.data_ptr(), osc.data_ptr())
        urc = int(rt.cudaHostUnregister(reg.ctypes.data))
        if urc != 0:
            print("unregister rc", urc, flush=True)
    del reg
    # fresh heap buffers, pageable, through the runtime's pageable-copy path
    r2 = rng.standard_normal((3 * 4096 + 77, 64)).astype(np.float32)
    ix2 = m.HipKnnIndex(64, dtype="bf16")
    ix2.add(r2)
    ix2.search(r2[:9], 10)
    ix2.close()
    q2 = np.ascontiguousarray(q.copy())
    ix.search(q2, k)
    it += 1
    if it % 20 == 0:
        print("iterations", it, "register rc", rc, flush=True)
print("done", it, flush=True)
ix.close()
