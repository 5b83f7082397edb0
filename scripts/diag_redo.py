#!/usr/bin/env python3
"""Which queries of bench.py's query pool fail the guarantee check on a rank's shard of a workload, and why (the fallback words of
cgv_debug_fbflags_: 1 = exact scan | why << 8; 2 = repaired by cell re-scan). python scripts/diag_redo.py [workload] [world]
With the measurement flavour loaded (CGV_LIB_PATH=.../libcgvec_hip_ablate.so) a batch that needs a repair is also TIMED both ways:
serial device-resident calls with the in-kernel repair (sample_repair = 1) and with the exact-scan fallback it replaced (0)."""
import ctypes as C
import importlib
import os
import sys

import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "mid"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[wl]
m = importlib.import_module("codegraph-rust_amd")
L = m.cgvec.lib()
ablate = "ablate" in os.environ.get("CGV_LIB_PATH", "")
if ablate:
    L.cgv_debug_set_.argtypes = [C.c_char_p, C.c_double]
dev = torch.device("cuda", 0)
qpool = bench.gen_query_pool(4, batch, dim, dev)
for rank in range(world):
    lo, hi = m.shard_range(n_total, rank, world)
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(hi - lo)
    ix.set_index_base(lo)
    nchunks = (n_total + bench.CHUNK - 1) // bench.CHUNK
    for c in range(nchunks):
        c_lo, c_hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        a, b = max(lo, c_lo), min(hi, c_hi)
        if a < b:
            ix.add(bench.gen_chunk(c, c_hi - c_lo, dim, dev)[a - c_lo: b - c_lo])
    plan = (C.c_uint32 * 16)()
    nw = L.cgv_debug_plan_(C.c_uint64(hi - lo), k, batch, 256, 0, plan, 16)
    print(f"rank {rank}/{world}: rows [{lo}, {hi}) plan {list(plan[:nw])}", flush=True)
    for p, q in enumerate(qpool):
        f0, r0 = ix.stats()["fallback_queries"], ix.sample_repairs()
        idx, sc = ix.search(q, k)
        torch.cuda.synchronize()
        st = ix.stats()
        flags = np.zeros(batch, dtype=np.uint32)
        for ctx in range(3):
            fl = np.zeros(batch, dtype=np.uint32)
            if L.cgv_debug_fbflags_(ix._h, ctx, fl.ctypes.data_as(C.c_void_p), batch) == 0:
                flags |= fl
        bad = np.nonzero(flags)[0]
        if ablate and (ix.sample_repairs() - r0 > 0 or st["fallback_queries"] - f0 > 0):
            for knob in (1, 0, 1, 0):
                assert L.cgv_debug_set_(b"sample_repair", float(knob)) == 0
                for _ in range(5):
                    ix.search(q, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    ix.search(q, k)
                torch.cuda.synchronize()
                print(f"    sample_repair={knob}: {1e1 * (time.perf_counter() - t0):.4f} ms per batch (100 serial device-resident calls)", flush=True)
            L.cgv_debug_set_(b"sample_repair", 1.0)
            qc = qpool[(p + 1) % len(qpool)]   # a neighbouring batch (clean unless listed itself), timed the same way
            for _ in range(5):
                ix.search(qc, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                ix.search(qc, k)
            torch.cuda.synchronize()
            print(f"    batch {(p + 1) % len(qpool)} of the pool for reference: {1e1 * (time.perf_counter() - t0):.4f} ms per batch", flush=True)
        print(f"  batch {p}: fallback +{st['fallback_queries'] - f0} put right in the final kernel +{ix.sample_repairs() - r0} path {st['last_path']} kprime {st['last_kprime']} eps {st['last_eps']:.3g} "
              f"max_err {st['max_observed_err']:.3g} flagged {[(int(i), hex(int(flags[i]))) for i in bad[:8]]}", flush=True)
    ix.close()
