#!/usr/bin/env python3
"""Host-side timing of the N = 1 `pipelined_host` loop (HipKnnIndex.search_begin_pinned / PendingSearch.wait, 3 in flight): how long
do begin and end take on the host, and how long does a batch take from its begin to the end of its wait?
   python scripts/piped_probe.py --workload c2shard8 --iters 60"""
import argparse
import collections
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
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--depth", type=int, default=3)
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
    qh = [torch.nn.functional.normalize(torch.randn((batch, dim), generator=g, device=dev), dim=1).cpu().pin_memory() for _ in range(4)]
    outs = [(torch.empty((batch, k), dtype=torch.int64).pin_memory(), torch.empty((batch, k), dtype=torch.float32).pin_memory())
            for _ in range(args.depth)]
    L = m.cgvec.lib()
    import ctypes as C
    vp = C.c_void_p
    alias = [(ix.device_alias(q)) for q in qh]
    oalias = [(ix.device_alias(a), ix.device_alias(b)) for a, b in outs]

    def run(raw):
        tb, te, tot = [], [], []
        pend = collections.deque()
        torch.cuda.synchronize()
        t00 = time.perf_counter()
        for i in range(args.iters):
            t0 = time.perf_counter()
            if raw:     # the C ABI directly on cached device aliases: no per-call Python checks
                t = C.c_uint64(0)
                m.cgvec._check(L.cgv_search_begin_f32_dev(ix._h, vp(alias[i % 4]), batch, k, vp(oalias[i % args.depth][0]),
                                                          vp(oalias[i % args.depth][1]), C.byref(t)))
                p = t.value
            else:
                p = ix.search_begin_pinned(qh[i % 4], k, outs[i % args.depth])
            tb.append(time.perf_counter() - t0)
            pend.append((p, t0))
            if len(pend) >= args.depth:
                p0, tstart = pend.popleft()
                t1 = time.perf_counter()
                if raw:
                    m.cgvec._check(L.cgv_search_end(ix._h, C.c_uint64(p0)))
                else:
                    p0.wait()
                t2 = time.perf_counter()
                te.append(t2 - t1)
                tot.append(t2 - tstart)
        while pend:
            p0, tstart = pend.popleft()
            if raw:
                m.cgvec._check(L.cgv_search_end(ix._h, C.c_uint64(p0)))
            else:
                p0.wait()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t00
        med = lambda x: 1e6 * float(np.median(x[5:]))
        print(f"{args.workload} {'C ABI, cached aliases' if raw else 'search_begin_pinned   '}: {1e3 * wall / args.iters:.4f} ms per batch; host time of begin "
              f"{med(tb):.1f} us, of end (wait) {med(te):.1f} us; begin -> end of wait {med(tot):.1f} us")

    ix.use_own_stream()
    knob = os.environ.get("PROBE_KNOB")
    if knob:
        L.cgv_debug_set_.argtypes = [C.c_char_p, C.c_double]          # measurement flavour (CGV_LIB_PATH=..._ablate.so): A/B of one 0 / 1 knob
    for v in ((1, 0, 1, 0) if knob else (None, None)):
        if knob:
            assert L.cgv_debug_set_(knob.encode(), float(v)) == 0, knob
            print(knob, "=", v)
        run(True)
        run(False)
    ix.close()


if __name__ == "__main__":
    main()
