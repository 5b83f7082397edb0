#!/usr/bin/env python3
"""In-process interleaved A/B of planner knobs (cgv_debug_set_) on one resident corpus.

  python scripts/ab.py --workload c2 --variants "legacy:plan_legacy=1;new:;m1:plan_launches=1" --rounds 3 --steps 15

Every variant runs `--steps` serial host-in/host-out batches per round, rounds interleaved (cdna guide §5.4 rule 24);
the results of every variant are compared with the first one's (ids and score bits). With --trace (needs
CGV_TRACE=1 in the environment) the host timeline of cgv_search_f32 and the final kernel's phase stamps are printed.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CGV_LIB_PATH", os.path.join(ROOT, "codegraph-rust_amd", "lib", "libcgvec_hip_ablate.so"))  # knobs live there
import bench  # noqa: E402  (workload table + generators)

KNOBS = {"plan_legacy": 0, "sample_tiles": 0, "plan_launches": 0, "hit_us": 1.7, "launch_us": 40.0, "zero_copy": 3, "pace": 1, "epi": 1, "fuse_sample": 0, "top2": 1, "sample_emit": 1, "top2_repair": 1, "fetch_queries": 1, "ladder": 0, "exact_small": 1, "self_publish": 1, "sample_repair": 1, "launch_events": 1, "profiling": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--variants", default="legacy:plan_legacy=1;new:")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--dev-resident", action="store_true", help="also time cgv_search_f32_dev with zero-copy host buffers")
    args = ap.parse_args()
    n_total, dim, dtype, metric, batch, k = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    m = importlib.import_module("codegraph-rust_amd")
    L = m.cgvec.lib()
    L.cgv_debug_set_.argtypes = [C.c_char_p, C.c_double]
    L.cgv_debug_plan_.restype = C.c_uint32
    L.cgv_debug_plan_.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.c_uint32]
    L.cgv_debug_trace_.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.c_void_p, C.c_uint32]
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=0)
    ix.reserve(n_total)
    for c in range((n_total + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c * bench.CHUNK, min(n_total, (c + 1) * bench.CHUNK)
        ix.add(bench.gen_chunk(c, hi - lo, dim, dev))
    gq = torch.Generator(device=dev).manual_seed(bench.SEED_QUERY)
    qpool = [torch.nn.functional.normalize(torch.randn((batch, dim), generator=gq, device=dev), dim=1) for _ in range(4)]
    qhost = [q.cpu().pin_memory() for q in qpool]
    out_i = torch.empty((batch, k), dtype=torch.int64).pin_memory()
    out_s = torch.empty((batch, k), dtype=torch.float32).pin_memory()
    ix.set_profiling(2)
    vp = C.c_void_p

    variants = []
    for v in args.variants.split(";"):
        name, _, kv = v.partition(":")
        knobs = dict(KNOBS)
        for item in filter(None, kv.split(",")):
            key, _, val = item.partition("=")
            knobs[key] = float(val)
        variants.append((name, knobs))

    def apply(knobs):
        for key, val in knobs.items():
            if key == "profiling":       # HIP events around the dominant launch / the pipeline (cgv_set_profiling)
                ix.set_profiling(int(val))
                continue
            assert L.cgv_debug_set_(key.encode(), float(val)) == 0, key

    def step(i):
        m.cgvec._check(L.cgv_search_f32(ix._h, vp(qhost[i % 4].data_ptr()), batch, k, vp(out_i.data_ptr()), vp(out_s.data_ptr())))

    res = {name: {"step_ms": [], "coarse_ms": [], "dev_ms": []} for name, _ in variants}
    ref = None
    plan = (C.c_uint32 * 32)()
    for name, knobs in variants:
        apply(knobs)
        w = L.cgv_debug_plan_(n_total, k, batch, 256, 1 if dtype == "f32s" else 0, plan, 32)
        res[name]["plan"] = list(plan[:w])
        for i in range(3):
            step(i)
        step(0)
        cur = (out_i.numpy().copy(), out_s.numpy().copy())
        if ref is None:
            ref = cur
        res[name]["same_as_first"] = bool(np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1].view(np.uint32), ref[1].view(np.uint32)))
        res[name]["fallbacks"] = int(ix.stats()["fallback_queries"])
    for r in range(args.rounds):
        for name, knobs in variants:
            apply(knobs)
            step(0)
            torch.cuda.synchronize()
            for i in range(args.steps):
                t0 = time.perf_counter()
                step(i)
                res[name]["step_ms"].append(1e3 * (time.perf_counter() - t0))
                st = ix.stats()
                res[name]["coarse_ms"].append(st["last_coarse_ms"])
                res[name]["dev_ms"].append(st["last_total_ms"])
    for name, _ in variants:
        d = res[name]
        print(json.dumps({"workload": args.workload, "variant": name, "plan": d["plan"], "same_as_first": d["same_as_first"],
                          "fallbacks_total": int(ix.stats()["fallback_queries"]),
                          "median_step_ms": round(float(np.median(d["step_ms"])), 4),
                          "min_step_ms": round(float(np.min(d["step_ms"])), 4),
                          "mean_step_ms": round(float(np.mean(d["step_ms"])), 4),
                          "median_dev_ms": round(float(np.median(d["dev_ms"])), 4),
                          "median_main_coarse_ms": round(float(np.median(d["coarse_ms"])), 4)}), flush=True)

    if args.dev_resident:
        # zero-copy probe: the library's device entry point reading pinned HOST queries and writing pinned HOST results
        apply(variants[-1][1])
        ts = []
        for i in range(3 + args.steps * args.rounds):
            t0 = time.perf_counter()
            m.cgvec._check(L.cgv_search_f32_dev(ix._h, vp(qhost[i % 4].data_ptr()), batch, k, vp(out_i.data_ptr()), vp(out_s.data_ptr())))
            ts.append(1e3 * (time.perf_counter() - t0))
        m.cgvec._check(L.cgv_search_f32_dev(ix._h, vp(qhost[0].data_ptr()), batch, k, vp(out_i.data_ptr()), vp(out_s.data_ptr())))
        zc = (out_i.numpy().copy(), out_s.numpy().copy())
        step(0)
        print(json.dumps({"workload": args.workload, "variant": variants[-1][0] + "+zero_copy_host_buffers",
                          "median_step_ms": round(float(np.median(ts[3:])), 4), "min_step_ms": round(float(np.min(ts[3:])), 4),
                          "same_as_search_f32": bool(np.array_equal(zc[0], out_i.numpy()) and np.array_equal(zc[1], out_s.numpy()))}), flush=True)

    if args.trace:
        apply(variants[-1][1])
        hosts, stamps = [], []
        hu = (C.c_double * 8)()
        st = np.zeros((batch, 8), dtype=np.uint64)
        for i in range(12):
            step(i)
            m.cgvec._check(L.cgv_debug_trace_(ix._h, 0, hu, vp(st.ctypes.data), batch))
            hosts.append(list(hu))
            stamps.append(st.copy())
        h = np.median(np.array(hosts[2:]), axis=0)
        print("host timeline us (median): order=%.1f h2d_enq=%.1f pipeline_enq=%.1f d2h_enq=%.1f stream_done=%.1f" % tuple(h[:5]))
        s = np.stack(stamps[2:]).astype(np.int64)            # [steps][nq][8], 100 MHz ticks
        t0 = s[:, :, 0].min(axis=1, keepdims=True)
        rel = (s[:, :, :8] - t0[:, :, None]) / 100.0         # us since the first workgroup started
        names = ["start", "keys", "topk", "rows_staged", "scored", "sorted", "outputs", "end"]
        print("final kernel phases, us since the first workgroup's start (median over steps of: min / median / max over queries)")
        for j, nme in enumerate(names):
            print("  %-12s min %.2f  med %.2f  max %.2f" % (nme, np.median(rel[:, :, j].min(axis=1)), np.median(np.median(rel[:, :, j], axis=1)),
                                                         np.median(rel[:, :, j].max(axis=1))))
        dur = (s[:, :, 1:8] - s[:, :, 0:7]) / 100.0
        print("  per-workgroup phase durations us (median over all): " + ", ".join("%s %.2f" % (names[j + 1], np.median(dur[:, :, j])) for j in range(7)))
    ix.close()


if __name__ == "__main__":
    main()
