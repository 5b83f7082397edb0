#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: queries/sec (+ recall@10) of the
brute-force cosine kNN at BASELINE.json config C2 (1M x 768 bf16, batch = 1024, k = 10).

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL. Started either by the launcher
  (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) or plainly
  (`python bench.py --gpus N`): with no WORLD_SIZE in the environment the script re-executes itself
  under torch.distributed.run on 127.0.0.1. The 1M-row corpus is row-sharded across the ranks
  (STRONG scaling: BASELINE's metric is quoted on ONE 1M-row corpus), every rank searches its shard
  with the full query batch, the per-shard top-k are combined with one RCCL all-gather of packed
  12-byte records + a merge kernel.

A "step" is SURVEY.md §8(d)'s unit: ONE batched search of 1024 queries against the whole corpus,
corpus resident in HBM, queries starting in (pinned) HOST memory and the B*k results ending in
host memory — H2D of the queries and D2H of the results are inside the step; steps are strictly
serial (one batch at a time). `value` = batch * K / wall time of the K timed steps (max over
ranks); `median_qps` is the same from the median step. The device-resident, two-batches-in-flight
rate of round 1 is reported beside it as `pipelined_qps`, never as `value`.
Other workloads (--workload): c4, c3shard, c5shard, c5mini, c2shard8, small, c2f32 — parity /
sizing cases of BASELINE.json, not the headline line.

  roofline     : the dominant kernel (MFMA coarse GEMM with fused top-k') — algorithmic
                 FLOPs 2*B*rows*D of one launch / its HIP-event duration (events recorded by
                 the library on the stream the kernel runs on), vs the 2.5 PFLOP/s dense bf16
                 MFMA peak (MI355X_MICROARCH.md); `traffic` = HBM bytes per launch from the
                 committed rocprofv3 PMC passes (profiles/).
  cpu_baseline : the CPU oracle (a port of the reference's parallel_top_k_search,
                 simd_ops.rs:361-383: AVX2+FMA scoring of separately allocated rows + full
                 parallel sort) timed on this box's host cores on a bounded sample of the same
                 workload (rank 0, N = 1 only).
"""
import argparse
import collections
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows, dim, dtype, metric, batch, k)
    "c2": (1_000_000, 768, "bf16", "cosine", 1024, 10),
    "c4": (1_000_000, 1536, "fp16", "dot", 256, 10),
    "c3shard": (1_250_000, 768, "bf16", "cosine", 4096, 10),   # one GPU's share of C3 (10M rows / 8)
    "small": (100_000, 768, "bf16", "cosine", 1024, 10),
    "c2shard8": (125_000, 768, "bf16", "cosine", 1024, 10),    # one rank's share of C2 at 8 GPUs (fixed-cost probe)
    # C5 = 500M x 768 fp8 over 8 GPUs, batch 8192: one GPU's share is 62.5M rows = 48 GB of codes
    "c5shard": (62_500_000, 768, "fp8", "cosine", 8192, 10),
    "c5mini": (4_000_000, 768, "fp8", "cosine", 8192, 10),   # same kernel shape, 1/16 of the shard
    # C2's shape on UNROUNDED f32 rows (results = the reference's own f32 arithmetic): f32 + bf16 shadow
    "c2f32": (1_000_000, 768, "f32s", "cosine", 1024, 10),
}
CHUNK = 125_000
# dense MFMA peaks (MI355X_MICROARCH.md); the fp8 path runs on the block-scaled K=64 MFMA (5 PF class)
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp8": 5000.0, "f32s": 2500.0}
PEAK_HBM_GBS = 8000.0   # HBM3E, MI355X_MICROARCH.md
ESIZE = {"bf16": 2, "fp16": 2, "fp8": 1, "f32s": 2}   # bytes per element the coarse kernel streams
SEED_CORPUS, SEED_QUERY = 0xC0DE6001, 0xC0DE6002


def storage_values(x, dtype):
    """f32 values the index scores on (rounded-then-upcast); fp8 = e4m3fn codes
    under the per-row power-of-two scale (largest e with amax * 2^e <= 448)."""
    if dtype == "f32s":
        return x
    if dtype == "bf16":
        return x.to(torch.bfloat16).float()
    if dtype == "fp16":
        return x.to(torch.float16).float()
    amax = x.abs().amax(dim=1, keepdim=True)
    mant, ex = torch.frexp(amax)                      # amax = mant * 2^ex, mant in [0.5, 1)
    e = torch.where(mant <= 0.875, 9 - ex, 8 - ex)     # 2*mant <= 1.75  ->  8 - (ex - 1)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    return torch.ldexp(x, e).to(torch.float8_e4m3fn).float()


def gen_chunk(c, rows, dim, device):
    g = torch.Generator(device=device).manual_seed(SEED_CORPUS + c)
    x = torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)
    return torch.nn.functional.normalize(x, dim=1)


def respawn_under_launcher(n):
    """`python bench.py --gpus N` without a launcher: run the same command as N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--pipelined-steps", type=int, default=20,
                    help="extra device-resident batches kept `--depth` in flight (0 = skip)")
    ap.add_argument("--depth", type=int, default=2, help="batches in flight of the pipelined side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--cpu-max-queries", type=int, default=128)
    ap.add_argument("--spawn-check", action="store_true",
                    help="print this rank's RANK/WORLD_SIZE and exit before touching a GPU (CPU test of the self-spawn)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn_under_launcher(args.gpus))

    n_total, dim, dtype, metric, batch, k = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.spawn_check:
        print(json.dumps({"spawn_check": True, "rank": rank, "local_rank": local_rank, "world": world,
                          "master": os.environ.get("MASTER_ADDR")}), flush=True)
        return
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m = importlib.import_module("codegraph-rust_amd")
    lo, hi = m.shard_range(n_total, rank, world)
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=local_rank)
    ix.reserve(hi - lo)
    ix.set_index_base(lo)
    want_cpu = (world == 1 and rank == 0 and args.cpu_seconds > 0)
    host_chunks = []
    if n_total > 4_000_000:
        want_cpu = False   # the f32 upcast of the corpus would not fit the CPU leg's time/memory bound
    nchunks = (n_total + CHUNK - 1) // CHUNK
    for c in range(nchunks):
        c_lo, c_hi = c * CHUNK, min(n_total, (c + 1) * CHUNK)
        a, b = max(lo, c_lo), min(hi, c_hi)
        if a >= b:
            continue
        x = gen_chunk(c, c_hi - c_lo, dim, dev)[a - c_lo: b - c_lo]
        ix.add(x)
        if want_cpu:
            host_chunks.append(storage_values(x, dtype).cpu().numpy())   # rounded-then-upcast values
        del x
    gq = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    npool = 4
    qpool = [torch.nn.functional.normalize(torch.randn((batch, dim), generator=gq, device=dev), dim=1)
             for _ in range(npool)]
    qhost = [q.cpu().pin_memory() for q in qpool]           # the caller's query batches: pinned host memory
    out_i = torch.empty((batch, k), dtype=torch.int64).pin_memory()
    out_s = torch.empty((batch, k), dtype=torch.float32).pin_memory()
    searcher = m.ShardedKnn(ix, rank=rank, world=world) if world > 1 else ix
    ix.set_profiling(True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1:
        L, C = m.cgvec.lib(), m.cgvec.C
        oi_p, os_p = C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr())

        def step(i):   # cgv_search_f32: host queries in, host results out (H2D + D2H inside)
            m.cgvec._check(L.cgv_search_f32(ix._h, C.c_void_p(qhost[i % npool].data_ptr()), batch, k, oi_p, os_p))
    else:
        def step(i):   # every rank: H2D of the (replicated) batch, shard search, all-gather + merge, D2H
            q = qhost[i % npool].to(dev, non_blocking=True)
            gi, gs = searcher.search(q, k)
            out_i.copy_(gi, non_blocking=True)
            out_s.copy_(gs, non_blocking=True)
            torch.cuda.current_stream().synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    coarse_ms, coarse_rows, step_ms = [], 0, []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(i)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        st = ix.stats()   # host-side read of that batch's HIP-event pair; no extra device sync
        coarse_ms.append(st["last_coarse_ms"])
        coarse_rows = st["coarse_rows"]
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()

    # side measurement: device-resident queries and results, `depth` batches in flight (round 1's headline)
    pipelined = None
    if args.pipelined_steps > 0:
        depth = max(1, min(args.depth, ix.max_in_flight))
        pend = collections.deque()
        sync_all()
        tp = time.perf_counter()
        for i in range(args.pipelined_steps):
            pend.append(searcher.search_begin(qpool[i % npool], k))
            if len(pend) >= depth:
                pend.popleft().wait()
        while pend:
            pend.popleft().wait()
        sync_all()
        dtp = time.perf_counter() - tp
        if dist is not None:
            t = torch.tensor([dtp], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtp = float(t.item())
        pipelined = {"queries_per_sec": round(batch * args.pipelined_steps / dtp, 1),
                     "ms_per_batch": round(1e3 * dtp / args.pipelined_steps, 4), "batches_in_flight": depth,
                     "note": "queries and results stay in HBM (cgv_search_begin_f32_dev / cgv_search_end)"}

    result = None
    if rank == 0:
        qps = batch * args.steps / elapsed
        med = float(np.median(step_ms))
        cms = float(np.mean([c for c in coarse_ms if c > 0])) if any(c > 0 for c in coarse_ms) else None
        roof = None
        if cms:
            flops = 2.0 * batch * coarse_rows * dim
            ach = flops / (cms * 1e-3) / 1e12
            traffic, traffic_src = None, None
            pdir = os.path.join(ROOT, "profiles")
            pmc = sorted(f for f in os.listdir(pdir) if f.endswith(f"_{args.workload}_pmc_main_kernel.json")) \
                if os.path.isdir(pdir) else []
            if pmc and world == 1:
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes
                # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; scripts/collect_profiles.sh)
                traffic = json.load(open(os.path.join(pdir, pmc[-1]))).get("hbm_bytes_per_launch")
                traffic_src = "profiles/" + pmc[-1]
            abytes = float(coarse_rows) * dim * ESIZE[dtype] + batch * dim * ESIZE[dtype] + coarse_rows * 4
            gbs = abytes / (cms * 1e-3) / 1e9
            mfma_frac, hbm_frac = ach / PEAK_TFLOPS[dtype], gbs / PEAK_HBM_GBS
            # report against whichever roof binds this shape (intensity ~ batch FLOP/B:
            # batch >= ~512 -> MFMA, C4's batch 256 -> HBM); the other fraction rides along.
            if hbm_frac > mfma_frac:
                roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4)}
            else:
                roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                        "frac": round(mfma_frac, 4), "hbm_frac": round(hbm_frac, 4)}
            roof.update({"kernel": "coarse_kernel (main stage)", "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "avg_launch_ms": round(cms, 4),
                         "rows_per_launch": int(coarse_rows), "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": abytes, "rank": 0})
        result = {
            "metric": "queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {n_total} x {dim} {dtype} {metric} brute-force kNN, "
                                   f"batch={batch}, k={k}", "rows": n_total, "dim": dim, "batch": batch, "k": k,
                       "metric": metric, "sharding": f"rows/{world}" if world > 1 else "none",
                       "step": "one batch: pinned host queries -> H2D -> search -> D2H host results, serial batches",
                       "exchange": f"RCCL all-gather, world={world}" if world > 1 else "none"},
            "median_ms_per_step": round(med, 4), "median_qps": round(batch / (med * 1e-3), 1),
            "pipelined_qps": pipelined["queries_per_sec"] if pipelined else None,
            "pipelined": pipelined,
            "roofline": roof,
            "pipeline": {"device_ms_last_step": round(st["last_total_ms"], 4), "kprime": st["last_kprime"],
                         "fallback_queries": int(st["fallback_queries"]), "eps": st["last_eps"],
                         "max_observed_coarse_err": st["max_observed_err"]},
        }

    if want_cpu:
        from oracle import oracle as o   # CPU baseline + recall checker only
        rows_host = np.concatenate(host_chunks)
        del host_chunks
        rs = o.RowSet(rows_host)
        del rows_host
        cores = o.max_threads()
        qh = storage_values(qpool[0][:args.cpu_max_queries], dtype).cpu().numpy()
        if dtype == "fp8":   # the torch expression of the storage format must be the oracle's
            assert np.array_equal(qh, o.round_trip(qpool[0][:args.cpu_max_queries].cpu().numpy(), o.FP8, fp8_codes=True))
        gi, gs = ix.search(qpool[0], k)
        gi = gi.cpu().numpy().view(np.uint64)
        gs = gs.cpu().numpy()
        omet = o.COSINE if metric == "cosine" else o.DOT
        rs.top_k(qh[0], k, omet, cores)   # warm-up
        t0 = time.perf_counter()
        nqc, hits, ordered, exact_scores = 0, 0, 0, 0
        while nqc < args.cpu_max_queries and (nqc < 4 or time.perf_counter() - t0 < args.cpu_seconds):
            ri, rsc = rs.top_k(qh[nqc], k, omet, cores)
            hits += len(set(ri.tolist()) & set(gi[nqc].tolist()))
            ordered += int(np.array_equal(ri, gi[nqc]))
            exact_scores += int(np.array_equal(rsc, gs[nqc]))
            nqc += 1
        cpu_t = time.perf_counter() - t0
        rs.close()
        result["cpu_baseline"] = {"value": round(nqc / cpu_t, 3), "unit": "queries/s", "cores": cores,
                                  "kind": "port",
                                  "sample": f"{nqc} single-query searches over the same {n_total} x {dim} corpus "
                                            f"(f32 upcast of the {dtype} values, rows separately allocated), "
                                            f"{cpu_t:.1f} s wall; C++ port of parallel_top_k_search: threaded AVX2 "
                                            f"scoring + threaded sort whose last merge levels are single-threaded "
                                            f"(rayon's par_sort_unstable_by is not) - a slight under-estimate"}
        result["recall_at_10"] = hits / (nqc * k)
        result["ordered_match_rate"] = ordered / nqc
        result["score_bit_exact_rate"] = exact_scores / nqc
        result["recall_sample"] = f"{nqc} of the {batch} queries of one batch"
        result["speedup_vs_cpu_baseline"] = round(result["value"] / (nqc / cpu_t), 1)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
