#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: queries/sec (+ recall@10) of the
brute-force cosine kNN at BASELINE.json config C2 (1M x 768 bf16, batch = 1024, k = 10).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`, one rank
   per GPU; the 1M-row corpus is row-sharded across ranks — STRONG scaling — and the
   per-shard partial top-k are combined with one RCCL all-gather + a merge kernel.)

A "step" is one batched search (1024 queries against the whole corpus) with corpus AND
queries already resident in HBM. `--depth` (default 2) batches are kept in flight through the
library's search-context pool (each batch on its own HIP stream); all K steps are begun and
completed inside the timed region. Rank 0 prints ONE JSON line.
Other workloads (--workload): c4, c3shard, c5shard, c5mini, c2shard8, small — parity / sizing
cases of BASELINE.json, not the headline line.

  roofline     : the dominant kernel (MFMA coarse GEMM with fused top-k') — algorithmic
                 FLOPs 2*B*rows*D of one launch / its HIP-event duration (events recorded by
                 the library on the stream the kernel runs on), vs the 2.5 PFLOP/s dense bf16
                 MFMA peak (MI355X_MICROARCH.md).
  cpu_baseline : the CPU oracle (a faithful port of the reference's
                 parallel_top_k_search, simd_ops.rs:361-383: AVX2+FMA scoring of separately
                 allocated rows + full parallel sort) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import collections
import importlib
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--host-io-steps", type=int, default=10, help="extra host-pointer-API batches (0 = skip)")
    ap.add_argument("--depth", type=int, default=2, help="batches in flight (1 = strictly serial steps)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--cpu-max-queries", type=int, default=128)
    args = ap.parse_args()

    n_total, dim, dtype, metric, batch, k = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
    searcher = m.ShardedKnn(ix, rank=rank, world=world) if world > 1 else ix
    ix.set_profiling(True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # `depth` batches in flight (the index' context pool: each batch on its own HIP stream), every
    # one of the K steps begun AND completed inside the timed region.
    depth = max(1, min(args.depth, ix.max_in_flight))
    for i in range(args.warmup):
        searcher.search(qpool[i % npool], k)
    sync_all()
    coarse_ms, coarse_rows = [], 0
    pend = collections.deque()

    def retire():
        nonlocal coarse_rows
        out = pend.popleft().wait()
        st = ix.stats()   # host-side read of that batch's HIP-event pair; no extra device sync
        coarse_ms.append(st["last_coarse_ms"])
        coarse_rows = st["coarse_rows"]
        return out

    t0 = time.perf_counter()
    for i in range(args.steps):
        pend.append(searcher.search_begin(qpool[i % npool], k))
        if len(pend) >= depth:
            out_idx, out_sc = retire()
    while pend:
        out_idx, out_sc = retire()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()

    result = None
    if rank == 0:
        qps = batch * args.steps / elapsed
        cms = float(np.mean([c for c in coarse_ms if c > 0])) if any(c > 0 for c in coarse_ms) else None
        roof = None
        if cms:
            flops = 2.0 * batch * coarse_rows * dim
            ach = flops / (cms * 1e-3) / 1e12
            traffic, traffic_src = None, None
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_main_kernel.json")) \
                if os.path.isdir(os.path.join(ROOT, "profiles")) else []
            if pmc and args.workload == "c2" and world == 1:
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes
                # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; scripts/collect_profiles.sh)
                traffic = json.load(open(os.path.join(ROOT, "profiles", pmc[-1]))).get("hbm_bytes_per_launch")
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
                         "algorithmic_bytes_per_launch": abytes})
        result = {
            "metric": "queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {n_total} x {dim} {dtype} {metric} brute-force kNN, "
                                   f"batch={batch}, k={k}", "rows": n_total, "dim": dim, "batch": batch, "k": k,
                       "metric": metric, "sharding": f"rows/{world}" if world > 1 else "none",
                       "batches_in_flight": depth},
            "roofline": roof,
            "pipeline": {"device_ms_last_step": round(st["last_total_ms"], 4), "kprime": st["last_kprime"],
                         "fallback_queries": int(st["fallback_queries"]), "eps": st["last_eps"],
                         "max_observed_coarse_err": st["max_observed_err"]},
        }

    if rank == 0 and world == 1 and args.host_io_steps > 0:
        # PCIe-inclusive rate of the host-pointer entry point (cgv_search_f32: queries H2D, results D2H,
        # serial batches). Reported beside `value`, never as it.
        qh_np = qpool[0].cpu().numpy()
        ix.search(qh_np, k)
        t0 = time.perf_counter()
        for _ in range(args.host_io_steps):
            ix.search(qh_np, k)
        dt = time.perf_counter() - t0
        result["host_pointer_api"] = {"queries_per_sec": round(batch * args.host_io_steps / dt, 1),
                                      "ms_per_batch": round(1e3 * dt / args.host_io_steps, 4),
                                      "note": "cgv_search_f32: pageable host queries in, host results out, one batch at a time"}

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
                                            f"{cpu_t:.1f} s wall"}
        result["recall_at_10"] = hits / (nqc * k)
        result["ordered_match_rate"] = ordered / nqc
        result["score_bit_exact_rate"] = exact_scores / nqc
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
