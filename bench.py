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
ranks); `median_qps` is the same from the median step. Beside it, never as `value`: `hbm_resident_serial`
(the same serial steps with the batch already in HBM and the results left there: what the PCIe hop
costs) and `pipelined_qps` (device-resident, two batches in flight: round 1's headline).
`--force-dist` (under `torch.distributed.run --nproc-per-node 1`) runs the N > 1 code path with one rank:
the dry run of the multi-GPU bench on a single-GPU box.
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
    # C3 whole: 10M x 768 bf16 (15.4 GB: fits ONE MI355X 18 times over), batch 4096, 8 row shards - `--sharded-handle 8` on a
    # 1-GPU box (device 0 listed 8 times), `--gpus 8` on a node
    "c3": (10_000_000, 768, "bf16", "cosine", 4096, 10),
    "small": (100_000, 768, "bf16", "cosine", 1024, 10),
    "c2shard8": (125_000, 768, "bf16", "cosine", 1024, 10),    # one rank's share of C2 at 8 GPUs (fixed-cost probe)
    # C2 with HALF the batch: two of these in flight (`pipelined`) are the proxy for running one cgv_search_f32 batch as two
    # 512-query halves on two contexts (VERDICT r3 'Next' 2b) - DESIGN.md §9.1 has what it measured
    "c2half": (1_000_000, 768, "bf16", "cosine", 512, 10),
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


def source_sha16():
    """Identity of the kernel sources a PMC pass was taken on: sha256 over codegraph-rust_amd/csrc/*.{h,hip} and host/*.cpp.
    scripts/pmc_summary.py stamps it into <tag>_<workload>_pmc_main_kernel.json; `roofline.traffic` is only quoted from a
    file whose stamp equals the sources this run was built from (the field used to be a stale constant)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "codegraph-rust_amd")
    for f in sorted(glob.glob(os.path.join(base, "csrc", "*.h")) + glob.glob(os.path.join(base, "csrc", "*.hip")) +
                    glob.glob(os.path.join(base, "host", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


def bench_sharded_handle(args, m, dev):
    """ONE cgv_sharded handle (the object the Rust seam would hold) over G shards, devices i % device_count: on a 1-GPU
    box device 0 is listed G times (exchange = device copies), on a multi-GPU box the devices are distinct and the
    exchange is the in-library ncclAllGather. Serial steps (begin + end per batch) and two batches in flight."""
    n_total, dim, dtype, metric, batch, k = WORKLOADS[args.workload]
    G = args.sharded_handle
    nd = m.device_count()
    sx = m.ShardedIndex(dim, [i % nd for i in range(G)], metric=metric, dtype=dtype)
    sx.reserve(n_total)
    for c in range((n_total + CHUNK - 1) // CHUNK):
        c_lo, c_hi = c * CHUNK, min(n_total, (c + 1) * CHUNK)
        sx.add(gen_chunk(c, c_hi - c_lo, dim, dev).cpu().numpy())
    gq = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    qhost = [torch.nn.functional.normalize(torch.randn((batch, dim), generator=gq, device=dev), dim=1).cpu().numpy()
             for _ in range(4)]
    for i in range(args.warmup):
        sx.search(qhost[i % 4], k)
    t0 = time.perf_counter()
    xms = []
    for i in range(args.steps):
        sx.search(qhost[i % 4], k)
        xms.append(sx.stats()["last_exchange_ms"])
    serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    prev = sx.search_begin(qhost[0], k)
    for i in range(1, args.steps):
        nxt = sx.search_begin(qhost[i % 4], k)
        prev.wait()
        prev = nxt
    prev.wait()
    piped = time.perf_counter() - t0
    st = sx.stats()
    # parity of the merged answer, in the same line: sampled queries through the EXACT device scan of every shard + the same
    # exchange and merge (ids global across the block-cyclic map) must equal the fast path's answer bit for bit
    check = None
    if args.check_queries > 0:
        nc = min(args.check_queries, batch)
        fi, fs = sx.search(qhost[0][:nc], k)
        fb_before = sx.stats()["fallback_queries"]
        sx.set_force_exact(True)
        ei, es = sx.search(qhost[0][:nc], k)
        sx.set_force_exact(False)
        hits = sum(len(set(ei[q].tolist()) & set(fi[q].tolist())) for q in range(nc))
        check = {"anchor": "device-exact-scan of every shard + the same merge", "queries": nc,
                 "recall_at_10": hits / (nc * k),
                 "ordered_match_rate": float(np.mean([np.array_equal(ei[q], fi[q]) for q in range(nc)])),
                 "score_bit_exact_rate": float(np.mean([np.array_equal(es[q], fs[q]) for q in range(nc)])),
                 "fallback_queries_fast_path": int(fb_before)}
    print(json.dumps({
        "metric": "queries_per_sec", "value": round(batch * args.steps / serial, 1), "unit": "queries/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * serial / args.steps, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload.upper()} through ONE cgv_sharded handle: {n_total} x {dim} {dtype} {metric}, "
                               f"batch={batch}, k={k}, {G} shards on {min(G, nd)} device(s)",
                   "rows": n_total, "dim": dim, "batch": batch, "k": k, "metric": metric, "sharding": f"block-cyclic rows/{G}",
                   "step": "cgv_sharded_search_f32: host queries -> every shard -> pack -> exchange -> merge -> host results",
                   "exchange": st["exchange"]},
        "two_in_flight": {"queries_per_sec": round(batch * args.steps / piped, 1), "ms_per_batch": round(1e3 * piped / args.steps, 4),
                          "note": "cgv_sharded_search_begin_f32 of batch i + 1 before cgv_sharded_search_end of batch i"},
        "last_exchange_ms": round(float(np.median(xms)), 4), "fallback_queries": int(st["fallback_queries"]),
        "shard_rows": sx.shard_counts(), "check": check,
        "recall_at_10": check["recall_at_10"] if check else None,
        "roofline": None, "cpu_baseline": None}), flush=True)
    sx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 200 for millisecond-scale steps, fewer for the big workloads - VERDICT r3: a 20-step "
                         "timed region was 28 ms of a 20 s run)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10; 3 for the big workloads)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--pipelined-steps", type=int, default=20,
                    help="extra device-resident batches kept `--depth` in flight (0 = skip)")
    ap.add_argument("--depth", type=int, default=2, help="batches in flight of the pipelined side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--cpu-max-queries", type=int, default=128)
    ap.add_argument("--check-queries", type=int, default=32,
                    help="N > 1 (or --force-dist): queries of one merged batch that rank 0 checks against the CPU oracle over the "
                         "WHOLE corpus (recall / order / score bits in the same JSON line); corpora too big for the CPU leg are "
                         "checked against the exact device scan of every shard + merge instead (0 = off)")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed searches before the W warm-up steps until this much wall time has passed: the part clocks "
                         "up over tens of milliseconds of load (a 20-step bench started cold measured 3-5 %% slower launches "
                         "than the same binary in steady state, r03d); 0 = off")
    ap.add_argument("--sharded-handle", type=int, default=0,
                    help="G > 0: drive ONE cgv_sharded handle over G shards (devices i %% device_count) with two batches in "
                         "flight (cgv_sharded_search_begin_f32 / _end) instead of the single index; N = 1 only")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group over RCCL, pinned-batch shard search, all-gather + merge, "
                         "multi_gpu block) with whatever world size the launcher gave - with ONE rank it is the dry run of the "
                         "multi-GPU bench on a single-GPU box (start it under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="print this rank's RANK/WORLD_SIZE and exit before touching a GPU (CPU test of the self-spawn)")
    args = ap.parse_args()
    big = args.workload in ("c5shard", "c5mini", "c3", "c3shard")       # steps of 6 - 300 ms
    if args.steps is None:
        args.steps = {"c5shard": 5, "c3": 20}.get(args.workload, 50 if big else 200)
    if args.warmup is None:
        args.warmup = 3 if big else 10

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
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m = importlib.import_module("codegraph-rust_amd")
    if args.sharded_handle > 0:
        if world != 1:
            sys.exit("bench.py: --sharded-handle runs in ONE process (N = 1); the handle itself spans the devices")
        return bench_sharded_handle(args, m, dev)
    lo, hi = m.shard_range(n_total, rank, world)
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=local_rank)
    ix.reserve(hi - lo)
    ix.set_index_base(lo)
    want_cpu = (world == 1 and dist is None and rank == 0 and args.cpu_seconds > 0)
    # N > 1 (or the one-rank dry run): rank 0 checks ONE merged batch against the oracle over the WHOLE corpus, so it keeps
    # the storage values of every chunk, not only its shard's (VERDICT r3 'Next' #4: the first SCALE line must carry parity)
    want_check = (dist is not None and rank == 0 and args.check_queries > 0)
    host_chunks = []
    if n_total > 4_000_000:
        want_cpu = False   # the f32 upcast of the corpus would not fit the CPU leg's time/memory bound
    oracle_fits = n_total <= 4_000_000
    nchunks = (n_total + CHUNK - 1) // CHUNK
    ingest_s, ingest_rows = 0.0, 0
    for c in range(nchunks):
        c_lo, c_hi = c * CHUNK, min(n_total, (c + 1) * CHUNK)
        a, b = max(lo, c_lo), min(hi, c_hi)
        if a >= b:
            if want_check and oracle_fits:   # a chunk of another rank's shard: rank 0 only needs its values on the host
                host_chunks.append(storage_values(gen_chunk(c, c_hi - c_lo, dim, dev), dtype).cpu().numpy())
            continue
        if want_check and oracle_fits and (a > c_lo or b < c_hi):   # the parts of a straddling chunk that are not mine
            xf = gen_chunk(c, c_hi - c_lo, dim, dev)
            if a > c_lo:
                host_chunks.append(storage_values(xf[: a - c_lo], dtype).cpu().numpy())
            x = xf[a - c_lo: b - c_lo]
            torch.cuda.synchronize()
            ti = time.perf_counter()
            ix.add(x)
            ingest_s += time.perf_counter() - ti
            ingest_rows += b - a
            host_chunks.append(storage_values(x, dtype).cpu().numpy())
            if b < c_hi:
                host_chunks.append(storage_values(xf[b - c_lo:], dtype).cpu().numpy())
            del x, xf
            continue
        x = gen_chunk(c, c_hi - c_lo, dim, dev)[a - c_lo: b - c_lo]
        torch.cuda.synchronize()
        ti = time.perf_counter()
        ix.add(x)                      # device f32 rows -> storage dtype + norms + block bounds (synchronous)
        ingest_s += time.perf_counter() - ti
        ingest_rows += b - a
        if want_cpu or (want_check and oracle_fits):
            host_chunks.append(storage_values(x, dtype).cpu().numpy())   # rounded-then-upcast values
        del x
    gq = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    npool = 4
    qpool = [torch.nn.functional.normalize(torch.randn((batch, dim), generator=gq, device=dev), dim=1)
             for _ in range(npool)]
    qhost = [q.cpu().pin_memory() for q in qpool]           # the caller's query batches: pinned host memory
    out_i = torch.empty((batch, k), dtype=torch.int64).pin_memory()
    out_s = torch.empty((batch, k), dtype=torch.float32).pin_memory()
    searcher = m.ShardedKnn(ix, rank=rank, world=world, force_collective=args.force_dist) if dist is not None else ix
    ix.set_profiling(True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    L, C = m.cgvec.lib(), m.cgvec.C
    if dist is None:
        oi_p, os_p = C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr())

        def step(i):   # cgv_search_f32: host queries in, host results out (H2D + D2H inside)
            m.cgvec._check(L.cgv_search_f32(ix._h, C.c_void_p(qhost[i % npool].data_ptr()), batch, k, oi_p, os_p))
    else:
        exchange_ms = []
        searcher.time_exchange = True

        def step(i):   # every rank: the (replicated) pinned batch read in place over its own PCIe link by the shard search,
            #                its top-k packed on the library's stream, ONE RCCL all-gather of the packed records + the merge
            #                kernel (writes the pinned host result arrays in place) enqueued behind an event - no host join
            #                between the search and the collective, ONE synchronisation per batch (ShardedKnn.step_packed)
            searcher.step_packed(qhost[i % npool], k, out=(out_i, out_s), device=dev)
            exchange_ms.append(searcher.last_exchange_ms)

    if args.settle_ms > 0:   # steady-state clocks before anything is measured (setup, like the index build)
        ts = time.perf_counter()
        i = 0
        while True:
            step(i)
            i += 1
            go = 1e3 * (time.perf_counter() - ts) < args.settle_ms
            if dist is not None:   # every rank runs the same number of (collective-carrying) steps: rank 0 decides
                flag = torch.tensor([1 if go else 0], device=dev)
                dist.broadcast(flag, 0)
                go = bool(flag.item())
            if not go:
                break
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
    ix.set_profiling(2)       # two un-timed steps with the whole-pipeline event pair (side field device_ms_last_step)
    step(0)
    step(1)
    st = ix.stats()
    ix.set_profiling(1)
    multi = None
    if dist is not None:
        # self-proof of the N-rank run: a collective-derived rank count, every rank's dominant-launch
        # time and shard size, and the exchange time - gathered with RCCL itself, not built from WORLD_SIZE
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        mine = torch.tensor([float(np.mean([c for c in coarse_ms if c > 0] or [0.0])), float(coarse_rows),
                             float(np.mean(exchange_ms[-args.steps:])), float(hi - lo), float(local_rank)],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = [t.cpu().tolist() for t in allr]
        multi = {"rccl_ranks_seen": int(round(float(ones.item()))),
                 "per_rank_avg_launch_ms": [round(r[0], 4) for r in allr],
                 "per_rank_rows_per_launch": [int(r[1]) for r in allr],
                 "per_rank_exchange_ms": [round(r[2], 4) for r in allr],
                 "per_rank_shard_rows": [int(r[3]) for r in allr],
                 "per_rank_device": [int(r[4]) for r in allr],
                 "exchange_ms": round(max(r[2] for r in allr), 4),
                 "redo_batches": searcher.redo_batches,
                 "exchange": "records packed on the search's stream, torch.distributed all_gather_into_tensor (backend nccl = "
                             "RCCL) + merge kernel enqueued behind an event (no host join), timed with events on the stream "
                             "they run on"}

    # side measurement: the same SERIAL steps with the query batch already in HBM and the results left in HBM
    # (cgv_search_f32_dev): what the PCIe hop of the host boundary costs per batch
    resident = None
    if dist is None and args.steps > 0:
        d_i = torch.empty((batch, k), dtype=torch.int64, device=dev)
        d_s = torch.empty((batch, k), dtype=torch.float32, device=dev)
        di_p, ds_p = C.c_void_p(d_i.data_ptr()), C.c_void_p(d_s.data_ptr())

        def rstep(i):
            m.cgvec._check(L.cgv_search_f32_dev(ix._h, C.c_void_p(qpool[i % npool].data_ptr()), batch, k, di_p, ds_p))
        for i in range(max(args.warmup, 1)):
            rstep(i)
        sync_all()
        tr = time.perf_counter()
        for i in range(args.steps):
            rstep(i)
        sync_all()
        dtr = time.perf_counter() - tr
        rstep(0)
        step(0)
        same = bool(torch.equal(d_i.cpu(), out_i) and torch.equal(d_s.cpu(), out_s))
        resident = {"queries_per_sec": round(batch * args.steps / dtr, 1), "ms_per_step": round(1e3 * dtr / args.steps, 4),
                    "same_results_as_host_step": same,
                    "note": "serial batches, queries already in HBM, results left in HBM (cgv_search_f32_dev); "
                            "`value` above includes the PCIe hop of both"}

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
            traffic_note = None
            if pmc and world == 1:
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
                # correction + WRITE_SIZE; scripts/collect_profiles.sh) - only from a pass taken on THESE kernel sources
                pj = json.load(open(os.path.join(pdir, pmc[-1])))
                if pj.get("source_sha16") == source_sha16():
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/" + pmc[-1]
                else:
                    traffic_note = (f"profiles/{pmc[-1]} was collected on other kernel sources (stamp "
                                    f"{pj.get('source_sha16')} != {source_sha16()}): not quoted")
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
                         "traffic_source": traffic_src, "traffic_note": traffic_note, "avg_launch_ms": round(cms, 4),
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
                       "exchange": "RCCL all-gather of per-shard top-k + merge (see multi_gpu)" if world > 1 else "none"},
            "median_ms_per_step": round(med, 4), "median_qps": round(batch / (med * 1e-3), 1),
            "pipelined_qps": pipelined["queries_per_sec"] if pipelined else None,
            "pipelined": pipelined,
            "hbm_resident_serial": resident,
            "roofline": roof,
            "multi_gpu": multi,
            "ingest": {"gb_per_s": round(ingest_rows * dim * (4 + ESIZE[dtype]) / max(ingest_s, 1e-9) / 1e9, 1),
                       "rows": int(ingest_rows), "seconds": round(ingest_s, 4),
                       "note": "device-resident f32 rows -> storage dtype + norms + block bounds (cgv_add_f32_dev, "
                               "synchronous per 125k-row chunk); bytes = rows x dim x (4 in + storage out)"},
            "pipeline": {"device_ms_last_step": round(st["last_total_ms"], 4), "kprime": st["last_kprime"],
                         "fallback_queries": int(st["fallback_queries"]), "eps": st["last_eps"],
                         "max_observed_coarse_err": st["max_observed_err"]},
        }

    def oracle_leg(o, rs, qh, gi, gs, budget_s, max_q):
        """Time the CPU port on single-query searches (the reference's shape) and compare every answer with the device's."""
        omet = o.COSINE if metric == "cosine" else o.DOT
        cores = o.max_threads()
        for _ in range(3):                    # warm-up: thread pool, the reusable (score, index) buffer's pages
            rs.top_k(qh[0], k, omet, cores)
        t0 = time.perf_counter()
        nqc, hits, ordered, exact_scores, sc_ms, so_ms = 0, 0, 0, 0, 0.0, 0.0
        while nqc < max_q and (nqc < 4 or time.perf_counter() - t0 < budget_s):
            ri, rsc = rs.top_k(qh[nqc], k, omet, cores)
            a, b = o.last_timing()
            sc_ms += a
            so_ms += b
            hits += len(set(ri.tolist()) & set(gi[nqc].tolist()))
            ordered += int(np.array_equal(ri, gi[nqc]))
            exact_scores += int(np.array_equal(rsc, gs[nqc]))
            nqc += 1
        cpu_t = time.perf_counter() - t0
        return {"n": nqc, "seconds": cpu_t, "cores": cores, "recall": hits / (nqc * k), "ordered": ordered / nqc,
                "exact": exact_scores / nqc, "score_ms": sc_ms / nqc, "sort_ms": so_ms / nqc}

    if want_cpu:
        from oracle import oracle as o   # CPU baseline + recall checker only
        rows_host = np.concatenate(host_chunks)
        del host_chunks
        rs = o.RowSet(rows_host)
        del rows_host
        qh = storage_values(qpool[0][:args.cpu_max_queries], dtype).cpu().numpy()
        if dtype == "fp8":   # the torch expression of the storage format must be the oracle's
            assert np.array_equal(qh, o.round_trip(qpool[0][:args.cpu_max_queries].cpu().numpy(), o.FP8, fp8_codes=True))
        gi, gs = ix.search(qpool[0], k)
        gi = gi.cpu().numpy().view(np.uint64)
        gs = gs.cpu().numpy()
        leg = oracle_leg(o, rs, qh, gi, gs, args.cpu_seconds, args.cpu_max_queries)
        rs.close()
        nqc, cpu_t = leg["n"], leg["seconds"]
        result["cpu_baseline"] = {"value": round(nqc / cpu_t, 3), "unit": "queries/s", "cores": leg["cores"],
                                  "kind": "port", "numa_nodes": o.numa_nodes(),
                                  "score_ms": round(leg["score_ms"], 2), "sort_ms": round(leg["sort_ms"], 2),
                                  "sample": f"{nqc} single-query searches over the same {n_total} x {dim} corpus "
                                            f"(f32 upcast of the {dtype} values, rows separately allocated and first touched "
                                            f"by the threads that scan them), {cpu_t:.1f} s wall; C++ port of "
                                            f"parallel_top_k_search: threaded AVX2 scoring (score_ms) + a fully parallel sort "
                                            f"of all (score, index) pairs (sort_ms; libstdc++ parallel multiway mergesort "
                                            f"standing in for rayon's par_sort_unstable_by), pair buffer reused across queries"}
        result["recall_at_10"] = leg["recall"]
        result["ordered_match_rate"] = leg["ordered"]
        result["score_bit_exact_rate"] = leg["exact"]
        result["recall_sample"] = f"{nqc} of the {batch} queries of one batch"
        result["speedup_vs_cpu_baseline"] = round(result["value"] / (nqc / cpu_t), 1)
    elif rank == 0:
        result["cpu_baseline"] = None

    # N > 1 (and the one-rank dry run): the MERGED result of one batch is checked in the same line (VERDICT r3 'Next' #4)
    if dist is not None and args.check_queries > 0:
        ncheck = min(args.check_queries, batch)
        step(0)                                           # out_i / out_s: merged results of batch 0 (pinned host arrays)
        gi = out_i.numpy().view(np.uint64)[:ncheck].copy()
        gs = out_s.numpy()[:ncheck].copy()
        if oracle_fits:
            if rank == 0:
                from oracle import oracle as o   # checker only
                rows_host = np.concatenate(host_chunks)
                del host_chunks
                assert rows_host.shape[0] == n_total
                rs = o.RowSet(rows_host)
                del rows_host
                qh = storage_values(qpool[0][:ncheck], dtype).cpu().numpy()
                leg = oracle_leg(o, rs, qh, gi, gs, 1e9, ncheck)
                rs.close()
                result["recall_at_10"] = leg["recall"]
                result["ordered_match_rate"] = leg["ordered"]
                result["score_bit_exact_rate"] = leg["exact"]
                result["recall_sample"] = (f"{leg['n']} queries of one MERGED batch (all {world} rank(s)' shards, exchange + merge "
                                           f"included) against the CPU oracle over the whole {n_total}-row corpus")
                result["check"] = {"anchor": "cpu-oracle", "queries": leg["n"], "oracle_score_ms": round(leg["score_ms"], 2),
                                   "oracle_sort_ms": round(leg["sort_ms"], 2), "redo_batches": searcher.redo_batches}
        else:
            # too big for the CPU leg: every rank answers the same queries by its EXACT device scan (the proven-correct
            # fallback path, held against the oracle by the -m gpu tests), the partial results are merged through the same
            # exchange, and the fast path's merged answer must equal that, bit for bit
            ix.set_force_exact(True)
            qs = qhost[0][:ncheck].clone().pin_memory()
            li, ls = ix.search_from_pinned(qs, k)
            ei, es = searcher._exchange(li, ls, k)
            ix.set_force_exact(False)
            torch.cuda.synchronize()
            if rank == 0:
                ei = ei.cpu().numpy().view(np.uint64)
                es = es.cpu().numpy()
                hits = sum(len(set(ei[q].tolist()) & set(gi[q].tolist())) for q in range(ncheck))
                result["recall_at_10"] = hits / (ncheck * k)
                result["ordered_match_rate"] = float(np.mean([np.array_equal(ei[q], gi[q]) for q in range(ncheck)]))
                result["score_bit_exact_rate"] = float(np.mean([np.array_equal(es[q], gs[q]) for q in range(ncheck)]))
                result["recall_sample"] = (f"{ncheck} queries of one MERGED batch against the exact device scan of every shard "
                                           f"+ the same merge (the {n_total}-row corpus is beyond the CPU leg)")
                result["check"] = {"anchor": "device-exact-scan", "queries": ncheck, "redo_batches": searcher.redo_batches}

    if rank == 0:
        print(json.dumps(result), flush=True)
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
