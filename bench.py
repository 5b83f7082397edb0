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

A "step" is ONE batched search of 1024 queries against the whole corpus, corpus resident in HBM, batches strictly
serial (one at a time). `value` = batch * K / wall time of the K timed steps (max over ranks); `median_qps` is the same from
the median step. By default (`--queries hbm`) the query batch is ALREADY RESIDENT IN HBM when the timed region starts - the
measurement contract's form: throughput with the inputs in device memory (N > 1: every rank's copy of the replicated batch
in its own HBM) - and the B*k results end in pinned HOST memory inside the step (cgv_search_f32_dev on the device aliases
of the caller's pinned arrays). SURVEY.md §8(d)'s PCIe-inclusive form - the batch starts in pinned host memory, H2D inside
the step (cgv_search_f32; `value` of rounds 1-5) - is timed in the same run over the same K steps and reported beside it as
`pcie_inclusive_serial` (never as `value`); `--queries host` swaps the two. Also beside it: `hbm_resident_results_in_hbm`
(results left in HBM too), `pipelined_host` / `pipelined` (three batches in flight), `concurrent_callers`,
`coalesced_callers`, `latency`.
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
import tempfile
import time

# Hardware queues of the HIP runtime (read when the runtime starts: before torch is imported). Its default of 4 is shared by every
# stream of the process - torch's, RCCL's, the library's - and with batches in flight on ONE index two of the handle's three context
# streams then share a queue: their batches run one after the other (kernel traces: profiles/r05_batches_in_flight_traces.txt; 125 k-row
# shard, `pipelined_host` 0.259 -> 0.23 ms with 8 queues). The rank program of the N > 1 form keeps the default: its batches run on torch
# streams beside the collective's, and there 8 queues measured WORSE (one rank over RCCL, same box: 0.225 -> 0.252 ms). An
# application-level setting like HSA_ENABLE_IPC_MODE_LEGACY; the serial `value` does not depend on it; reported in config.runtime_env.
# (setdefault: an operator's own value wins.)


def _rank_program():
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 or "--force-dist" in sys.argv:
        return True
    for i, a in enumerate(sys.argv):
        v = a.split("=", 1)[1] if a.startswith("--gpus=") else (sys.argv[i + 1] if a == "--gpus" and i + 1 < len(sys.argv) else None)
        if v is not None and v.isdigit() and int(v) > 1:
            return True
    return False


if not _rank_program():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows, dim, dtype, metric, batch, k)
    # C1 = BASELINE config 1 on the device: 10k x 384 f32 (the reference's own layout), ONE query per call - the shape of the
    # trait-level call (traits.rs:14 search_similar(&self, &[f32], limit); surreal_store.rs:61-85; caller search.rs:114-117)
    "c1": (10_000, 384, "f32", "cosine", 1, 10),
    "c2": (1_000_000, 768, "bf16", "cosine", 1024, 10),
    "c4": (1_000_000, 1536, "fp16", "dot", 256, 10),
    "c3shard": (1_250_000, 768, "bf16", "cosine", 4096, 10),   # one GPU's share of C3 (10M rows / 8)
    # C3 whole: 10M x 768 bf16 (15.4 GB: fits ONE MI355X 18 times over), batch 4096, 8 row shards - `--sharded-handle 8` on a
    # 1-GPU box (device 0 listed 8 times), `--gpus 8` on a node
    "c3": (10_000_000, 768, "bf16", "cosine", 4096, 10),
    "small": (100_000, 768, "bf16", "cosine", 1024, 10),
    # 2.4 chunks of 125 k rows: with 2 / 8 ranks the shard bookkeeping meets whole chunks of other ranks' shards AND straddling
    # chunks (the dry run of the multi-rank program, tests/test_gpu_bench_dist.py)
    "mid": (300_000, 768, "bf16", "cosine", 1024, 10),
    "c2shard8": (125_000, 768, "bf16", "cosine", 1024, 10),    # one rank's share of C2 at 8 GPUs (fixed-cost probe)
    # C2 with HALF the batch: two of these in flight (`pipelined`) are the proxy for running one cgv_search_f32 batch as two
    # 512-query halves on two contexts (VERDICT r3 'Next' 2b) - HISTORY.md §9.1 has what it measured
    "c2half": (1_000_000, 768, "bf16", "cosine", 512, 10),
    # C5 = 500M x 768 fp8 over 8 GPUs, batch 8192: one GPU's share is 62.5M rows = 48 GB of codes
    "c5shard": (62_500_000, 768, "fp8", "cosine", 8192, 10),
    "c5mini": (4_000_000, 768, "fp8", "cosine", 8192, 10),   # same kernel shape, 1/16 of the shard
    # C2's shape on UNROUNDED f32 rows (results = the reference's own f32 arithmetic): f32 + bf16 shadow
    "c2f32": (1_000_000, 768, "f32s", "cosine", 1024, 10),
}
CHUNK = 125_000
# dense MFMA peaks (MI355X_MICROARCH.md); the fp8 path runs on the block-scaled K=64 MFMA (5 PF class)
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp8": 5000.0, "f32s": 2500.0, "f32": 2500.0}
PEAK_HBM_GBS = 8000.0   # HBM3E, MI355X_MICROARCH.md
ESIZE = {"bf16": 2, "fp16": 2, "fp8": 1, "f32s": 2, "f32": 4}   # bytes per element the coarse kernel (f32: the exact scan) streams
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
    if dtype in ("f32s", "f32"):
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


def load_callers_lib(m):
    """tests/c_client/callers.c as a shared library: T NATIVE caller threads, each in a serial loop of small cgv_search_f32 calls
    (a Rust host's spawn_blocking threads; Python threads would serialise on the interpreter lock between their calls).
    Measurement infrastructure: built by __graft_entry__.build(), or here on demand."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "c_client", "callers.c")
    so = os.path.join(ROOT, "tests", "c_client", "libcgv_callers.so")
    libdir = os.path.dirname(m.cgvec.LIB_PATH)
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"),
                               src, "-o", so, "-L", libdir, "-lcgvec_hip", "-Wl,-rpath,$ORIGIN/../../codegraph-rust_amd/lib"])
    m.cgvec.lib()
    L = C.CDLL(so)
    L.cgv_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_char_p]
    L.cgv_callers_run.restype = C.c_int
    return L


def run_native_callers(CL, ix, q, k, threads, calls, warm=3, nq_per_call=1):
    """-> (ids [threads*calls, nq_per_call, k], scores, per-call latency in us, wall seconds)"""
    import ctypes as C
    ncalls = threads * calls
    oi = np.empty((ncalls, nq_per_call, k), dtype=np.uint64)
    os_ = np.empty((ncalls, nq_per_call, k), dtype=np.float32)
    lat = np.zeros(ncalls, dtype=np.float64)
    wall = C.c_double(0)
    err = C.create_string_buffer(256)
    rc = CL.cgv_callers_run(ix._h, q.ctypes.data, q.shape[0], q.shape[1], k, threads, calls, warm, nq_per_call, oi.ctypes.data,
                            os_.ctypes.data, lat.ctypes.data, C.byref(wall), err)
    if rc:
        raise RuntimeError(f"cgv_callers_run: status {rc}: {err.value.decode(errors='replace')}")
    return oi, os_, lat, wall.value


RNG = "counter"   # --rng: "counter" = SURVEY.md section 8(d)'s generator; "torch" = rounds 1-5's torch.randn chunks


def gen_chunk(c, rows, dim, device):
    """Rows [c * CHUNK, c * CHUNK + rows) of the corpus, unit-norm f32 on the device. `counter`: the counter-based stream
    keyed (seed, row, col) of SURVEY.md section 8(d) (cgv_synth_rows_f32_dev; the CPU oracle's cgo_synth_rows produces the same
    bits, tests/test_synth.py) - the same corpus whatever the chunking, the rank count or the torch / rocRAND build."""
    if RNG == "counter":
        from importlib import import_module
        return import_module("codegraph-rust_amd").cgvec.synth_rows_dev(SEED_CORPUS, c * CHUNK, rows, dim,
                                                                         device=torch.device(device).index or 0)
    g = torch.Generator(device=device).manual_seed(SEED_CORPUS + c)
    x = torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)
    return torch.nn.functional.normalize(x, dim=1)


def gen_query_pool(npool, batch, dim, device):
    """npool query batches of `batch` unit-norm f32 rows on the device (stream SEED_QUERY; batch p = rows [p * batch, ...))."""
    if RNG == "counter":
        from importlib import import_module
        cg = import_module("codegraph-rust_amd").cgvec
        return [cg.synth_rows_dev(SEED_QUERY, p * batch, batch, dim, device=torch.device(device).index or 0) for p in range(npool)]
    gq = torch.Generator(device=device).manual_seed(SEED_QUERY)
    return [torch.nn.functional.normalize(torch.randn((batch, dim), generator=gq, device=device), dim=1) for _ in range(npool)]


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


_REAL_STDOUT = None


def protect_stdout():
    """Multi-rank runs: RCCL prints a version banner on the process's stdout (file descriptor 1) when the first communicator
    forms - in front of the ONE JSON line the caller parses. Everything written to descriptor 1 from here on goes to stderr; the
    line itself is written to a saved duplicate of the original stdout (emit)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


class Watchdog:
    """No-progress limit for a multi-rank run: every phase of the rank program calls kick(stage); when nothing has been kicked
    for `limit` seconds (a collective that never completes: a rank died, a communicator never formed) the process prints ONE
    JSON line with an `error` field (rank 0; the other ranks write to stderr and fire a little later so that rank 0's line gets
    out before the launcher tears the job down) and exits - instead of hanging until the driver's own clock kills it."""

    def __init__(self, limit_s, rank, world):
        import threading
        self.limit, self.rank, self.world = float(limit_s), rank, world
        self.stage, self.last, self.on = "start", time.monotonic(), limit_s > 0
        # every rank leaves its last completed stage in a file of its own (one node: a shared /tmp), so that the error line of a
        # hung run names the stage EVERY rank reached - the ranks that hang in a collective cannot be asked through one
        self.dir = os.path.join(tempfile.gettempdir(), f"cgv_bench_{os.environ.get('MASTER_PORT', 'solo')}_{os.getuid()}")
        if self.on:
            os.makedirs(self.dir, exist_ok=True)
            self._write()
            threading.Thread(target=self._run, daemon=True).start()

    def _write(self):
        try:
            with open(os.path.join(self.dir, f"rank{self.rank}.stage"), "w") as f:
                f.write(f"{self.stage}\t{time.time():.3f}")
        except OSError:
            pass

    def kick(self, stage):
        changed = stage != self.stage
        self.stage, self.last = stage, time.monotonic()
        if self.on and changed:
            self._write()

    def stages_of_all_ranks(self):
        out = []
        for r in range(self.world):
            try:
                with open(os.path.join(self.dir, f"rank{r}.stage")) as f:
                    st, ts = f.read().split("\t")
                out.append({"rank": r, "last_stage": st, "seconds_ago": round(time.time() - float(ts), 1)})
            except (OSError, ValueError):
                out.append({"rank": r, "last_stage": None, "seconds_ago": None})
        return out

    def stop(self):
        self.on = False

    def _run(self):
        limit = self.limit + (0.0 if self.rank == 0 else 20.0)
        while self.on:
            time.sleep(0.5)
            if self.on and time.monotonic() - self.last > limit:
                msg = f"no progress for {limit:.0f} s in stage '{self.stage}' (rank {self.rank} of {self.world})"
                if self.rank == 0:
                    emit(error_line(self.world, msg, self.stage, self.stages_of_all_ranks()))
                else:
                    print(f"bench.py: {msg}", file=sys.stderr, flush=True)
                os._exit(3)


def error_line(world, msg, stage=None, per_rank=None):
    return {"metric": "queries_per_sec", "value": None, "unit": "queries/s", "n_gpus": world, "higher_is_better": True,
            "error": msg, "stage": stage, "per_rank_last_stage": per_rank}


def rank_diagnostics(dist, searcher, phases, st, dev, dev_index, ctl, world, gloo, batch, dim, k, wd):
    """What makes a first run on N GPUs readable in one shot (nobody has seen this path with more than one RCCL rank): where one
    batch of EVERY rank goes (HIP events at the phase boundaries: the library's profiling level 3 + ShardedKnn's events, on the
    stream the batch runs on), what an all-gather of the record size and of the query-batch size costs in this process group,
    which physical device every rank sits on (two ranks on one GPU would show up here) and whether peers are reachable."""
    ph = searcher.last_phase_ms or {"queries": 0.0, "search_pack": 0.0, "all_gather": 0.0, "merge": 0.0}
    lib_us = [phases["prep"], phases["first_threshold"], phases["emitting"], phases["final_publish"]]
    mine = [*lib_us, max(0.0, 1e3 * ph["search_pack"] - sum(lib_us)), 1e3 * ph["queries"], 1e3 * ph["all_gather"], 1e3 * ph["merge"],
            1e3 * st["last_total_ms"]]
    names = ["prep", "sample_tau", "emitting", "final_publish", "pack_and_gaps", "query_exchange", "all_gather", "merge",
             "search_device_total"]
    t = torch.tensor(mine, dtype=torch.float64, device=ctl)
    allr = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allr, t)
    allr = [x.cpu().tolist() for x in allr]
    per_phase = {n: {"max": round(max(r[i] for r in allr), 1), "min": round(min(r[i] for r in allr), 1),
                     "per_rank": [round(r[i], 1) for r in allr]} for i, n in enumerate(names)}
    wd.kick("phase breakdown gathered")

    def gather_us(words, iters=50):   # all_gather_into_tensor of `words` int32 per rank, HIP events on the current stream
        if gloo:
            src = torch.zeros(words, dtype=torch.int32)
            dst = [torch.empty_like(src) for _ in range(world)]
            for _ in range(3):
                dist.all_gather(dst, src)
            t0 = time.perf_counter()
            for _ in range(iters):
                dist.all_gather(dst, src)
            return 1e6 * (time.perf_counter() - t0) / iters
        src = torch.zeros(words, dtype=torch.int32, device=dev)
        dst = torch.empty(world * words, dtype=torch.int32, device=dev)
        for _ in range(5):
            dist.all_gather_into_tensor(dst, src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            dist.all_gather_into_tensor(dst, src)
        e1.record()
        e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / iters
    from importlib import import_module
    rec_words = batch * import_module("codegraph-rust_amd").sharded.packed_width(k)
    ag_rec = gather_us(rec_words)
    ag_q = gather_us((batch + world - 1) // world * dim)
    wd.kick("all-gather micro-latency measured")
    props = torch.cuda.get_device_properties(dev)
    ident = {"rank": dist.get_rank(), "device_index": dev_index, "uuid": str(getattr(props, "uuid", "")),
             "pci_bus_id": int(getattr(props, "pci_bus_id", -1)), "pci_device_id": int(getattr(props, "pci_device_id", -1)),
             "name": props.name, "visible_devices": torch.cuda.device_count(),
             "can_access_peer": [bool(torch.cuda.can_device_access_peer(dev_index, j)) for j in range(torch.cuda.device_count())
                                 if j != dev_index]}
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    keys = [(i["uuid"], i["pci_bus_id"], i["pci_device_id"]) for i in idents]
    return {"per_rank_phases_us": per_phase,
            "all_gather_latency_us": {"packed_records": round(ag_rec, 1), "packed_records_bytes_per_rank": rec_words * 4,
                                      "query_slices": round(ag_q, 1), "query_slice_bytes_per_rank": (batch + world - 1) // world * dim * 4,
                                      "note": "all_gather_into_tensor in this process group, 50 back to back on one stream, HIP events "
                                              "(gloo dry run: host wall time)"},
            "per_rank_device_identity": idents,
            "ranks_on_distinct_devices": len(set(keys)) == world}


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
    qhost = [q.cpu().numpy() for q in gen_query_pool(4, batch, dim, dev)]
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
                   "rows": n_total, "dim": dim, "batch": batch, "k": k, "metric": metric, "sharding": f"block-cyclic rows/{G}", "rng": RNG,
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
    ap.add_argument("--rng", default="counter", choices=("counter", "torch"),
                    help="synthetic inputs: `counter` = SURVEY.md 8(d)'s counter-based generator keyed (seed, row, col), identical "
                         "on the device and in the CPU oracle; `torch` = torch.randn per 125 k-row chunk (rounds 1-5)")
    ap.add_argument("--pipelined-steps", type=int, default=None,
                    help="batches of each pipelined side measurement (default: --steps; 0 = skip): `pipelined_host` = the SAME "
                         "work as a step (pinned host batch in, host results out, exchange + merge included when N > 1) with "
                         "`--depth` batches in flight; N = 1 also `pipelined` = device-resident batches, the round-1 headline")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight of the pipelined side measurements (<= 3)")
    ap.add_argument("--callers", type=int, default=3,
                    help="N = 1 side measurement `concurrent_callers`: this many host THREADS, each issuing serial cgv_search_f32 calls on its "
                         "own pinned buffers against the one index (the reference's threading model: a Send + Sync store called from a "
                         "multi-thread runtime); 0 = skip")
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
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend of an N > 1 (or --force-dist) run. nccl = RCCL, one rank per GPU (the real thing). "
                         "gloo = DRY RUN of the same rank program on fewer GPUs than ranks (rank r uses device r %% device_count; "
                         "RCCL refuses two ranks on one device): every rank owns a real device shard, packs and merges on the "
                         "device, the packed records and the control collectives travel over gloo through the host - the whole "
                         "world > 1 control flow of this file on a 1-GPU box; its numbers are not a scaling measurement")
    ap.add_argument("--query-exchange", default="auto", choices=("auto", "replicated", "sharded"),
                    help="N > 1: how the host query batch reaches the ranks - every rank reads all of it over its own PCIe link "
                         "(replicated), or each rank moves 1/N of it and one all-gather of the f32 slices over xGMI completes it "
                         "(sharded; ShardedKnn.query_exchange). auto = both forms are timed in a short trial before the warm-up "
                         "(multi_gpu.query_exchange) and the faster one runs the timed steps")
    ap.add_argument("--queries", default=None, choices=("hbm", "host"),
                    help="where a timed step's query batch starts. hbm (default for batched workloads) = already resident in device memory when the timed "
                         "region starts (the measurement contract: `value` = throughput with the inputs in HBM); the B*k results "
                         "still end in pinned HOST memory inside the step. host = the batch starts in pinned host memory and "
                         "crosses PCIe inside the step (SURVEY.md 8(d)'s form, `value` of rounds 1-5); whichever form is not "
                         "`value` is timed as well and reported beside it (`pcie_inclusive_serial` / `hbm_resident_serial`). "
                         "Single-query workloads (c1: the trait-level call hands over ONE host slice, traits.rs:14) default to host")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="seconds: process-group timeout AND the no-progress limit of the watchdog - a hung collective ends the "
                         "run with a JSON line carrying an `error` field instead of hanging the launcher")
    ap.add_argument("--coalesced-threads", type=int, default=64,
                    help="N = 1 side measurement `coalesced_callers`: this many NATIVE host threads (tests/c_client/callers.c), each in a "
                         "serial loop of SINGLE-query cgv_search_f32 calls on the one index - the reference's trait-level call shape "
                         "(traits.rs:14; search.rs:358-361 issues B of them concurrently); the library merges concurrent callers into "
                         "shared device batches (csrc/coalesce.h). 0 = skip")
    ap.add_argument("--coalesced-calls", type=int, default=0, help="calls per thread of `coalesced_callers` (0 = sized for ~0.3 s)")
    ap.add_argument("--latency-tail", type=int, default=20000,
                    help="N = 1: single-query calls (distinct queries, one native thread) behind latency.nq1_tail's p50 / p99 / p99.9 / "
                         "p99.99 (a twentieth of it on corpora beyond 2M rows); 0 = skip")
    ap.add_argument("--latency", type=int, default=1,
                    help="N = 1: side fields with the median latency of nq = 1 / 8 / 32 searches through cgv_search_f32 (pageable "
                         "and pinned buffers) on this workload's index (0 = skip)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="print this rank's RANK/WORLD_SIZE and exit before touching a GPU (CPU test of the self-spawn)")
    args = ap.parse_args()
    global RNG
    RNG = args.rng
    big = args.workload in ("c5shard", "c5mini", "c3", "c3shard")       # steps of 6 - 300 ms
    if args.steps is None:
        # (single-query workloads: 2000 steps of ~50 us - the first device-wide synchronisation of a process is followed by
        #  a few milliseconds in which small calls take 60-70 us instead of 47, scripts/c1_probe.py: 200 steps sat inside them)
        args.steps = {"c5shard": 5, "c3": 20}.get(args.workload, 50 if big else (2000 if WORKLOADS[args.workload][4] < 64 else 200))
    if args.warmup is None:
        args.warmup = 3 if big else 10
    if args.pipelined_steps is None:
        args.pipelined_steps = min(args.steps, 200)
    if args.queries is None:
        args.queries = "hbm" if WORKLOADS[args.workload][4] >= 64 else "host"

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
    if os.environ.get("BENCH_TEST_DROP_RANK") == str(rank) and world > 1:
        sys.exit(7)   # (tests: a rank that never joins the process group)
    wd = Watchdog(args.dist_timeout if (world > 1 or args.force_dist) else 0.0, rank, world)
    try:
        return run(args, wd, world, rank, local_rank)
    except SystemExit:
        raise
    except BaseException as e:   # noqa: BLE001 - a failed rank must leave a parseable line, not only a traceback
        import traceback
        traceback.print_exc()
        if rank == 0:
            emit(error_line(world, f"{type(e).__name__}: {e}", wd.stage, wd.stages_of_all_ranks() if wd.on else None))
        wd.stop()
        os._exit(1)   # (not sys.exit: a broken process group may block interpreter shutdown in its destructors)


def run(args, wd, world, rank, local_rank):
    n_total, dim, dtype, metric, batch, k = WORKLOADS[args.workload]
    dist = None
    gloo = False
    dev_index = local_rank
    if world > 1 or args.force_dist:
        import datetime
        import torch.distributed as dist
        protect_stdout()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        gloo = args.dist_backend == "gloo"
        if gloo:   # dry run: more ranks than GPUs - rank r on device r % device_count, records + control over gloo
            dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        wd.kick("init_process_group")
        # (the process group's own timeout stays generous: rank 0's CPU-oracle leg runs while the others sit in the last barrier;
        # hangs in the measured phases are the watchdog's business)
        tmo = datetime.timedelta(seconds=max(600.0, args.dist_timeout))
        if gloo:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctl = torch.device("cpu") if gloo else dev      # where the control collectives' tensors live
    wd.kick("import")

    m = importlib.import_module("codegraph-rust_amd")
    if args.sharded_handle > 0:
        if world != 1:
            sys.exit("bench.py: --sharded-handle runs in ONE process (N = 1); the handle itself spans the devices")
        return bench_sharded_handle(args, m, dev)
    lo, hi = m.shard_range(n_total, rank, world)
    ix = m.HipKnnIndex(dim, metric=metric, dtype=dtype, device=dev_index)
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
    wd.kick("corpus loaded")
    npool = 4 if batch >= 64 else 128      # (single-query workloads: enough distinct queries for the CPU leg and the medians)
    qpool = gen_query_pool(npool, batch, dim, dev)
    qhost = [q.cpu().pin_memory() for q in qpool]           # the caller's query batches: pinned host memory
    out_i = torch.empty((batch, k), dtype=torch.int64).pin_memory()
    out_s = torch.empty((batch, k), dtype=torch.float32).pin_memory()
    searcher = m.ShardedKnn(ix, rank=rank, world=world, force_collective=args.force_dist) if dist is not None else ix
    ix.set_profiling(1)       # the dominant launch carries its start / end event pair (avg_launch_ms); 2 = + whole-pipeline events

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    L, C = m.cgvec.lib(), m.cgvec.C
    hbm = args.queries == "hbm"
    if dist is None:
        oi_p, os_p = C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr())
        oi_a, os_a = C.c_void_p(ix.device_alias(out_i)), C.c_void_p(ix.device_alias(out_s))
        ix.use_own_stream()

        qh_p = [C.c_void_p(q.data_ptr()) for q in qhost]     # (the step is the library call: no interpreter work beside it)
        qd_p = [C.c_void_p(q.data_ptr()) for q in qpool]
        chk, s_host, s_dev, hh = m.cgvec._check, L.cgv_search_f32, L.cgv_search_f32_dev, ix._h

        def host_step(i):   # cgv_search_f32: host queries in, host results out (H2D + D2H inside)
            chk(s_host(hh, qh_p[i % npool], batch, k, oi_p, os_p))

        def hbm_step(i):    # cgv_search_f32_dev: the batch is in HBM already; the last kernel writes the caller's pinned host arrays
            chk(s_dev(hh, qd_p[i % npool], batch, k, oi_a, os_a))
    else:
        exchange_ms = []
        # (the all-gather + merge are timed with stream events in the two diagnostic steps behind the timed region, not in it: an
        #  event pair is two more packets on the batch's stream - a few microseconds of a 0.3 ms step on an 8-GPU shard)

        def host_step(i):   # every rank: the (replicated) pinned batch read in place over its own PCIe link by the shard search,
            #                its top-k packed behind the search's last kernel, ONE RCCL all-gather of the packed records + the
            #                merge kernel (writes the pinned host result arrays in place) in line on the batch's stream - no host
            #                join between the search and the collective, ONE synchronisation per batch (ShardedKnn.step_packed)
            searcher.step_packed(qhost[i % npool], k, out=(out_i, out_s), device=dev)
            exchange_ms.append(searcher.last_exchange_ms)

        def hbm_step(i):    # the same with every rank's copy of the batch resident in ITS HBM (queries are replicated:
            #                SURVEY.md 8(e)); results still end in the pinned host arrays
            searcher.step_packed(qpool[i % npool], k, out=(out_i, out_s), device=dev)
            exchange_ms.append(searcher.last_exchange_ms)
    step = hbm_step if hbm else host_step
    other_step = host_step if hbm else hbm_step

    # The interpreter's cyclic garbage collector stays out of the timed regions: with torch imported a full collection walks
    # ~10^6 objects (tens of milliseconds) - one of them inside a 200-step region of 1.4 ms steps showed up as a mean 12 % above
    # the median (r05 first run: 1.532 vs 1.362 ms; round 4's runs happened not to catch one). Everything allocated so far is
    # frozen out of the collector's sight; it runs again, explicitly, between the measurements. It is done HERE, in front of the
    # settle loop and the warm-up, not between the warm-up and the timed steps: the collection idles the GPU for tens of
    # milliseconds, the clocks drop, and the first ~20 steps behind it run 5-20 % slow - invisible in 200 steps, but a driver run
    # of `--steps 20` measured 696 k q/s where 200 steps gave 749 k on the same box.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    if args.settle_ms > 0:   # steady-state clocks before anything is measured (setup, like the index build)
        ts = time.perf_counter()
        i = 0
        while True:
            step(i)
            i += 1
            go = 1e3 * (time.perf_counter() - ts) < args.settle_ms
            if dist is not None:   # every rank runs the same number of (collective-carrying) steps: rank 0 decides
                flag = torch.tensor([1 if go else 0], device=ctl)
                dist.broadcast(flag, 0)
                go = bool(flag.item())
            wd.kick("settle")
            if not go:
                break
    query_exchange = None
    if dist is not None:
        # How the host batch reaches the ranks: both forms timed in the same process group, same clocks (setup, like the settle
        # loop); max_over_ranks gives every rank the same two numbers, so every rank picks the same form
        def trial(mode, nsteps=30):
            searcher.query_exchange = mode
            for i in range(3):
                host_step(i)
            sync_all()
            tq = time.perf_counter()
            for i in range(nsteps):
                host_step(i)
            sync_all()
            wd.kick(f"query-exchange trial: {mode}")
            return 1e3 * max_over_ranks(time.perf_counter() - tq) / nsteps
        rep_ms = trial("replicated")
        sha_ms = trial("sharded")
        # (auto: the sharded form must win by 3 % - one rank pays all of its cost and gets none of its saving, and a difference
        # inside the noise should not flip the form from run to run)
        chosen = args.query_exchange if args.query_exchange != "auto" else ("sharded" if sha_ms < 0.97 * rep_ms else "replicated")
        searcher.query_exchange = chosen
        query_exchange = {"replicated_ms": round(rep_ms, 4), "sharded_ms": round(sha_ms, 4),
                          "timed_steps_use": "n/a (the timed steps' batch is resident in every rank's HBM)" if hbm else chosen,
                          "host_batch_steps_use": chosen,
                          "selection": args.query_exchange + (" (sharded when >= 3 % faster)" if args.query_exchange == "auto" else ""),
                          "note": "ms per serial step, 30 steps each, max over ranks; replicated = every rank reads the whole pinned "
                                  "batch over its own PCIe link; sharded = each rank copies batch/N rows to its device and ONE "
                                  "all-gather of the f32 slices over xGMI completes the batch in HBM (identical results)"}
    for i in range(args.warmup):
        step(i)
    wd.kick("warm-up done")
    sync_all()
    coarse_ms, coarse_rows, step_ms = [], 0, []
    # (the launch time of each batch's dominant kernel: the raw cgv_get_stats call into one preallocated struct, ~1 us - ix.stats()
    #  builds a 13-field dict per call, ~12 us of interpreter time inside every timed step: 1 % of a C2 step, a quarter of a C1 one)
    raw_stats, get_stats, h_ix = m.cgvec.Stats(), L.cgv_get_stats, ix._h
    raw_ref = C.byref(raw_stats)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(i)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        get_stats(h_ix, raw_ref)   # host-side read of that batch's HIP-event pair; no extra device sync
        coarse_ms.append(raw_stats.last_coarse_ms)
    t_loop = time.perf_counter() - t0
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if os.environ.get("BENCH_DEBUG_STEPS"):
        print(f"timed loop {1e3 * t_loop:.3f} ms, closing synchronisation {1e3 * (elapsed - t_loop):.3f} ms", file=sys.stderr, flush=True)
    coarse_rows = int(raw_stats.coarse_rows)
    gc.collect()
    if os.environ.get("BENCH_DEBUG_STEPS"):   # (diagnostics: where in the timed region the slow steps sit)
        print("step_ms:", " ".join(f"{x:.3f}" for x in step_ms), file=sys.stderr, flush=True)
    wd.kick("timed steps done")
    ix.set_profiling(3)       # two un-timed steps with the whole-pipeline event pair (side field device_ms_last_step) and the phase events
    if dist is not None:
        searcher.time_phases = True
    if dist is not None:
        del exchange_ms[:]
    step(0)
    step(1)
    st = ix.stats()
    phases = ix.phase_times_us()
    ix.set_profiling(1)
    if dist is not None:
        searcher.time_phases = False
        exchange_steps = list(exchange_ms)   # all-gather + merge of the two diagnostic steps (events on the batch's stream)
    multi = None
    if dist is not None:
        diag = rank_diagnostics(dist, searcher, phases, st, dev, dev_index, ctl, world, gloo, batch, dim, k, wd)
        # self-proof of the N-rank run: a collective-derived rank count, every rank's dominant-launch
        # time and shard size, and the exchange time - gathered with RCCL itself, not built from WORLD_SIZE
        ones = torch.ones(1, device=ctl)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        mine = torch.tensor([float(np.mean([c for c in coarse_ms if c > 0] or [0.0])), float(coarse_rows),
                             float(np.mean(exchange_steps or [0.0])), float(hi - lo), float(dev_index)],
                            dtype=torch.float64, device=ctl)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = [t.cpu().tolist() for t in allr]
        wd.kick("multi_gpu block gathered")
        multi = {"backend": "gloo (DRY RUN: records and control through the host; ranks share devices)" if gloo else "nccl (RCCL)",
                 "rccl_ranks_seen": int(round(float(ones.item()))),
                 "per_rank_avg_launch_ms": [round(r[0], 4) for r in allr],
                 "per_rank_rows_per_launch": [int(r[1]) for r in allr],
                 "per_rank_exchange_ms": [round(r[2], 4) for r in allr],
                 "per_rank_shard_rows": [int(r[3]) for r in allr],
                 "per_rank_device": [int(r[4]) for r in allr],
                 "exchange_ms": round(max(r[2] for r in allr), 4),
                 "redo_batches": searcher.redo_batches,
                 "query_exchange": query_exchange,
                 **diag,
                 "exchange": ("records packed behind the search's last kernel; DRY RUN: D2H, gloo all_gather, H2D, merge kernel "
                              "(host join in the middle - not timed)") if gloo else
                             ("records packed behind the search's last kernel, torch.distributed all_gather_into_tensor (backend "
                              "nccl = RCCL) + merge kernel in line on the batch's stream (no host join), timed with events on "
                              "the stream they run on")}

    # side measurement: the OTHER form of the serial step (--queries): with `value` on HBM-resident batches this is SURVEY.md 8(d)'s
    # PCIe-inclusive form (pinned host batch in, host results out: `value` of rounds 1-5), and the other way round
    other_serial = None
    if args.steps > 0:
        for i in range(max(args.warmup, 1)):
            other_step(i)
        sync_all()
        tr = time.perf_counter()
        for i in range(args.steps):
            other_step(i)
        sync_all()
        dto = max_over_ranks(time.perf_counter() - tr)
        other_step(0)
        oi2, os2 = out_i.clone(), out_s.clone()
        step(0)
        other_serial = {"queries_per_sec": round(batch * args.steps / dto, 1), "ms_per_step": round(1e3 * dto / args.steps, 4),
                        "same_results_as_value_step": bool(torch.equal(oi2, out_i) and torch.equal(os2, out_s)),
                        "note": ("serial batches that START IN PINNED HOST MEMORY and cross PCIe inside the step (cgv_search_f32; N > 1: "
                                 "the query exchange form multi_gpu.query_exchange.host_batch_steps_use), results to host memory - "
                                 "SURVEY.md 8(d)'s form, `value` of rounds 1-5") if hbm else
                                ("serial batches already resident in HBM when the step starts (cgv_search_f32_dev), results written to "
                                 "pinned host memory")}
        wd.kick("other serial form timed")

    # side measurement: serial steps with the query batch in HBM AND the results left in HBM (cgv_search_f32_dev on device
    # arrays): what delivering the B*k results to the host costs per batch
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
                    "same_results_as_value_step": same,
                    "note": "serial batches, queries already in HBM, results LEFT in HBM (cgv_search_f32_dev on device arrays)"}

    # side measurement: THE SAME WORK AS A STEP - pinned host batch in, host results out, and (N > 1) the exchange + merge of
    # every batch - with `depth` batches in flight: each batch's whole pipeline is enqueued on its own stream, the host only
    # waits for the oldest one (VERDICT r4 'Next' 1). N = 1: cgv_search_begin_f32_dev on the buffers' device aliases /
    # cgv_search_end; N > 1: ShardedKnn.step_packed_begin / _end (join-free: search -> pack -> all-gather -> merge in line).
    depth = max(1, min(args.depth, ix.max_in_flight))
    pipelined_host = None
    if args.pipelined_steps > 0:
        outs = [(torch.empty((batch, k), dtype=torch.int64).pin_memory(), torch.empty((batch, k), dtype=torch.float32).pin_memory())
                for _ in range(depth)]
        if dist is None:
            def hbegin(i):
                return ix.search_begin_pinned(qhost[i % npool], k, outs[i % depth])

            def hend(p):
                p.wait()
        else:
            def hbegin(i):
                return searcher.step_packed_begin(qhost[i % npool], k, out=outs[i % depth], device=dev)

            def hend(p):
                searcher.step_packed_end(p)

        piped_ends = []

        def run_piped(nsteps):
            pend = collections.deque()
            for i in range(nsteps):
                pend.append(hbegin(i))
                if len(pend) >= depth:
                    hend(pend.popleft())
                    piped_ends.append(time.perf_counter())
            while pend:
                hend(pend.popleft())
                piped_ends.append(time.perf_counter())
        gc.collect()     # (in front of the warm batches: nothing idles the device between them and the timed ones)
        # Warm batches for at least 40 ms (every rank the same number: rank 0 decides): the first ~10 ms of host batches in flight
        # after a stretch of HBM-resident work run at a third of the steady rate (BENCH_DEBUG_STEPS prints the completion intervals:
        # 12 ms for the first batch, then 1.2; the copy engine / PCIe path wakes up) - with 5 warm batches a 20-batch region read
        # 1.87 ms per batch where 200 batches read 1.23 (the driver-form lines of profiles/r06z*, r06zz*)
        tw = time.perf_counter()
        for _ in range(40):
            run_piped(depth)
            go = 1e3 * (time.perf_counter() - tw) < 40.0
            if dist is not None:
                flag = torch.tensor([1 if go else 0], device=ctl)
                dist.broadcast(flag, 0)
                go = bool(flag.item())
            if not go:
                break
        wd.kick("pipelined_host warm")
        redo_before = searcher.redo_batches if dist is not None else 0
        sync_all()
        tp = time.perf_counter()
        run_piped(args.pipelined_steps)
        sync_all()
        dtp = max_over_ranks(time.perf_counter() - tp)
        if os.environ.get("BENCH_DEBUG_STEPS"):   # (diagnostics: completion-to-completion intervals of the timed batches)
            ends = piped_ends[-args.pipelined_steps:]
            print("pipelined_host batch intervals ms:", " ".join(f"{1e3 * (b_ - a_):.2f}" for a_, b_ in zip([tp] + ends[:-1], ends)), file=sys.stderr, flush=True)
        gc.collect()
        wd.kick("pipelined_host timed")
        # parity of the pipelined batches: the last `depth` batches' host results against a SERIAL step on the same queries
        same = True
        for j in range(depth):
            i = args.pipelined_steps - 1 - j
            if i < 0:
                break
            pi, ps = outs[i % depth][0].clone(), outs[i % depth][1].clone()
            step(i)
            same = same and bool(torch.equal(pi, out_i) and torch.equal(ps, out_s))
        pipelined_host = {"queries_per_sec": round(batch * args.pipelined_steps / dtp, 1),
                          "ms_per_batch": round(1e3 * dtp / args.pipelined_steps, 4), "batches_in_flight": depth,
                          "batches": args.pipelined_steps, "same_results_as_serial_step": same,
                          "redo_batches": (searcher.redo_batches - redo_before) if dist is not None else None,
                          "note": ("the same work as a host-batch serial step (`pcie_inclusive_serial`) - pinned host batch in (fetched by the copy engine while the batches before "
                                   "it compute), shard search, packed records, all-gather, merge into pinned host arrays - with batches in "
                                   "flight on their own streams (ShardedKnn.step_packed_begin / _end); the host waits for the oldest "
                                   "batch only")
                          if dist is not None else
                          ("the same work as a host-batch serial step (`pcie_inclusive_serial`) - pinned host batch in (fetched by the copy engine while the batches before it "
                           "compute), host results written in place - with batches in flight (cgv_search_begin_f32_dev on the "
                           "buffers' device aliases / cgv_search_end)")}

    # side measurement: device-resident queries and results, `depth` batches in flight (round 1's headline); N = 1 only
    pipelined = None
    if args.pipelined_steps > 0 and dist is None:
        pend = collections.deque()
        sync_all()
        tp = time.perf_counter()
        for i in range(args.pipelined_steps):
            pend.append(ix.search_begin(qpool[i % npool], k))
            if len(pend) >= depth:
                pend.popleft().wait()
        while pend:
            pend.popleft().wait()
        sync_all()
        dtp = time.perf_counter() - tp
        pipelined = {"queries_per_sec": round(batch * args.pipelined_steps / dtp, 1),
                     "ms_per_batch": round(1e3 * dtp / args.pipelined_steps, 4), "batches_in_flight": depth,
                     "note": "queries and results stay in HBM (cgv_search_begin_f32_dev / cgv_search_end)"}
    # side measurement: several host threads, each in its own serial cgv_search_f32 loop on the ONE index - what a server built on
    # the reference's traits does (VectorStore is Send + Sync, called from a multi-thread tokio runtime through spawn_blocking;
    # SURVEY.md section 8(b) threading): the handle's three search contexts overlap the callers' batches on the device
    concurrent = None
    if dist is None and args.callers > 1 and args.pipelined_steps > 0:
        import threading
        T = min(args.callers, ix.max_in_flight)
        per = max(10, args.pipelined_steps // T)
        tb = [(qhost[t % npool], torch.empty((batch, k), dtype=torch.int64).pin_memory(),
               torch.empty((batch, k), dtype=torch.float32).pin_memory()) for t in range(T)]
        gate = threading.Barrier(T + 1)
        errs = []

        def caller(t):
            q_, oi_, os__ = tb[t]
            try:
                for _ in range(3):
                    ix.search_host_ptr(q_.data_ptr(), batch, k, oi_.data_ptr(), os__.data_ptr())
                gate.wait()
                for _ in range(per):
                    ix.search_host_ptr(q_.data_ptr(), batch, k, oi_.data_ptr(), os__.data_ptr())
            except Exception as e:   # noqa: BLE001 - reported in the line
                errs.append(f"{type(e).__name__}: {e}")
                try:
                    gate.abort()
                except Exception:   # noqa: BLE001
                    pass

        ths = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        for th in ths:
            th.start()
        try:
            gate.wait()
        except threading.BrokenBarrierError:
            pass
        tp = time.perf_counter()
        for th in ths:
            th.join()
        sync_all()
        dtc = time.perf_counter() - tp
        same_c = not errs
        for t in range(T):     # each caller's last results against a lone serial step on the same queries
            step(t % npool)
            same_c = same_c and bool(torch.equal(tb[t][1], out_i) and torch.equal(tb[t][2], out_s))
        concurrent = {"threads": T, "calls_per_thread": per, "queries_per_sec": round(batch * T * per / dtc, 1),
                      "ms_per_batch": round(1e3 * dtc / (T * per), 4), "same_results_as_serial_step": same_c,
                      "errors": errs or None,
                      "note": "host threads in serial cgv_search_f32 loops on one index (pinned host batch in, host results out: the same "
                              "work as `pcie_inclusive_serial` per call); the library fetches a caller's batch with the copy engine while other callers' "
                              "batches compute"}
        wd.kick("concurrent callers")

    # side measurement: the reference's call shape at scale - T native threads, each in a serial loop of SINGLE-query calls on
    # the one index (VectorStore::search_similar is one query, traits.rs:14; multi_vector_search issues B concurrent single-query
    # searches, search.rs:358-361). A lone call streams the whole corpus for one query column; concurrent callers are merged into
    # shared device batches of up to 64 queries by the library (csrc/coalesce.h) - `lone` is the same loop with ONE thread.
    coalesced = None
    if dist is None and args.coalesced_threads > 1 and st["last_path"] in (0, 1):
        try:
            CL = load_callers_lib(m)
            T = args.coalesced_threads
            qall = np.ascontiguousarray(torch.cat(qpool).cpu().numpy())[: max(64, min(4096, npool * batch))]
            lone_calls = 200 if n_total <= 2_000_000 else 30
            _, _, lat1, wall1 = run_native_callers(CL, ix, qall, k, 1, lone_calls)
            lone_p50 = float(np.median(lat1))
            calls = args.coalesced_calls or int(max(30, min(2000, 0.3 / max(lone_p50 * 1e-6, 1e-6) * 40.0 / T)))
            cs0 = ix.coalesce_stats()
            gc.collect()
            ci, csc, latc, wallc = run_native_callers(CL, ix, qall, k, T, calls)
            cs1 = ix.coalesce_stats()
            # every 37th call (and each thread's last one) against a LONE call of the same query (merging switched off)
            ix.set_coalesce(0, 0, 0)
            sample = sorted(set(range(0, T * calls, 37)) | {t * calls + calls - 1 for t in range(T)})[:400]
            same_cc = True
            li, ls = np.empty((1, k), dtype=np.uint64), np.empty((1, k), dtype=np.float32)
            for c_ in sample:
                ix.search_host_ptr(qall[c_ % qall.shape[0]].ctypes.data, 1, k, li.ctypes.data, ls.ctypes.data)
                same_cc = same_cc and bool(np.array_equal(li[0], ci[c_, 0]) and np.array_equal(ls[0], csc[c_, 0]))
            ix.set_coalesce()   # (back to the library's defaults)
            nb = max(1, cs1["batches"] - cs0["batches"])
            coalesced = {
                "threads": T, "calls_per_thread": calls, "nq_per_call": 1,
                "queries_per_sec": round(T * calls / wallc, 1),
                "call_us": {"p50": round(float(np.percentile(latc, 50)), 1), "p90": round(float(np.percentile(latc, 90)), 1),
                            "p99": round(float(np.percentile(latc, 99)), 1), "max": round(float(latc.max()), 1)},
                "lone": {"queries_per_sec": round(lone_calls / wall1, 1), "p50_us": round(lone_p50, 1),
                         "p99_us": round(float(np.percentile(lat1, 99)), 1), "calls": lone_calls},
                "p50_vs_lone_p50": round(float(np.percentile(latc, 50)) / lone_p50, 3),
                "speedup_vs_lone_caller": round((T * calls / wallc) / (lone_calls / wall1), 1),
                "device_batches": cs1["batches"] - cs0["batches"],
                "avg_queries_per_batch": round((cs1["batched_queries"] - cs0["batched_queries"]) / nb, 1),
                "calls_that_ran_alone": cs1["lone_calls"] - cs0["lone_calls"],
                "retried_alone": cs1["retried_alone"] - cs0["retried_alone"],
                "same_results_as_lone_calls": same_cc, "checked_calls": len(sample),
                "corpus_stream_gb_per_s": round(float(n_total) * dim * ESIZE[dtype] * nb / wallc / 1e9, 1),
                "note": "native threads (tests/c_client/callers.c), each in a serial loop of single-query cgv_search_f32 calls with "
                        "pageable buffers; concurrent callers share device batches (group commit, csrc/coalesce.h; the library's default "
                        "policy: CGV_COALESCE_* in include/cgvec.h); `lone` = the same loop with one thread; checked calls are "
                        "compared bit for bit with lone calls of the same queries"}
        except Exception as e:   # noqa: BLE001 - a side measurement never takes the line down
            coalesced = {"error": f"{type(e).__name__}: {e}"}
        wd.kick("coalesced callers")

    # side measurement: latency of SMALL calls through the host boundary - what a Rust caller swapping this backend in issues
    # from SemanticSearch::search_by_embedding (search.rs:114-117 -> surreal_store.rs:61-85 -> traits.rs:14): ONE query per call.
    # Median microseconds of cgv_search_f32 at nq = 1 / 8 / 32 with pageable buffers (a Rust Vec<f32>; what host/store.cpp hands
    # down) and with pinned ones, and the rate the corpus is streamed at (rows x ld x s / time) against the 8 TB/s HBM peak.
    latency = None
    if dist is None and args.latency > 0:
        latency = {}
        iters = 300 if n_total <= 2_000_000 else 40
        for lnq in (1, 8, 32):
            qsrc = torch.cat(qpool)[:lnq] if batch < lnq else qpool[0][:lnq]
            qpg = np.ascontiguousarray(qsrc.cpu().numpy())
            o_i, o_s = np.empty((lnq, k), dtype=np.uint64), np.empty((lnq, k), dtype=np.float32)
            qpn = qsrc.cpu().pin_memory()
            p_i, p_s = torch.empty((lnq, k), dtype=torch.int64).pin_memory(), torch.empty((lnq, k), dtype=torch.float32).pin_memory()
            row = {}
            for name, (qp_, ip_, sp_) in (("pageable", (qpg.ctypes.data, o_i.ctypes.data, o_s.ctypes.data)),
                                          ("pinned", (qpn.data_ptr(), p_i.data_ptr(), p_s.data_ptr()))):
                ts = []
                for it in range(iters + 10):
                    t1 = time.perf_counter()
                    ix.search_host_ptr(qp_, lnq, k, ip_, sp_)
                    ts.append(time.perf_counter() - t1)
                row[name + "_us"] = round(1e6 * float(np.median(ts[10:])), 1)
                row[name + "_p99_us"] = round(1e6 * float(np.percentile(ts[10:], 99)), 1)
            same = bool(np.array_equal(o_i, p_i.numpy().view(np.uint64)) and np.array_equal(o_s, p_s.numpy()))
            best = min(row["pageable_us"], row["pinned_us"]) * 1e-6
            stream_gbs = float(n_total) * dim * ESIZE[dtype] / best / 1e9
            row.update({"path": ix.stats()["last_path"], "same_results": same, "corpus_stream_gb_per_s": round(stream_gbs, 1),
                        "hbm_frac": round(stream_gbs / PEAK_HBM_GBS, 4)})
            latency[f"nq{lnq}"] = row
        if n_total <= 200_000:   # the trait surface itself: cgvs_search_similar of a VectorStore over the same rows (small corpora)
            vs = m.store.VectorStore(dtype=dtype, device=dev_index)
            import uuid
            ids = [uuid.UUID(int=i + 1) for i in range(n_total)]
            vs.store_embeddings(ids, np.concatenate([gen_chunk(c, min(CHUNK, n_total - c * CHUNK), dim, dev).cpu().numpy()
                                                     for c in range(nchunks)]))
            q1 = qpool[0][0].cpu().numpy()
            ts = []
            for it in range(iters + 10):
                t1 = time.perf_counter()
                got = vs.search_similar(q1, k)
                ts.append(time.perf_counter() - t1)
            latency["cgvs_search_similar_nq1"] = {"median_us": round(1e6 * float(np.median(ts[10:])), 1),
                                                  "p99_us": round(1e6 * float(np.percentile(ts[10:], 99)), 1), "hits": len(got)}
            vs.close()
        # the TAIL of the single-query call: many DISTINCT queries (the loops above repeat one), native caller thread. What sits
        # out there: a small batch keeps two rows per 1/1024 of the corpus, and a query with three of its best rows in one cell
        # fails its check (5e-4 per query on C2's corpus) - round 5 sent it through the exact scan of the whole corpus (+0.4 ms),
        # round 6 re-scans the offending cells only (kernels_repair.h); small_batch_stats counts both kinds.
        if args.latency_tail > 0 and st["last_path"] == 1:
            try:
                CLt = load_callers_lib(m)
                ncalls = args.latency_tail if n_total <= 2_000_000 else max(200, args.latency_tail // 20)
                if RNG == "counter":
                    qtail = m.cgvec.synth_rows_dev(SEED_QUERY, 1 << 24, ncalls, dim, device=dev_index).cpu().numpy()
                else:
                    qtail = np.ascontiguousarray(torch.cat(qpool).cpu().numpy())
                sb0 = ix.small_batch_stats()
                ix.set_coalesce(0, 0, 0)       # lone calls
                _, _, latt, _ = run_native_callers(CLt, ix, qtail, k, 1, ncalls)
                ix.set_coalesce()
                sb1 = ix.small_batch_stats()
                latency["nq1_tail"] = {"calls": int(ncalls), "distinct_queries": int(qtail.shape[0]),
                                       **{f"p{str(p_).replace('.', '_')}_us": round(float(np.percentile(latt, p_)), 1) for p_ in (50, 99, 99.9, 99.99)},
                                       "max_us": round(float(latt.max()), 1),
                                       "queries_that_failed_the_small_batch_check": sb1["failed_queries"] - sb0["failed_queries"],
                                       "answered_by_cell_rescan": sb1["repaired_by_cell_rescan"] - sb0["repaired_by_cell_rescan"],
                                       "answered_by_exact_scan": sb1["exact_scans"] - sb0["exact_scans"]}
            except Exception as e:   # noqa: BLE001 - a side field must not take the line down
                latency["nq1_tail"] = {"error": f"{type(e).__name__}: {e}"}
        latency["note"] = ("median wall time of one cgv_search_f32 call (host buffers in, host results out), path 0 = exact scan, "
                           "1 = MFMA coarse + exact re-score; corpus_stream_gb_per_s = rows x dim x s / best median")
    wd.kick("side measurements done")

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
            cycles_xcd, mfma_busy = None, None
            if pmc and world == 1:
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
                # correction + WRITE_SIZE; scripts/collect_profiles.sh) - only from a pass taken on THESE kernel sources
                pj = json.load(open(os.path.join(pdir, pmc[-1])))
                if pj.get("source_sha16") == source_sha16():
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/" + pmc[-1]
                    if pj.get("GRBM_GUI_ACTIVE"):
                        cycles_xcd = pj["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
                        if pj.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                            mfma_busy = pj["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles_xcd   # 1024 SIMDs
                else:
                    traffic_note = (f"profiles/{pmc[-1]} was collected on other kernel sources (stamp "
                                    f"{pj.get('source_sha16')} != {source_sha16()}): not quoted")
            abytes = float(coarse_rows) * dim * ESIZE[dtype] + batch * dim * ESIZE[dtype] + coarse_rows * 4
            gbs = abytes / (cms * 1e-3) / 1e9
            mfma_frac, hbm_frac = ach / PEAK_TFLOPS[dtype], gbs / PEAK_HBM_GBS
            # report against whichever roof binds this shape (intensity ~ batch FLOP/B:
            # batch >= ~512 -> MFMA, C4's batch 256 -> HBM); the other fraction rides along. The nominal MFMA peak assumes 2.4 GHz;
            # under MFMA load the part runs at 1.5-1.8 GHz (power), so where the two nominal fractions are close (C4) the
            # comparison is made at the MEASURED clock: cycles per launch from the stamped PMC pass / this run's launch time
            # (C4's launch ran at 1.50 GHz with MFMA busy 0.74 - the matrix pipe, not HBM, binds it).
            # (cycles per launch are a property of the kernel and its data; this run's duration turns them into this run's clock)
            clock_ghz = (cycles_xcd / (cms * 1e-3) / 1e9) if cycles_xcd else None
            mfma_at_clock = (mfma_frac * 2.4 / clock_ghz) if clock_ghz else None
            hbm_achievable = hbm_frac * 8.0 / 6.3     # MI355X_MICROARCH.md: ~6.3 TB/s is what a streaming kernel reaches
            hbm_binds = (hbm_achievable > mfma_at_clock) if mfma_at_clock else (hbm_frac > mfma_frac)
            limits = {"clock_ghz": round(clock_ghz, 3) if clock_ghz else None, "nominal_clock_ghz": 2.4,
                      "mfma_busy": round(mfma_busy, 3) if mfma_busy else None,
                      "mfma_frac_nominal": round(mfma_frac, 4),
                      "mfma_frac_at_measured_clock": round(mfma_at_clock, 4) if mfma_at_clock else None,
                      "hbm_frac_of_8TBs": round(hbm_frac, 4), "hbm_frac_of_achievable_6.3TBs": round(hbm_achievable, 4),
                      "clock_source": ("GRBM_GUI_ACTIVE / 8 of " + traffic_src + " / this run's avg_launch_ms") if clock_ghz else None}
            if hbm_binds:
                roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4)}
            else:
                roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                        "frac": round(mfma_frac, 4), "hbm_frac": round(hbm_frac, 4)}
            roof.update({"limits": limits, "kernel": "coarse_kernel (main stage)", "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "traffic_note": traffic_note, "avg_launch_ms": round(cms, 4),
                         "rows_per_launch": int(coarse_rows), "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": abytes, "rank": 0})
        if roof is None and st["last_path"] == 0 and med > 0:
            # exact-scan index (C1: f32 rows, one query): no coarse launch; the step IS the scan, priced against HBM (SURVEY.md
            # section 8(d): single-query shapes are HBM-bound) - latency-dominated at this size, and said so
            abytes = float(n_total) * dim * ESIZE[dtype] * batch
            gbs = abytes / (med * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "kernel": "exact_scores_kernel + topk_chunk_kernel (whole host-timed step: the device pipeline of one "
                              "single-query call is launch- and latency-bound at this corpus size)",
                    "traffic": None, "avg_launch_ms": None, "median_step_ms": round(med, 4),
                    "algorithmic_bytes_per_launch": abytes, "rank": 0}
        result = {
            "metric": "queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {n_total} x {dim} {dtype} {metric} brute-force kNN, "
                                   f"batch={batch}, k={k}", "rows": n_total, "dim": dim, "batch": batch, "k": k,
                       "metric": metric, "sharding": f"rows/{world}" if world > 1 else "none",
                       "rng": ("counter-based (Philox4x32-10 keyed seed,row,col -> N(0,1) f32 -> unit norm; corpus seed 0xC0DE6001, "
                               "queries 0xC0DE6002: SURVEY.md 8(d))") if RNG == "counter" else "torch.randn per 125 k-row chunk",
                       "step": ("one batch of queries ALREADY RESIDENT IN HBM (every rank's copy, N > 1) -> search (-> all-gather + merge) "
                                "-> results written to pinned host memory; serial batches") if hbm else
                               "one batch: pinned host queries -> H2D -> search -> D2H host results, serial batches",
                       "queries_start_in": "hbm" if hbm else "pinned host memory",
                       "exchange": "RCCL all-gather of per-shard top-k + merge (see multi_gpu)" if world > 1 else "none",
                       "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}},
            "median_ms_per_step": round(med, 4), "median_qps": round(batch / (med * 1e-3), 1),
            "step_ms_percentiles": {p_: round(float(np.percentile(step_ms, p_)), 4) for p_ in (1, 10, 50, 90, 99, 100)},
            "pipelined_host_qps": pipelined_host["queries_per_sec"] if pipelined_host else None,
            "pipelined_host": pipelined_host,
            "concurrent_callers": concurrent,
            "coalesced_callers": coalesced,
            "pipelined_qps": pipelined["queries_per_sec"] if pipelined else None,
            "pipelined": pipelined,
            ("pcie_inclusive_serial" if hbm else "hbm_resident_serial"): other_serial,
            "hbm_resident_results_in_hbm": resident,
            "latency": latency,
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

    def host_cpus():
        """What the CPU leg may actually use: the affinity mask and the cgroup CPU quota of this process (a container can see
        128 cores and be allowed 16)."""
        info = {"nproc_affinity": len(os.sched_getaffinity(0)), "os_cpu_count": os.cpu_count(), "cgroup_cpu_max": None}
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                info["cgroup_cpu_max"] = open(f).read().strip()
                break
            except OSError:
                pass
        return info

    def oracle_leg(o, rs, qh, gi, gs, budget_s, max_q, sweep=False):
        """Time the CPU port on single-query searches (the reference's shape) and compare every answer with the device's."""
        omet = o.COSINE if metric == "cosine" else o.DOT
        cores = o.max_threads()
        for _ in range(3):                    # warm-up: thread pool, the reusable (score, index) buffer's pages
            rs.top_k(qh[0], k, omet, cores)
        swept = None
        if sweep:   # the baseline runs at ITS best thread count: more threads than the quota / the memory system serves only hurt
            swept = {}
            quota = None
            try:   # cgroup v2 "max period" / "<quota us> <period us>": the CPUs this container may actually burn
                qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                quota = None if qv == "max" else max(1, int(round(int(qv) / int(pv))))
            except (OSError, ValueError):
                pass
            cand = {cores, max(1, cores // 2), max(1, cores // 4), min(cores, len(os.sched_getaffinity(0)))}
            if quota:
                cand |= {min(cores, quota), min(cores, 2 * quota)}
            for c in sorted(cand):
                # >= 0.4 s of wall time per setting and the MEAN of it: a quota throttles a burst of 128 threads only after
                # the first periods (r05: three 8 ms queries on 128 threads looked best, the timed leg then ran at 70 ms / query)
                ts, t_set = [], time.perf_counter()
                while len(ts) < 3 or (time.perf_counter() - t_set < 0.4 and len(ts) < 200):
                    t1 = time.perf_counter()
                    rs.top_k(qh[len(ts) % len(qh)], k, omet, c)
                    ts.append(time.perf_counter() - t1)
                swept[c] = round(1e3 * float(np.mean(ts)), 2)
            cores = min(swept, key=swept.get)
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
                "exact": exact_scores / nqc, "score_ms": sc_ms / nqc, "sort_ms": so_ms / nqc, "thread_sweep_ms": swept}

    if want_cpu:
        from oracle import oracle as o   # CPU baseline + recall checker only
        rows_host = np.concatenate(host_chunks)
        del host_chunks
        rng_same = None
        if RNG == "counter" and lo == 0:   # the device's corpus and queries are what the CPU generator produces (sampled here; -m gpu: tests/test_synth.py)
            ocode = {"f32": o.F32, "f32s": o.F32, "bf16": o.BF16, "fp16": o.FP16, "fp8": o.FP8}[dtype]
            ns = min(4096, len(rows_host))
            rng_same = bool(np.array_equal(rows_host[:ns], o.round_trip(o.synth_rows(SEED_CORPUS, 0, ns, dim), ocode, fp8_codes=True))
                            and np.array_equal(qpool[0].cpu().numpy(), o.synth_rows(SEED_QUERY, 0, batch, dim)))
        rs = o.RowSet(rows_host)
        del rows_host
        qcat = qpool[0] if batch >= args.cpu_max_queries else torch.cat(qpool)    # (single-query workloads: one query per batch)
        qcat = qcat[:args.cpu_max_queries]
        qh = storage_values(qcat, dtype).cpu().numpy()
        if dtype == "fp8":   # the torch expression of the storage format must be the oracle's
            assert np.array_equal(qh, o.round_trip(qcat.cpu().numpy(), o.FP8, fp8_codes=True))
        if batch >= args.cpu_max_queries:
            gi, gs = ix.search(qpool[0], k)
        else:
            parts = [ix.search(q, k) for q in qpool[: (len(qcat) + batch - 1) // batch]]      # one device call per batch, as timed
            gi, gs = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
        gi = gi.cpu().numpy().view(np.uint64)
        gs = gs.cpu().numpy()
        leg = oracle_leg(o, rs, qh, gi, gs, args.cpu_seconds, len(qcat), sweep=True)
        rs.close()
        nqc, cpu_t = leg["n"], leg["seconds"]
        scan_gbs = n_total * dim * 4 / max(leg["score_ms"], 1e-9) / 1e6
        result["cpu_baseline"] = {"value": round(nqc / cpu_t, 3), "unit": "queries/s", "cores": leg["cores"],
                                  "kind": "port", "numa_nodes": o.numa_nodes(), "host": host_cpus(),
                                  "inputs_same_as_cpu_generator": rng_same,
                                  "thread_sweep_ms_per_query": leg["thread_sweep_ms"],
                                  "score_ms": round(leg["score_ms"], 2), "sort_ms": round(leg["sort_ms"], 2),
                                  "scan_gb_per_s": round(scan_gbs, 1), "scan_gb_per_s_per_thread": round(scan_gbs / leg["cores"], 2),
                                  "sample": f"{nqc} single-query searches over the same {n_total} x {dim} corpus "
                                            f"(f32 upcast of the {dtype} values, rows separately allocated and first touched "
                                            f"by the threads that scan them), {cpu_t:.1f} s wall; C++ port of "
                                            f"parallel_top_k_search: threaded AVX2 scoring (score_ms) + a fully parallel sort "
                                            f"of all (score, index) pairs (sort_ms; libstdc++ parallel multiway mergesort "
                                            f"standing in for rayon's par_sort_unstable_by), pair buffer reused across queries"}
        result["recall_at_10"] = leg["recall"]
        result["ordered_match_rate"] = leg["ordered"]
        result["score_bit_exact_rate"] = leg["exact"]
        result["recall_sample"] = f"{nqc} of the {batch} queries of one batch against the CPU oracle (all {batch}: exact_check)"
        result["speedup_vs_cpu_baseline"] = round(result["value"] / (nqc / cpu_t), 1)
    elif rank == 0:
        result["cpu_baseline"] = None

    # EVERY answer of one batch (all `batch` queries; fewer only where the exact scan of that many would take minutes) against the
    # EXACT device scan - the proven-correct fallback path, held against the oracle by the -m gpu tests - through the same step:
    # N = 1: cgv_search_f32 with the coarse pass switched off; N > 1: every record PROVISIONAL -> redo word -> every rank's
    # exact scan -> second exchange + merge, i.e. the redo protocol itself runs across all ranks
    wd.kick("exact check")
    nex = int(min(batch, max(32, 2_000_000_000 // max(n_total, 1))))
    exact_check = None
    if args.check_queries > 0 or want_cpu:
        step(0)
        fi, fs = out_i.numpy().view(np.uint64).copy(), out_s.numpy().copy()
        redo0 = searcher.redo_batches if dist is not None else 0
        ix.set_force_exact(True)
        if nex == batch:
            step(0)
            ei, es = out_i.numpy().view(np.uint64).copy(), out_s.numpy().copy()
        else:
            qs = qhost[0][:nex].clone().pin_memory()
            o_i = torch.empty((nex, k), dtype=torch.int64).pin_memory()
            o_s = torch.empty((nex, k), dtype=torch.float32).pin_memory()
            if dist is None:
                ix.search_host_ptr(qs.data_ptr(), nex, k, o_i.data_ptr(), o_s.data_ptr())
            else:
                searcher.step_packed(qs, k, out=(o_i, o_s), device=dev)
            ei, es = o_i.numpy().view(np.uint64).copy(), o_s.numpy().copy()
        ix.set_force_exact(False)
        if rank == 0:
            hits = sum(len(set(ei[q].tolist()) & set(fi[q].tolist())) for q in range(nex))
            exact_check = {"anchor": "device-exact-scan through the same step" + (" (all ranks, redo protocol)" if dist is not None else ""),
                           "queries": nex, "recall_at_10": hits / (nex * k),
                           "ordered_match_rate": float(np.mean([np.array_equal(ei[q], fi[q]) for q in range(nex)])),
                           "score_bit_exact_rate": float(np.mean([np.array_equal(es[q], fs[q]) for q in range(nex)])),
                           "redo_batches_of_the_check": (searcher.redo_batches - redo0) if dist is not None else None}
            result["exact_check"] = exact_check

    # N > 1 (and the one-rank dry run): the MERGED result of one batch against the CPU oracle over the WHOLE corpus, in the same
    # line (VERDICT r3 'Next' #4); corpora beyond the CPU leg (C3 at 8 GPUs) carry the exact check above as their anchor
    if dist is not None and args.check_queries > 0:
        ncheck = min(args.check_queries, batch)
        step(0)                                           # out_i / out_s: merged results of batch 0 (pinned host arrays)
        gi = out_i.numpy().view(np.uint64)[:ncheck].copy()
        gs = out_s.numpy()[:ncheck].copy()
        wd.stop()                                         # (the other ranks wait in the final barrier for rank 0's CPU leg)
        if rank == 0 and oracle_fits:
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
                                       f"included) against the CPU oracle over the whole {n_total}-row corpus; all {nex}: exact_check")
            result["check"] = {"anchor": "cpu-oracle", "queries": leg["n"], "oracle_score_ms": round(leg["score_ms"], 2),
                               "oracle_sort_ms": round(leg["sort_ms"], 2), "redo_batches": multi["redo_batches"]}
        elif rank == 0:
            result["recall_at_10"] = exact_check["recall_at_10"]
            result["ordered_match_rate"] = exact_check["ordered_match_rate"]
            result["score_bit_exact_rate"] = exact_check["score_bit_exact_rate"]
            result["recall_sample"] = (f"{nex} queries of one MERGED batch against the exact device scan of every shard + the same "
                                       f"exchange and merge (the {n_total}-row corpus is beyond the CPU leg)")
            result["check"] = {"anchor": "device-exact-scan", "queries": nex, "redo_batches": multi["redo_batches"]}

    wd.stop()
    gc.enable()
    if rank == 0:
        emit(result)
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
