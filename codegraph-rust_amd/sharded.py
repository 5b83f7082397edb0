"""Row-sharded kNN across ranks (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm). SURVEY.md §8(e): the corpus is split into contiguous row ranges, every
rank searches its shard with the full query batch, the per-shard partial top-k
(nq*k (id, score) records per rank) are exchanged with ONE all-gather, and every rank
merges G*k candidates per query with (score desc, id asc). Exact: the global top-k is a
subset of the union of the local top-k.

The reference has no distributed path (no collective anywhere); this is the
multi-GPU extension BASELINE.json's north_star asks for.
"""
import numpy as np
import torch
import torch.distributed as dist

PROVISIONAL_ID = 0xFFFFFFFFFFFFFFFE   # CGV_PROVISIONAL_ID (include/cgvec.h): id slot 0 of a record the rank will redo


def packed_width(k):
    """int32 words per query of a packed record: k u64 ids | k f32 scores | one pad word when k is odd (cgv_packed_width)."""
    return 3 * int(k) + (int(k) & 1)


def pack_records_host(idx, score, provisional=None):
    """Host restatement of pack_topk_kernel (csrc/kernels_select.h) for the CPU (gloo) tests and as documentation of the wire
    format: [nq, k] uint64 ids + f32 scores -> int32 [nq, packed_width(k)]; rows of `provisional` queries carry
    PROVISIONAL_ID in id slot 0."""
    idx = np.ascontiguousarray(idx, dtype=np.uint64).copy()
    score = np.ascontiguousarray(score, dtype=np.float32)
    nq, k = idx.shape
    if provisional is not None:
        idx[np.asarray(provisional, dtype=bool), 0] = np.uint64(PROVISIONAL_ID)
    rec = np.zeros((nq, packed_width(k)), dtype=np.int32)
    rec[:, :2 * k] = idx.view(np.int32).reshape(nq, 2 * k)
    rec[:, 2 * k:3 * k] = score.view(np.int32).reshape(nq, k)
    return rec


def merge_packed_host(gathered, k):
    """Host restatement of merge_topk_kernel over the all-gather output [G, nq, packed_width(k)] int32 ->
    (ids uint64 [nq, k], scores f32 [nq, k], redo): (score desc, id asc), padding (UINT64_MAX) last; redo is True when any
    list is provisional (its entries are then ignored, the batch's exchange must be repeated)."""
    g = np.ascontiguousarray(gathered, dtype=np.int32)
    G, nq, _ = g.shape
    ids = np.ascontiguousarray(g[:, :, :2 * k]).view(np.uint64).reshape(G, nq, k)
    sc = np.ascontiguousarray(g[:, :, 2 * k:3 * k]).view(np.float32).reshape(G, nq, k)
    out_i = np.full((nq, k), np.uint64(2**64 - 1), dtype=np.uint64)
    out_s = np.full((nq, k), -np.inf, dtype=np.float32)
    redo = False
    for q in range(nq):
        cand = []
        for r in range(G):
            for j in range(k):
                i = int(ids[r, q, j])
                if i == PROVISIONAL_ID:
                    redo = True
                    continue
                if i != 2**64 - 1:
                    cand.append((-float(sc[r, q, j]), i, sc[r, q, j]))
        cand.sort(key=lambda t: (t[0], t[1]))
        for j, (_, i, s_) in enumerate(cand[:k]):
            out_i[q, j], out_s[q, j] = np.uint64(i), s_
    return out_i, out_s, redo


def shard_range(n_total, rank, world):
    """Contiguous rows [lo, hi) of rank `rank`: ceil(N/G) rows per rank (last may be short)."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


class _Batch:
    """One batch in flight of ShardedKnn.step_packed_begin / _end: its record / gather buffers, the pinned redo word the merge
    kernel raises, the stream the whole batch is enqueued on (shard search, packing, all-gather, merge - in line, no hop) and
    the event behind its last kernel. Reused round-robin; `busy` between begin and end."""
    __slots__ = ("key", "rec", "gathered", "redo", "redo_np", "stream", "stream_ptr", "done", "ev", "ticket", "k", "nq", "out", "res",
                 "busy", "mode", "timed", "queries", "qkey", "qslice", "qfull")

    def __init__(self):
        self.key = self.qkey = None
        self.stream = self.done = self.ev = None
        self.qslice = self.qfull = None
        self.busy = False


class ShardedKnn:
    """`local` is any object with search(queries, k) -> (idx int64[nq,k], score f32[nq,k]) whose
    ids are already GLOBAL (HipKnnIndex.set_index_base(lo)); `merge` maps gathered
    [G,nq,k] tensors to [nq,k] (default: the HIP merge kernel through the C ABI)."""

    MAX_IN_FLIGHT = 3   # = the handle's search contexts (cgv_max_batches_in_flight)

    def __init__(self, local, rank=None, world=None, group=None, merge=None, force_collective=False):
        self.local = local
        self.force_collective = force_collective   # world == 1: run pack / all-gather / merge anyway (one-rank RCCL smoke test)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self._device_merge = merge is None   # default: packed records + the HIP merge kernel
        self._gathered = None
        self._slots = [_Batch() for _ in range(self.MAX_IN_FLIGHT)]
        self._next = 0
        if merge is None:
            from .cgvec import merge_topk
            merge = merge_topk
        self.merge = merge

    def search(self, queries, k):
        return self._exchange(*self.local.search(queries, k), k)

    # ---- join-free step: search -> pack -> all-gather -> merge enqueued in line on ONE stream per batch, one host
    # synchronisation per batch; provisional records make every rank repeat the exchange (include/cgvec.h). Round 5: split into
    # begin / end so that 2-3 batches are in flight (each on its own stream): the host reads batch i's redo word while batch
    # i + 1 runs - the fixed per-batch cost that does not shrink with the shard (PCIe read of the replicated batch, sample, final,
    # exchange, host) hides behind the neighbours' coarse kernels ("unless batches are pipelined") ----
    redo_batches = 0
    time_exchange = False      # HIP-event pair around all-gather + merge on the stream they run on (last_exchange_ms)
    time_phases = False        # ... and at every phase boundary of the batch (last_phase_ms): three more records per batch,
                               # a few microseconds each - for diagnostic steps, not for timed ones
    last_exchange_ms = 0.0     # all-gather + merge of the last ended batch
    last_phase_ms = None       # {"queries", "search_pack", "all_gather", "merge"} of the last ended batch (time_exchange)
    # How a HOST query batch reaches the ranks ("queries replicated ... broadcast or loaded by every rank"):
    #   "replicated"  every rank reads the whole batch over its own PCIe link (in place, by the shard search's conversion kernel,
    #                 or by the copy engine while another batch is in flight): 3 MB per rank at C2 whatever the rank count - the
    #                 largest per-batch cost that does not shrink with the shard;
    #   "sharded"     rank r copies only rows [r * ceil(nq / G), ...) to its device (nq * D * 4 / G bytes over PCIe) and ONE
    #                 all-gather of the f32 slices over xGMI gives every rank the whole batch in HBM; the shard search then
    #                 converts from device memory. The f32 values every rank converts are the same as in the replicated form,
    #                 so the results are identical bit for bit. (The slices travel as f32, not as converted rows: 3 MB instead
    #                 of 1.5 MB is latency-bound either way at these sizes, and the conversion stays one code path.)
    query_exchange = "replicated"

    def _collective(self):
        """None: no collective (one rank, not forced); else the backend of the group ("nccl" = RCCL, "gloo")."""
        if self.world == 1 and not self.force_collective:
            return None
        return self._backend()

    def _backend(self):
        # (no process group: a test's in-process stand-in for all_gather_into_tensor - the device branch)
        return dist.get_backend(self.group) if dist.is_available() and dist.is_initialized() else "nccl"

    def _take_slot(self, nq, k, like):
        w = packed_width(k)
        for _ in range(len(self._slots)):
            b = self._slots[self._next]
            self._next = (self._next + 1) % len(self._slots)
            if not b.busy:
                break
        else:
            raise RuntimeError(f"ShardedKnn: {len(self._slots)} batches already in flight - call step_packed_end first")
        key = (nq, w, like.device if like.is_cuda else None, self.world)
        if b.key != key:
            dev = like.device if like.is_cuda else torch.device("cpu")
            b.rec = torch.empty((nq, w), dtype=torch.int32, device=dev)
            b.gathered = torch.empty((self.world, nq, w), dtype=torch.int32, device=dev)
            b.redo = torch.zeros(1, dtype=torch.int32)
            if like.is_cuda:
                b.redo = b.redo.pin_memory()   # written in place by the merge kernel, read by the host after the batch's event
                b.stream = torch.cuda.Stream(device=dev)
                b.stream_ptr = b.stream.cuda_stream
                b.done = torch.cuda.Event()
                b.ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(5))   # start | queries | search + pack | all-gather | merge
            b.redo_np = b.redo.numpy()         # (the same word: the interpreter reads and clears it without a tensor op)
            b.key = key
        return b

    def _shard_bounds(self, nq):
        per = (nq + self.world - 1) // self.world
        lo = min(nq, self.rank * per)
        return per, lo, min(nq, lo + per)

    def _gather_queries(self, b, queries, device):
        """query_exchange == "sharded": this rank's slice of the host batch -> device, all-gather of the slices -> the whole
        batch in device memory (CUDA ranks), or through gloo (CPU stand-ins; device shards of several ranks on one GPU)."""
        nq, d = queries.shape
        per, lo, hi = self._shard_bounds(nq)
        coll = self._collective()
        if b.rec.is_cuda:
            key = (per, d, self.world)
            if b.qkey != key:
                b.qslice = torch.zeros((per, d), dtype=torch.float32, device=b.rec.device)   # (a short last slice: zero tail, never read)
                b.qfull = torch.empty((self.world * per, d), dtype=torch.float32, device=b.rec.device)
                b.qkey = key
            if coll == "gloo" or coll is None:
                torch.cuda.current_stream(b.rec.device).synchronize()
                mine = torch.zeros((per, d), dtype=torch.float32)
                mine[: hi - lo] = queries[lo:hi]
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                if self.world > 1 and coll is not None:
                    dist.all_gather(parts, mine, group=self.group)
                else:
                    parts[0].copy_(mine)
                b.qfull.copy_(torch.cat(parts))
            else:
                if hi > lo:
                    b.qslice[: hi - lo].copy_(queries[lo:hi], non_blocking=True)    # pinned host -> HBM on the batch's stream
                dist.all_gather_into_tensor(b.qfull, b.qslice, group=self.group)
            return b.qfull[:nq]
        mine = torch.zeros((per, d), dtype=torch.float32)
        mine[: hi - lo] = queries[lo:hi]
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(parts, mine, group=self.group)
        else:
            parts[0].copy_(mine)
        return torch.cat(parts)[:nq].contiguous()

    def _gather_merge(self, b):
        """all-gather of the batch's records + merge (redo word) - enqueued on the current stream (RCCL), or, when the records
        travel over gloo, through the host (CPU stand-ins of the tests; real device shards of several ranks on ONE GPU, which RCCL
        refuses: the dry run of the multi-GPU bench on a single-GPU box)."""
        rec, gathered, k, out, redo = b.rec, b.gathered, b.k, b.out, b.redo
        coll = self._collective()
        if rec.is_cuda:
            from .cgvec import merge_packed
            if coll is None:
                g = rec.view(1, rec.shape[0], rec.shape[1])
            elif coll == "gloo":
                torch.cuda.current_stream(rec.device).synchronize()
                mine = rec.cpu()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                if self.world > 1:
                    dist.all_gather(parts, mine, group=self.group)
                else:
                    parts[0].copy_(mine)
                gathered.copy_(torch.stack(parts))
                g = gathered
            else:
                dist.all_gather_into_tensor(gathered, rec, group=self.group)
                g = gathered
            if b.timed == 2 and b.mode == "stream":
                b.ev[3].record()
            return merge_packed(g, k, out=out, redo=redo)
        parts = [gathered[r] for r in range(self.world)]
        if self.world > 1:
            dist.all_gather(parts, rec, group=self.group)     # gloo (CPU tests)
        else:
            parts[0].copy_(rec)
        ids, sc, again = merge_packed_host(gathered.numpy(), k)
        redo[0] = 1 if again else 0
        oi, os_ = torch.from_numpy(ids.view(np.int64)), torch.from_numpy(sc)
        if out is not None:
            out[0].copy_(oi)
            out[1].copy_(os_)
            return out
        return oi, os_

    @staticmethod
    def _wait(event, spin_s=0.005):
        """The batch's ONE host synchronisation: poll its event for a few milliseconds before blocking - a batch takes 0.3-1.5
        ms, and the wake-up of a blocked synchronize costs tens of microseconds of it (the library waits for its own
        streams the same way, search.hip: wait_stream)."""
        import time
        t0 = time.perf_counter()
        while not event.query():
            if time.perf_counter() - t0 > spin_s:
                event.synchronize()
                return

    def step_packed_begin(self, queries, k, out=None, device=None):
        """First half of one batch: local.search_packed_begin (shard search + packed records) and - when the records travel
        over RCCL - the all-gather and the merge with the redo flag, all enqueued in line on the batch's own stream; returns a
        handle for step_packed_end. Up to MAX_IN_FLIGHT batches may be begun before the first is ended; EVERY rank must
        begin and end its batches in the same order (the collectives are posted in that order).
        `queries`: CUDA tensor, or pinned CPU tensor read in place over PCIe (then `device` names the GPU); `out` = (ids, scores)
        tensors to fill (CUDA, or pinned CPU written in place by the merge kernel).
        `local` needs search_packed_begin(queries, k, rec) -> ticket and search_packed_end(ticket) -> bool."""
        nq = queries.shape[0]
        like = queries if queries.is_cuda else (torch.empty(0, device=device) if device is not None else queries)
        b = self._take_slot(nq, int(k), like)
        b.k, b.nq, b.out, b.res, b.queries = int(k), nq, out, None, queries
        b.redo_np[0] = 0                     # (the slot is idle: no kernel writes it)
        b.timed = 2 if (b.rec.is_cuda and self.time_phases) else (1 if (b.rec.is_cuda and self.time_exchange) else 0)
        if b.rec.is_cuda:
            coll = self._collective()
            b.mode = "host" if coll == "gloo" else "stream"
            if queries.is_cuda:              # the producer of a CUDA batch comes first - when it still has work in flight
                cur = torch.cuda.current_stream(b.rec.device)
                if not cur.query():
                    b.stream.wait_stream(cur)
            sharded_q = self.query_exchange == "sharded" and not queries.is_cuda and coll is not None
            early = not sharded_q and b.timed != 2 and hasattr(self.local, "set_stream")
            if early:   # the shard search goes onto the batch's stream before the interpreter enters the stream context (which only
                #         the collective and the merge need): step_packed spent 24 us before this call, scripts/step_packed_probe.py
                b.ticket = self.local.search_packed_begin(queries, k, b.rec, stream=b.stream_ptr)
            with torch.cuda.stream(b.stream):
                if b.timed == 2:
                    b.ev[0].record()
                q_in = self._gather_queries(b, queries, device) if sharded_q else queries
                if b.timed == 2:
                    b.ev[1].record()
                if not early:
                    b.ticket = self.local.search_packed_begin(q_in, k, b.rec)
                if b.mode == "stream":
                    if b.timed:
                        b.ev[2].record()
                    b.res = self._gather_merge(b)
                    if b.timed:
                        b.ev[4].record()
                b.done.record()
        else:
            b.mode = "cpu"
            q_in = self._gather_queries(b, queries, device) if (self.query_exchange == "sharded" and self.world > 1) else queries
            b.ticket = self.local.search_packed_begin(q_in, k, b.rec)
        b.busy = True
        return b

    def step_packed_end(self, b):
        """Second half: ONE host wait for the batch, local.search_packed_end, and - only when the merge met a provisional record
        (the same word on every rank) - the exchange once more with the final records. Returns (ids, scores)."""
        if not b.busy:
            raise RuntimeError("ShardedKnn.step_packed_end: the batch is not in flight")
        cuda = b.rec.is_cuda
        try:
            if cuda:
                self._wait(b.done)
                if b.mode == "host":             # records over gloo: the exchange happens here, through the host
                    with torch.cuda.stream(b.stream):
                        b.res = self._gather_merge(b)
                        b.done.record()
                    self._wait(b.done)
                elif b.timed == 2:
                    t = [b.ev[i].elapsed_time(b.ev[i + 1]) for i in range(4)]
                    self.last_phase_ms = {"queries": t[0], "search_pack": t[1], "all_gather": t[2], "merge": t[3]}
                    self.last_exchange_ms = t[2] + t[3]
                elif b.timed:
                    self.last_exchange_ms = b.ev[2].elapsed_time(b.ev[4])
            else:
                b.res = self._gather_merge(b)
            # A search that fails on ONE rank only (an OOM in its exact rescan ...) must not take that rank out of the collective
            # sequence: the redo word is the same on every rank, so every rank - this one included - runs the second
            # exchange, and the error is raised after it
            err = None
            try:
                self.local.search_packed_end(b.ticket)
            except Exception as e:   # noqa: BLE001 - re-raised below, after the collective
                err = e
            if int(b.redo_np[0]) != 0:
                self.redo_batches += 1
                b.redo_np[0] = 0
                if cuda:
                    with torch.cuda.stream(b.stream):
                        b.res = self._gather_merge(b)
                        b.done.record()
                    self._wait(b.done)
                else:
                    b.res = self._gather_merge(b)
                if err is None and int(b.redo_np[0]) != 0:
                    raise RuntimeError("records still provisional after search_packed_end (a rank's search failed)")
            if err is not None:
                raise err
            return b.res
        finally:
            b.busy = False
            b.queries = None

    def step_packed(self, queries, k, out=None, device=None):
        """One batch, one host synchronisation: step_packed_begin + step_packed_end (strictly serial batches)."""
        return self.step_packed_end(self.step_packed_begin(queries, k, out=out, device=device))

    def _exchange(self, idx, score, k, out=None):
        """The unpacked exchange of search(): results of a finished local search -> all-gather -> merge.
        out = (ids, scores): optional result tensors of the device path (CUDA, or pinned CPU tensors the merge kernel
        fills in place - the caller synchronises the stream before reading them)."""
        if self.world == 1 and not self.force_collective:
            return idx, score
        nq = idx.shape[0]
        # ONE all-gather of nq packed records (k u64 ids + k f32 scores, 12 B per hit) per rank:
        # latency-bound (<= 1 MiB per rank at nq=8192, k=10), so a single collective.
        gloo = self._backend() == "gloo"
        if idx.is_cuda and self._device_merge and not gloo:
            from .cgvec import merge_packed, pack_topk
            rec = pack_topk(idx, score)                       # one kernel instead of cat + 2 slice copies
            key = (self.world,) + tuple(rec.shape)
            if self._gathered is None or tuple(self._gathered.shape) != key or self._gathered.device != rec.device:
                self._gathered = torch.empty(key, dtype=torch.int32, device=rec.device)   # reused across batches
            dist.all_gather_into_tensor(self._gathered, rec, group=self.group)
            return merge_packed(self._gathered, k, out=out)
        rec = torch.cat([idx.contiguous().view(torch.int32).reshape(nq, 2 * k),
                         score.contiguous().view(torch.int32).reshape(nq, k)], dim=1).contiguous()
        dev = rec.device
        if gloo:  # CPU tests; device shards of several ranks on one GPU (records through the host)
            rec = rec.cpu()
            gathered = torch.empty((self.world, nq, 3 * k), dtype=torch.int32)
            parts = [gathered[r] for r in range(self.world)]
            dist.all_gather(parts, rec, group=self.group)
            gathered = gathered.to(dev)
        else:
            gathered = torch.empty((self.world, nq, 3 * k), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(gathered, rec, group=self.group)
        g_idx = gathered[:, :, : 2 * k].contiguous().view(torch.int64).reshape(self.world, nq, k)
        g_score = gathered[:, :, 2 * k:].contiguous().view(torch.float32).reshape(self.world, nq, k)
        res = self.merge(g_idx, g_score)
        if out is not None:
            out[0].copy_(res[0])
            out[1].copy_(res[1])
            return out
        return res
