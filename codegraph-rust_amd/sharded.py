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


class ShardedKnn:
    """`local` is any object with search(queries, k) -> (idx int64[nq,k], score f32[nq,k]) whose
    ids are already GLOBAL (HipKnnIndex.set_index_base(lo)); `merge` maps gathered
    [G,nq,k] tensors to [nq,k] (default: the HIP merge kernel through the C ABI)."""

    def __init__(self, local, rank=None, world=None, group=None, merge=None, force_collective=False):
        self.local = local
        self.force_collective = force_collective   # world == 1: run pack / all-gather / merge anyway (one-rank RCCL smoke test)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self._device_merge = merge is None   # default: packed records + the HIP merge kernel
        self._gathered = None
        if merge is None:
            from .cgvec import merge_topk
            merge = merge_topk
        self.merge = merge

    def search(self, queries, k):
        return self._exchange(*self.local.search(queries, k), k)

    # ---- join-free step (round 4): search -> pack on the library's stream, collective + merge enqueued behind an event,
    # ONE host synchronisation per batch; provisional records make every rank repeat the exchange (include/cgvec.h) --------
    redo_batches = 0
    time_exchange = False      # step_packed: HIP-event pair around all-gather + merge on the stream they run on
    last_exchange_ms = 0.0
    _ev = None

    def _buffers(self, nq, k, like):
        w = packed_width(k)
        key = (nq, w, str(like.device) if like.is_cuda else "cpu")
        if getattr(self, "_pk_key", None) != key:
            dev = like.device if like.is_cuda else torch.device("cpu")
            self._pk_rec = torch.empty((nq, w), dtype=torch.int32, device=dev)
            self._pk_gathered = torch.empty((self.world, nq, w), dtype=torch.int32, device=dev)
            self._pk_redo = torch.zeros(1, dtype=torch.int32)
            if like.is_cuda:
                self._pk_redo = self._pk_redo.pin_memory()   # written in place by the merge kernel, read by the host after the sync
            self._pk_key = key
        return self._pk_rec, self._pk_gathered, self._pk_redo

    def _gather_merge(self, rec, gathered, k, out, redo):
        if rec.is_cuda:
            from .cgvec import merge_packed
            dist.all_gather_into_tensor(gathered, rec, group=self.group)
            return merge_packed(gathered, k, out=out, redo=redo)
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
    def _wait(device, spin_s=0.005):
        """The batch's ONE host synchronisation: poll the stream for a few milliseconds before blocking - a batch takes 0.3-1.5
        ms, and the wake-up of a blocked hipStreamSynchronize costs tens of microseconds of it (the library waits for its own
        streams the same way, cgvec.hip: wait_stream)."""
        import time
        st = torch.cuda.current_stream(device)
        t0 = time.perf_counter()
        while not st.query():
            if time.perf_counter() - t0 > spin_s:
                st.synchronize()
                return

    def step_packed(self, queries, k, out=None, device=None):
        """One batch, one host synchronisation: local.search_packed_begin (shard search + packed records, the consumer stream
        waits for them) -> all-gather -> merge with the redo flag -> sync -> local.search_packed_end. When any rank's records
        were provisional (the same flag on every rank), the exchange is repeated with the final records.
        `local` needs search_packed_begin(queries, k, rec) -> ticket and search_packed_end(ticket) -> bool."""
        nq = queries.shape[0]
        like = queries if queries.is_cuda else (torch.empty(0, device=device) if device is not None else queries)
        rec, gathered, redo = self._buffers(nq, k, like)
        redo.zero_()
        ticket = self.local.search_packed_begin(queries, k, rec)
        timed = rec.is_cuda and self.time_exchange
        if timed:   # the consumer stream already waits for the records: ev0 = records ready, ev1 = merged results written
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        res = self._gather_merge(rec, gathered, k, out, redo)
        if timed:
            self._ev[1].record()
        if rec.is_cuda:
            self._wait(rec.device)
        if timed:
            self.last_exchange_ms = self._ev[0].elapsed_time(self._ev[1])
        self.local.search_packed_end(ticket)
        if int(redo[0]) != 0:
            self.redo_batches += 1
            redo.zero_()
            res = self._gather_merge(rec, gathered, k, out, redo)
            if rec.is_cuda:
                self._wait(rec.device)
            if int(redo[0]) != 0:
                raise RuntimeError("records still provisional after search_packed_end")
        return res

    def search_begin(self, queries, k):
        """Pipelined form: the local shard search is enqueued now; wait() completes it, then runs
        the all-gather + merge. With two batches in flight the exchange of batch i overlaps the
        shard search of batch i+1 (which runs on its own stream inside the index)."""
        pending = self.local.search_begin(queries, k)
        outer = self

        class _Pending:
            def wait(self_inner):
                return outer._exchange(*pending.wait(), k)
        return _Pending()

    def _exchange(self, idx, score, k, out=None):
        """out = (ids, scores): optional result tensors of the device path (CUDA, or pinned CPU tensors the merge kernel
        fills in place - the caller synchronises the stream before reading them)."""
        if self.world == 1 and not self.force_collective:
            return idx, score
        nq = idx.shape[0]
        # ONE all-gather of nq packed records (k u64 ids + k f32 scores, 12 B per hit) per rank:
        # latency-bound (<= 1 MiB per rank at nq=8192, k=10), so a single collective.
        if idx.is_cuda and self._device_merge:
            from .cgvec import merge_packed, pack_topk
            rec = pack_topk(idx, score)                       # one kernel instead of cat + 2 slice copies
            key = (self.world,) + tuple(rec.shape)
            if self._gathered is None or tuple(self._gathered.shape) != key or self._gathered.device != rec.device:
                self._gathered = torch.empty(key, dtype=torch.int32, device=rec.device)   # reused across batches
            dist.all_gather_into_tensor(self._gathered, rec, group=self.group)
            return merge_packed(self._gathered, k, out=out)
        rec = torch.cat([idx.contiguous().view(torch.int32).reshape(nq, 2 * k),
                         score.contiguous().view(torch.int32).reshape(nq, k)], dim=1).contiguous()
        gathered = torch.empty((self.world, nq, 3 * k), dtype=torch.int32, device=rec.device)
        if dist.get_backend(self.group) == "gloo":  # CPU tests
            parts = [gathered[r] for r in range(self.world)]
            dist.all_gather(parts, rec, group=self.group)
        else:
            dist.all_gather_into_tensor(gathered, rec, group=self.group)
        g_idx = gathered[:, :, : 2 * k].contiguous().view(torch.int64).reshape(self.world, nq, k)
        g_score = gathered[:, :, 2 * k:].contiguous().view(torch.float32).reshape(self.world, nq, k)
        return self.merge(g_idx, g_score)
