"""Row-sharded kNN across ranks (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm). SURVEY.md §8(e): the corpus is split into contiguous row ranges, every
rank searches its shard with the full query batch, the per-shard partial top-k
(nq*k (id, score) records per rank) are exchanged with ONE all-gather, and every rank
merges G*k candidates per query with (score desc, id asc). Exact: the global top-k is a
subset of the union of the local top-k.

The reference has no distributed path (no collective anywhere); this is the
multi-GPU extension BASELINE.json's north_star asks for.
"""
import torch
import torch.distributed as dist


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
