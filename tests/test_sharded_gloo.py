"""world_size-2 gloo test of the row-sharded path on CPU: shard ranges,
global ids, ONE all-gather of partial top-k, merge — with the oracle standing in for the
per-shard device search and merge (the HIP kernels are exercised by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleShard:
    def __init__(self, rows, base):
        self.rows, self.base = rows, base

    def search(self, queries, k):
        from oracle import oracle as o
        q = queries.numpy()
        idx = np.empty((q.shape[0], k), np.uint64)
        sc = np.empty((q.shape[0], k), np.float32)
        for i in range(q.shape[0]):
            if len(self.rows):
                idx[i], sc[i] = o.parallel_top_k(q[i], self.rows, k, threads=1)
                idx[i] = np.where(idx[i] == np.uint64(2**64 - 1), idx[i], idx[i] + np.uint64(self.base))
            else:
                idx[i], sc[i] = np.uint64(2**64 - 1), -np.inf
        return torch.from_numpy(idx.view(np.int64)), torch.from_numpy(sc)

    def search_begin(self, queries, k):   # same shape as HipKnnIndex.search_begin -> PendingSearch
        outer = self

        class _P:
            def wait(self_inner):
                return outer.search(queries, k)
        return _P()


def _oracle_merge(g_idx, g_score):
    from oracle import oracle as o
    G, nq, k = g_idx.shape
    oi = np.empty((nq, k), np.uint64)
    os_ = np.empty((nq, k), np.float32)
    gi = g_idx.numpy().view(np.uint64)
    gs = g_score.numpy()
    for q in range(nq):
        oi[q], os_[q] = o.merge_topk(gi[:, q, :], gs[:, q, :], k)
    return torch.from_numpy(oi.view(np.int64)), torch.from_numpy(os_)


def _worker(rank, world, port, n, d, nq, k, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = pkg()
    rng = np.random.default_rng(123)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5] = rows[n - 3]  # a cross-shard tie
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    lo, hi = m.shard_range(n, rank, world)
    sh = m.ShardedKnn(_OracleShard(rows[lo:hi], lo), merge=_oracle_merge)
    idx, sc = sh.search(torch.from_numpy(queries), k)
    # the pipelined form bench.py drives (two batches begun before the first is awaited)
    p1 = sh.search_begin(torch.from_numpy(queries), k)
    p2 = sh.search_begin(torch.from_numpy(queries[::-1].copy()), k)
    i1, s1 = p1.wait()
    i2, s2 = p2.wait()
    assert torch.equal(i1, idx) and torch.equal(s1, sc)
    assert torch.equal(i2, torch.flip(idx, dims=[0])) and torch.equal(s2, torch.flip(sc, dims=[0]))
    if rank == 0:
        np.save(out + ".idx.npy", idx.numpy().view(np.uint64))
        np.save(out + ".sc.npy", sc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    m = pkg()
    for n in (0, 1, 7, 8, 9, 1000, 1_000_000):
        for w in (1, 2, 3, 8):
            rs = [m.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert all(hi - lo <= (n + w - 1) // w for lo, hi in rs)


def test_two_rank_gloo_matches_single_process_oracle(tmp_path, oracle):
    n, d, nq, k = 1001, 48, 5, 10
    out = str(tmp_path / "r0")
    mp.spawn(_worker, args=(2, _free_port(), n, d, nq, k, out), nprocs=2, join=True)
    rng = np.random.default_rng(123)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5] = rows[n - 3]
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    ref_i, ref_s = oracle.batch_top_k(queries, rows, k, threads=1)
    assert np.array_equal(np.load(out + ".idx.npy"), ref_i)
    assert np.array_equal(np.load(out + ".sc.npy"), ref_s)
