"""Every BASELINE.json configuration at its FULL single-GPU size (VERDICT r1 'Missing' 3 / 'next' 3):

  C3  10M x 768 bf16, batch 4096, 8 GPUs   -> one GPU's shard: 1.25M rows, the whole 4096-query batch
  C4  1M x 1536 fp16 dot, batch 256, 1 GPU -> full size
  C5  500M x 768 fp8, batch 8192, 8 GPUs   -> one GPU's shard: 62.5M rows (48 GB of codes), batch 8192
(C1 and C2 run in test_gpu_parity.py.)

The CPU oracle cannot scan these corpora inside a test budget, so each case is verified by
  * size-independent properties: planted probes come back first, idempotence, sortedness, no fallback on random data;
  * an ORACLE-ANCHORED check of sampled queries that is complete, not statistical: (1) the reported scores are
    bit-equal to the oracle's arithmetic (simd_ops.rs:15-78 / :149-183) on the rows the library returns for those
    ids; (2) the exact device scan of the same query over ALL rows (cgv_batch_similarity_f32, itself pinned to the
    oracle in test_gpu_parity.py and spot-checked here) shows no row outside the reported set that beats the
    k-th reported (score, id)."""
import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu


def _anchored_check(ix, oracle, q_host, idx, sc, metric, odt, queries, rng):
    op = {"cosine": "cosine", "dot": "dot"}[metric]
    fn = oracle.cosine_adaptive if metric == "cosine" else oracle.dot_avx2
    n = len(ix)
    for qi in queries:
        qs = oracle.round_trip(q_host[qi], odt, fp8_codes=True) if odt == 3 else oracle.round_trip(q_host[qi], odt)
        ids = idx[qi].astype(np.int64)
        # (1) reported score == oracle arithmetic on the stored row behind each reported id
        for j, rid in enumerate(ids):
            row = ix.get_row(int(rid))
            if odt == 3:   # fp8 scores are defined on the e4m3 codes (row * 2^e): cosine is invariant under the row scale
                row = oracle.round_trip(row, 3, fp8_codes=True)
            assert np.float32(fn(qs, row)) == sc[qi, j], (qi, j, rid)
        # (2) completeness against the exact device scan of every row
        allsc = ix.batch_similarity(q_host[qi], op)
        assert allsc.shape == (n,)
        assert np.array_equal(allsc[ids], sc[qi])
        kth_s, kth_id = sc[qi, -1], ids[-1]
        better = np.nonzero((allsc > kth_s) | ((allsc == kth_s) & (np.arange(n) < kth_id)))[0]
        assert set(better.tolist()) == set(ids[:-1].tolist()), (qi, len(better))
        order = np.lexsort((ids, -sc[qi].astype(np.float64)))
        assert np.array_equal(order, np.arange(len(ids)))              # (score desc, id asc)
        # spot-check the exact device scan itself against the oracle on random rows
        for rid in rng.integers(0, n, 6):
            row = ix.get_row(int(rid))
            if odt == 3:
                row = oracle.round_trip(row, 3, fp8_codes=True)
            assert np.float32(fn(qs, row)) == allsc[rid]


def _fill(ix, n, d, chunk, seed, probe_chunks, per_chunk, unit=True, scale=None):
    """Append n rows generated on the device; remember `per_chunk` evenly spaced rows of the chunks in
    `probe_chunks` (probe vectors and the row ids they must come back as)."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    probes, want = [], []
    ix.reserve(n)
    for ci, lo in enumerate(range(0, n, chunk)):
        rows = min(chunk, n - lo)
        x = torch.randn((rows, d), generator=gen, device="cuda")
        if unit:
            x = torch.nn.functional.normalize(x, dim=1)
        elif scale is not None:
            x = x * scale
        if ci in probe_chunks:
            take = torch.arange(0, rows, max(1, rows // per_chunk), device="cuda")[:per_chunk]
            probes.append(x[take].clone())
            want.append(take + lo)
        ix.add(x)
    del x
    return torch.cat(probes), torch.cat(want).cpu()


def test_c3_shard_full_batch(oracle):
    """C3: one GPU's 1.25M-row shard against the whole 4096-query batch (16 query tiles x 16 corpus splits)."""
    import torch
    m = pkg()
    n, d, nq, k = 1_250_000, 768, 4096, 10
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        nch = n // 125_000
        probe, want = _fill(ix, n, d, 125_000, 0xC0DE6003, (0, nch // 2, nch - 1), 1400)
        probe, want = probe[:nq], want[:nq]
        ix.set_index_base(3 * n)                         # the shard of rank 3: ids are global
        idx, sc = ix.search(probe, k)
        idx2, sc2 = ix.search(probe, k)
        torch.cuda.synchronize()
        assert torch.equal(idx, idx2) and torch.equal(sc, sc2)
        assert torch.equal(idx[:, 0].cpu(), want + 3 * n)
        assert (sc[:, 0] > 0.999).all() and (sc[:, 1:] < 0.5).all() and (sc[:, :-1] >= sc[:, 1:]).all()
        st = ix.stats()
        assert st["last_path"] == 1 and st["fallback_queries"] == 0 and st["max_observed_err"] <= 0.5 * st["last_eps"]
        ix.set_index_base(0)
        rng = np.random.default_rng(3)
        qh = torch.nn.functional.normalize(torch.randn((nq, d), generator=torch.Generator(device="cuda").manual_seed(33),
                                                       device="cuda"), dim=1)
        gi, gs = ix.search(qh, k)
        _anchored_check(ix, oracle, qh.cpu().numpy(), gi.cpu().numpy().view(np.uint64), gs.cpu().numpy(), "cosine", 1,
                        [0, 255, 256, 2047, 4095], rng)
        assert ix.stats()["fallback_queries"] == 0
    finally:
        ix.close()


def test_c4_full_size_fp16_dot(oracle):
    """C4: 1M x 1536 fp16, dot product on UN-normalised rows (norms 0.5..2: the dot metric must not assume unit
    vectors), batch 256 - the HBM-bound shape (one query tile, 256 corpus splits)."""
    import torch
    m = pkg()
    n, d, nq, k = 1_000_000, 1536, 256, 10
    ix = m.HipKnnIndex(d, metric="dot", dtype="fp16")
    try:
        gen = torch.Generator(device="cuda").manual_seed(0xC0DE6004)
        ix.reserve(n)
        for lo in range(0, n, 125_000):
            x = torch.nn.functional.normalize(torch.randn((125_000, d), generator=gen, device="cuda"), dim=1)
            x = x * (0.5 + 1.5 * torch.rand((125_000, 1), generator=gen, device="cuda"))
            ix.add(x)
        del x
        q = torch.nn.functional.normalize(torch.randn((nq, d), generator=gen, device="cuda"), dim=1)
        idx, sc = ix.search(q, k)
        idx2, sc2 = ix.search(q, k)
        torch.cuda.synchronize()
        assert torch.equal(idx, idx2) and torch.equal(sc, sc2)
        assert (sc[:, :-1] >= sc[:, 1:]).all() and (idx >= 0).all() and (idx < n).all()
        st = ix.stats()
        assert st["last_path"] == 1 and st["fallback_queries"] == 0 and st["max_observed_err"] <= 0.5 * st["last_eps"]
        rng = np.random.default_rng(4)
        _anchored_check(ix, oracle, q.cpu().numpy(), idx.cpu().numpy().view(np.uint64), sc.cpu().numpy(), "dot", 2,
                        list(range(0, 256, 16)), rng)                  # 16 sampled queries
    finally:
        ix.close()


def test_c5_full_shard_fp8_batch_8192(oracle):
    """C5: one GPU's full shard - 62.5M x 768 fp8 (48 GB of e4m3 codes + norms / exponents) - against the whole
    8192-query batch. Probes planted at both ends and in the middle come back first; ids exceed 2^24 (nothing on
    the path may carry a row id in an f32); 288 GB sizing: device_bytes is reported."""
    import torch
    m = pkg()
    n, d, nq, k, chunk = 62_500_000, 768, 8192, 10, 250_000
    ix = m.HipKnnIndex(d, dtype="fp8")
    try:
        nch = n // chunk
        probe, want = _fill(ix, n, d, chunk, 0xC0DE6005, (0, nch // 2, nch - 1), 2800)
        probe, want = probe[:nq], want[:nq]
        assert probe.shape[0] == nq and int(want.max()) > 60_000_000 > (1 << 24)
        idx, sc = ix.search(probe, k)
        torch.cuda.synchronize()
        assert torch.equal(idx[:, 0].cpu(), want)
        assert (sc[:, 0] > 0.999).all() and (sc[:, 1:] < 0.5).all() and (sc[:, :-1] >= sc[:, 1:]).all()
        st = ix.stats()
        assert st["n_rows"] == n and st["device_bytes"] > n * d and st["last_path"] == 1
        assert st["fallback_queries"] == 0 and st["max_observed_err"] <= 0.5 * st["last_eps"]
        idx2, sc2 = ix.search(probe[:1024], k)                         # a smaller batch over the same shard: same answers
        assert torch.equal(idx2, idx[:1024]) and torch.equal(sc2, sc[:1024])
        rng = np.random.default_rng(5)
        qh = torch.nn.functional.normalize(torch.randn((64, d), generator=torch.Generator(device="cuda").manual_seed(55),
                                                       device="cuda"), dim=1)
        gi, gs = ix.search(qh, k)
        _anchored_check(ix, oracle, qh.cpu().numpy(), gi.cpu().numpy().view(np.uint64), gs.cpu().numpy(), "cosine", 3,
                        [0, 63], rng)
    finally:
        ix.close()


def test_c2_full_size_anchored_and_strong_scaling_shards(oracle):
    """C2 - the headline configuration (1M x 768 bf16, batch 1024) - with the ORACLE-ANCHORED check on 10 sampled queries
    (VERDICT r2 'weak' 9: the full-size C2 test had properties only), then the per-rank workloads of BASELINE's metric
    at 2 / 4 / 8 GPUs: the first 500 k / 250 k / 125 k rows of the same corpus as their own index with
    cgv_set_index_base (what bench.py --gpus N builds on every rank), full 1024-query batch. Each shard's top-k must be
    the restriction of the exact full scan to its rows: ids global, scores bit-equal, (score desc, id asc), no fallback;
    the plan is the sample launch + ONE emitting launch for 125 k / 250 k / 500 k rows and two for 1M."""
    import torch
    m = pkg()
    n, d, nq, k = 1_000_000, 768, 1024, 10
    gen = torch.Generator(device="cuda").manual_seed(0xC0DE6001)
    chunks = []
    for lo in range(0, n, 125_000):
        chunks.append(torch.nn.functional.normalize(torch.randn((125_000, d), generator=gen, device="cuda"), dim=1))
    q = torch.nn.functional.normalize(torch.randn((nq, d), generator=torch.Generator(device="cuda").manual_seed(0xC0DE6002),
                                                  device="cuda"), dim=1)
    qh = q.cpu().numpy()
    rng = np.random.default_rng(62)
    full = m.HipKnnIndex(d, dtype="bf16")
    try:
        full.reserve(n)
        for x in chunks:
            full.add(x)
        idx, sc = full.search(q, k)
        st = full.stats()
        assert st["last_path"] == 1 and st["fallback_queries"] == 0 and st["max_observed_err"] <= 0.5 * st["last_eps"]
        gi, gs = idx.cpu().numpy().view(np.uint64), sc.cpu().numpy()
        sampled = [0, 1, 255, 256, 511, 512, 700, 767, 768, 1023]
        _anchored_check(full, oracle, qh, gi, gs, "cosine", 1, sampled, rng)
        # exact scans of the sampled queries over all rows: the reference for every shard below
        exact = {qi: full.batch_similarity(qh[qi], "cosine") for qi in sampled}
        # the same queries as SMALL batches at full size (round 5: COARSE_TOP2 - the whole 1M-row corpus in one launch without
        # thresholds, per-cell top-2 + floor): bit-equal to their rows of the 1024-query batch's answer (oracle-anchored
        # above), pageable host buffers in and out, no fallback on random data
        import ctypes as C
        L = m.cgvec.lib()
        L.cgv_debug_last_top2_.argtypes = [C.c_void_p]
        L.cgv_debug_last_top2_.restype = C.c_int
        for lo, hi in ((0, 1), (255, 256), (0, 8), (500, 532), (960, 1024)):
            si, ss = full.search(qh[lo:hi], k)
            assert L.cgv_debug_last_top2_(full._h) == 1
            assert np.array_equal(si, gi[lo:hi]) and np.array_equal(ss, gs[lo:hi]), (lo, hi)
        # a small batch sends a query to the exact scan when three of its top-(k + 1) rows share one of the launch's
        # 1024 cells: measured 5e-4 per query on this corpus (scripts/top2_fallback_rate.py, 3 of 6000); the answers
        # above are equal either way, so the count is only bounded here
        assert full.stats()["fallback_queries"] <= 2
    finally:
        full.close()
    for rows, base in ((500_000, 0), (250_000, 250_000), (125_000, 875_000)):
        lo_chunk = base // 125_000
        ix = m.HipKnnIndex(d, dtype="bf16")
        try:
            ix.reserve(rows)
            for x in chunks[lo_chunk: lo_chunk + rows // 125_000]:
                ix.add(x)
            ix.set_index_base(base)              # rank r's shard reports GLOBAL ids
            si, ss = ix.search(q, k)
            st = ix.stats()
            assert st["last_path"] == 1 and st["fallback_queries"] == 0, (rows, st)
            si, ss = si.cpu().numpy().view(np.uint64), ss.cpu().numpy()
            assert (si >= base).all() and (si < base + rows).all()
            assert (ss[:, :-1] >= ss[:, 1:]).all()
            for qi in sampled:
                ref = exact[qi][base: base + rows]
                order = np.lexsort((np.arange(rows), -ref.astype(np.float64)))[:k]      # (score desc, id asc)
                assert np.array_equal(si[qi], (order + base).astype(np.uint64)), (rows, qi)
                assert np.array_equal(ss[qi], ref[order]), (rows, qi)
        finally:
            ix.close()
    del chunks


def test_c3_full_size_through_one_sharded_handle(oracle):
    """BASELINE config 3 at its REAL size and partition (VERDICT r3 'Missing' 2): 10M x 768 bf16 (15.4 GB - it fits one
    MI355X 18 times over), batch 4096, k = 10, row-sharded 8 ways through ONE cgv_sharded handle; on a 1-GPU box device 0 is
    listed 8 times (everything runs but the collective: the exchange is device copies), on an 8-GPU box the devices are
    distinct and the exchange is the in-library RCCL all-gather. Properties (planted probes come back first with GLOBAL ids
    across the block-cyclic map, idempotence, sortedness, zero fallbacks) + the oracle-anchored check of 8 sampled queries:
    (1) reported scores bit-equal to the oracle's arithmetic on the stored rows behind the reported ids; (2) completeness
    against the exact device scan of EVERY shard (cgv_batch_similarity_f32 on the shard handles, local -> global ids)."""
    import ctypes as C
    import torch
    m = pkg()
    L = m.cgvec.lib()
    n, d, nq, k, G, CH = 10_000_000, 768, 4096, 10, 8, 4096
    nd = m.device_count()
    sx = m.ShardedIndex(d, [i % nd for i in range(G)], dtype="bf16")
    try:
        sx.reserve(n)
        gen = torch.Generator(device="cuda").manual_seed(0xC0DE6033)
        chunk = 250_000
        probes, want = [], []
        for ci, lo in enumerate(range(0, n, chunk)):
            x = torch.nn.functional.normalize(torch.randn((chunk, d), generator=gen, device="cuda"), dim=1)
            if ci in (0, 7, 19, 39):                                  # probes from both ends and the middle
                take = torch.arange(0, chunk, chunk // 1024, device="cuda")[:1024]
                probes.append(x[take].clone())
                want.append((take + lo).cpu())
            sx.add(x.cpu().numpy())
            del x
        assert len(sx) == n
        cnt = sx.shard_counts()
        assert sum(cnt) == n and max(cnt) - min(cnt) <= CH
        probe = torch.cat(probes)[:nq].cpu().numpy()
        want = torch.cat(want)[:nq].numpy().astype(np.uint64)
        idx, sc = sx.search(probe, k)
        idx2, sc2 = sx.search(probe, k)
        assert np.array_equal(idx, idx2) and np.array_equal(sc, sc2)
        assert np.array_equal(idx[:, 0], want)                         # global ids across the block-cyclic map
        assert (sc[:, 0] > 0.999).all() and (sc[:, 1:] < 0.5).all() and (sc[:, :-1] >= sc[:, 1:]).all()
        st = sx.stats()
        assert st["fallback_queries"] == 0 and st["n_shards"] == G and st["last_exchange_ms"] > 0
        # ---- anchored check of sampled random queries ----
        rng = np.random.default_rng(33)
        qh = torch.nn.functional.normalize(torch.randn((nq, d), generator=torch.Generator(device="cuda").manual_seed(34),
                                                       device="cuda"), dim=1).cpu().numpy()
        gi, gs = sx.search(qh, k)
        assert sx.stats()["fallback_queries"] == 0
        for qi in (0, 255, 256, 1023, 2047, 2048, 3333, 4095):
            qs = oracle.round_trip(qh[qi], 1)
            ids = gi[qi].astype(np.int64)
            for j, rid in enumerate(ids):                               # (1) the oracle's arithmetic on the stored rows
                assert np.float32(oracle.cosine_adaptive(qs, sx.get_row(int(rid)))) == gs[qi, j], (qi, j, rid)
            allsc = np.empty(n, dtype=np.float32)                       # (2) exact device scan of every shard, global order
            for g in range(G):
                loc = np.empty(cnt[g], dtype=np.float32)
                m.cgvec._check(L.cgv_batch_similarity_f32(sx.shard_handle(g), qh[qi].ctypes.data_as(C.c_void_p), 0, 0,
                                                          loc.ctypes.data_as(C.c_void_p)))
                l = np.arange(cnt[g], dtype=np.int64)
                allsc[((l // CH) * G + g) * CH + l % CH] = loc
            assert np.array_equal(allsc[ids], gs[qi])
            kth_s, kth_id = gs[qi, -1], ids[-1]
            better = np.nonzero((allsc > kth_s) | ((allsc == kth_s) & (np.arange(n) < kth_id)))[0]
            assert set(better.tolist()) == set(ids[:-1].tolist()), (qi, len(better))
            assert np.array_equal(np.lexsort((ids, -gs[qi].astype(np.float64))), np.arange(k))
            for rid in rng.integers(0, n, 4):                           # spot-check the exact device scan itself
                assert np.float32(oracle.cosine_adaptive(qs, sx.get_row(int(rid)))) == allsc[rid]
    finally:
        sx.close()
