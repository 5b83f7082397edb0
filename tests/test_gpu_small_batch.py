"""Small batches (nq <= 64) - the trait-level call is ONE query (traits.rs:14 search_similar(&self, &[f32], limit);
surreal_store.rs:61-85; caller search.rs:114-117; BASELINE config 1). Round 5: such batches take COARSE_TOP2 - the whole corpus
in ONE coarse launch without thresholds, every cell (workgroup, M-half, lane half) keeping its two best rows per query and the
best score it left out; final_kernel checks the guarantee against max(k'-th candidate, floor) - and exact-scan-only indexes
enqueue their scan without a host round trip. Results must be the oracle's, bit for bit, on both sides of every switch."""
import ctypes as C

import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu


def _last_top2(m, ix):
    L = m.cgvec.lib()
    L.cgv_debug_last_top2_.argtypes = [C.c_void_p]
    L.cgv_debug_last_top2_.restype = C.c_int
    return int(L.cgv_debug_last_top2_(ix._h))


@pytest.mark.parametrize("dtype,odt,metric", [("bf16", 1, "cosine"), ("bf16", 1, "dot"), ("fp16", 2, "cosine"), ("fp16", 2, "dot"),
                                               ("fp8", 3, "cosine"), ("f32s", 0, "cosine")])
def test_small_batches_equal_the_oracle(oracle, dtype, odt, metric):
    m = pkg()
    rng = np.random.default_rng(41)
    n, d, k = 23_456 + 77, 160, 10                      # ragged: the last tile is partial; D not a multiple of 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5000] = 0.0                                    # a zero row scores 0 (norm product 0), never NaN
    rows *= (0.5 + rng.random((n, 1))).astype(np.float32)   # un-normalised rows: the cells compare acc * invn_c
    ix = m.HipKnnIndex(d, dtype=dtype, metric=metric)
    omet = oracle.COSINE if metric == "cosine" else oracle.DOT
    try:
        ix.add(rows)
        fb0 = 0
        for nq in (1, 2, 7, 32, 33, 64, 65):
            q = rng.standard_normal((nq, d)).astype(np.float32)
            if nq >= 7:
                q[3] = 0.0                              # a zero query: every score 0 -> cannot be proven -> exact scan
            gi, gs = ix.search(q, k)
            ri, rs = oracle.batch_top_k(q, rows, k, metric=omet, dtype=odt)
            assert np.array_equal(gi, ri), (dtype, metric, nq)
            assert np.array_equal(gs, rs), (dtype, metric, nq)
            assert _last_top2(m, ix) == (1 if nq <= 64 else 0), nq
            st = ix.stats()
            assert st["last_path"] == 1
            assert st["fallback_queries"] - fb0 == (1 if nq >= 7 else 0), (nq, st["fallback_queries"], fb0)
            fb0 = st["fallback_queries"]
        # larger k on the path (k' <= 64) and just off it
        q = rng.standard_normal((5, d)).astype(np.float32)
        # (fp8 keeps k' = 2k + 12 candidates - its coarse error bound does not shrink with D, plan.cpp coarse_eps_scale (1') -
        # so k' <= 64 holds up to k = 26)
        for kk, on in (((26, 1), (27, 0)) if dtype == "fp8" else ((50, 1), (60, 0))):
            if dtype == "f32s" and kk > 12:
                continue                                # (f32 + shadow re-scores 4k + 16 candidates: k' <= 64 only up to k = 12)
            gi, gs = ix.search(q, kk)
            ri, rs = oracle.batch_top_k(q, rows, kk, metric=omet, dtype=odt)
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs), (dtype, kk)
            assert _last_top2(m, ix) == on, kk
    finally:
        ix.close()


def test_three_top_rows_in_one_cell_are_repaired_by_a_cell_rescan(oracle):
    """Rows 0..3 of a tile (M-block 0, registers 0..3 of lane half 0) belong to ONE cell: three near-identical copies of the
    query there leave the third out of the cell's top-2, the floor rises to its score and the guarantee check fails. Round 6:
    the offending cell alone is scanned again with the reference's arithmetic (kernels_repair.h) instead of the whole corpus -
    the other queries of the batch stay on the fast path. Identical rows (exact ties) too."""
    m = pkg()
    rng = np.random.default_rng(43)
    n, d, k = 30_000, 128, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    t = 57 * 256
    base = rng.standard_normal(d).astype(np.float32)
    for j in range(3):
        rows[t + j] = base * (1 + 1e-3 * (j + 1))       # same direction: cosine ties up to rounding, all above everything else
    rows[90 * 256 + 1] = rows[90 * 256 + 0]             # an exact tie inside a cell (two rows: both kept, nothing left out)
    q = rng.standard_normal((9, d)).astype(np.float32)
    q[4] = base
    q[6] = rows[90 * 256]
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.add(rows)
        gi, gs = ix.search(q, k)
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=1)
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        assert _last_top2(m, ix) == 1
        assert ix.stats()["fallback_queries"] == 1       # query 4 only failed its check ...
        sb = ix.small_batch_stats()
        assert sb["failed_queries"] == 1 and sb["repaired_by_cell_rescan"] == 1 and sb["exact_scans"] == 0, sb   # ... and was repaired
        assert set(gi[4][:3].tolist()) == {t, t + 1, t + 2}
        # the same call again (the context's flag words and repair buffers are reused), and one query at a time
        for _ in range(2):
            gi, gs = ix.search(q, k)
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        g1, s1 = ix.search(q[4], k)
        assert np.array_equal(g1[0], ri[4]) and np.array_equal(s1[0], rs[4])
        assert ix.small_batch_stats()["exact_scans"] == 0
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt,metric", [("bf16", 1, "cosine"), ("fp16", 2, "dot"), ("fp8", 3, "cosine"), ("f32s", 0, "cosine")])
def test_clustered_corpus_stays_off_the_exact_scan(oracle, dtype, odt, metric):
    """A code-embedding store inserts the chunks of one file next to each other and holds near-duplicates: here every
    'file' is 5 adjacent near-copies of one vector, and each query sits next to one file - 5 of its top-10 are adjacent rows, 4 of
    them in one cell. Most queries fail the COARSE_TOP2 check; all of them are answered by re-scanning a cell or two, none by the
    exact scan of the corpus, and every answer is the oracle's."""
    m = pkg()
    rng = np.random.default_rng(71)
    n_files, per, d, k = 8000, 5, 96, 10
    centers = rng.standard_normal((n_files, d)).astype(np.float32)
    rows = np.repeat(centers, per, axis=0) + 0.02 * rng.standard_normal((n_files * per, d)).astype(np.float32)
    rows = np.concatenate([rng.standard_normal((3, d)).astype(np.float32), rows])     # files do not start on a multiple of 4
    nq = 48
    picks = rng.choice(n_files, nq, replace=False)
    q = centers[picks] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    omet = oracle.COSINE if metric == "cosine" else oracle.DOT
    ix = m.HipKnnIndex(d, dtype=dtype, metric=metric)
    try:
        ix.add(rows)
        ri, rs = oracle.batch_top_k(q, rows, k, metric=omet, dtype=odt)
        gi, gs = ix.search(q, k)
        assert _last_top2(m, ix) == 1
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        sb = ix.small_batch_stats()
        assert sb["failed_queries"] >= nq // 2, sb               # the placement does what the test says it does
        assert sb["exact_scans"] == 0 and sb["repaired_by_cell_rescan"] == sb["failed_queries"], sb
        for j in (0, 7, 31):                                     # the trait-level call: one query
            g1, s1 = ix.search(q[j], k)
            assert np.array_equal(g1[0], ri[j]) and np.array_equal(s1[0], rs[j]), j
        assert ix.small_batch_stats()["exact_scans"] == 0
    finally:
        ix.close()


def test_small_batch_switch_is_invisible_in_the_results(oracle):
    """The same queries one by one, in small batches and inside a 300-query batch (staged thresholds): identical answers.
    Also: tiny corpora (<= 16 tiles: the dense boot path) and an index that grows across the switch."""
    m = pkg()
    rng = np.random.default_rng(47)
    d, k = 96, 10
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        rows = rng.standard_normal((3000, d)).astype(np.float32)
        ix.add(rows)
        q = rng.standard_normal((300, d)).astype(np.float32)
        gi, gs = ix.search(q[:4], k)
        assert _last_top2(m, ix) == 0                     # 12 tiles: boot path
        ri, rs = oracle.batch_top_k(q[:4], rows, k, dtype=1)
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        more = rng.standard_normal((40_000, d)).astype(np.float32)
        ix.add(more)
        allrows = np.concatenate([rows, more])
        ri, rs = oracle.batch_top_k(q, allrows, k, dtype=1)
        big_i, big_s = ix.search(q, k)
        assert _last_top2(m, ix) == 0 and np.array_equal(big_i, ri) and np.array_equal(big_s, rs)
        for lo, hi in ((0, 1), (1, 2), (2, 34), (34, 98), (98, 100)):
            gi, gs = ix.search(q[lo:hi], k)
            assert _last_top2(m, ix) == 1
            assert np.array_equal(gi, ri[lo:hi]) and np.array_equal(gs, rs[lo:hi]), (lo, hi)
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt", [("f32", 0), ("bf16", 1)])
def test_single_query_calls_through_pageable_and_pinned_buffers(oracle, dtype, odt):
    """BASELINE config 1's shape on the device: 10k x 384, ONE query per call, pageable host buffers (a Rust Vec<f32>) and
    pinned ones; the f32 index enqueues its exact scan behind the query conversion (one host synchronisation per call)."""
    import torch
    m = pkg()
    rng = np.random.default_rng(53)
    n, d, k = 10_000, 384, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        qs = rng.standard_normal((12, d)).astype(np.float32)
        ri, rs = oracle.batch_top_k(qs, rows, k, dtype=odt)
        for j in range(12):
            gi, gs = ix.search(qs[j], k)                  # pageable numpy buffers
            assert np.array_equal(gi[0], ri[j]) and np.array_equal(gs[0], rs[j]), j
        qp = torch.from_numpy(qs).pin_memory()
        oi = torch.empty((1, k), dtype=torch.int64).pin_memory()
        osc = torch.empty((1, k), dtype=torch.float32).pin_memory()
        for j in range(12):
            ix.search_host_ptr(qp[j].data_ptr(), 1, k, oi.data_ptr(), osc.data_ptr())
            assert np.array_equal(oi.numpy().view(np.uint64)[0], ri[j]) and np.array_equal(osc.numpy()[0], rs[j]), j
        bad = qs[0].copy()
        bad[7] = np.nan
        with pytest.raises(m.CgvError) as ei:
            ix.search(bad, k)
        assert ei.value.code == m.cgvec.CGV_ERR_NONFINITE
        gi, gs = ix.search(qs[1], k)                      # and the handle still answers
        assert np.array_equal(gi[0], ri[1]) and np.array_equal(gs[0], rs[1])
        gi, gs = ix.search(qs, 100)                       # large k (next_pow2 = 128 > 64: the LDS bitonic chunk reduction)
        r2i, r2s = oracle.batch_top_k(qs, rows, 100, dtype=odt)
        assert np.array_equal(gi, r2i) and np.array_equal(gs, r2s)
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt,metric", [("f32", 0, "cosine"), ("f32", 0, "dot"), ("bf16", 1, "cosine"), ("fp8", 3, "cosine")])
def test_exact_scan_as_one_kernel_on_every_side_of_its_switches(oracle, dtype, odt, metric):
    """The exact scan of <= 8 queries is ONE kernel (csrc/kernels_exact_small.h: scores, per-workgroup top-K by counting or by
    register extraction, the last workgroup's merge over the lists with the K largest heads, results, flags). Every size class:
    fewer rows than a pass of one workgroup, fewer than k, ragged last pass, > 256 keys per workgroup (extraction instead of
    counting), K = 2 .. 64 (K x K = 4096 candidate keys in the merge), 8 | 9 queries (one kernel | the staged reduction), 64 | 65 queries,
    duplicate rows (ties broken by the row index), a zero row and a zero query. Bit-exact against the oracle everywhere."""
    m = pkg()
    rng = np.random.default_rng(61)
    d = 72                                                # not a multiple of 64 (blocked layouts pad), a multiple of 8
    omet = oracle.COSINE if metric == "cosine" else oracle.DOT
    for n in (5, 31, 32, 33, 1000, 8193, 70_001):
        rows = rng.standard_normal((n, d)).astype(np.float32)
        if n > 40:
            rows[7] = 0.0
            rows[n - 1] = rows[3]                         # an exact tie across two workgroups' rows
            rows[n // 2] = rows[3]
        ix = m.HipKnnIndex(d, dtype=dtype, metric=metric)
        try:
            ix.add(rows)
            ix.set_force_exact(True)                      # (bf16 / fp8 indexes: the same scan the guarantee falls back to)
            for nq, k in ((1, 10), (1, 1), (3, 2), (8, 16), (9, 16), (2, 17), (1, 64), (4, 50), (64, 10), (65, 10)):
                q = rng.standard_normal((nq, d)).astype(np.float32)
                if nq >= 3:
                    q[1] = 0.0
                q[0] = rows[3] if n > 3 else q[0]
                gi, gs = ix.search(q, k)
                ri, rs = oracle.batch_top_k(q, rows, k, metric=omet, dtype=odt)
                assert np.array_equal(gi, ri), (dtype, metric, n, nq, k)
                assert np.array_equal(gs, rs), (dtype, metric, n, nq, k)
                assert ix.stats()["last_path"] == 0
        finally:
            ix.close()
