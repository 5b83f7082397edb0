"""Behaviour at the edges of the single-device C ABI (ADVICE r1 + VERDICT r1): failed adds leave no trace,
k beyond the MFMA path's range, the begin/end pool cannot self-deadlock, per-hit scoring in one launch."""
import ctypes as C
import os

import mmap

import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu
_REGISTERED = []   # anonymous mappings handed to hipHostRegister: kept for the session


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


@pytest.mark.parametrize("dtype,odt", [("bf16", 1), ("f32s", 0), ("fp8", 3)])
def test_nonfinite_add_is_not_applied(oracle, dtype, odt):
    """A rejected cgv_add_f32 (NaN/Inf row) must not publish rows, poison the index, or block later adds."""
    m = pkg()
    rng = np.random.default_rng(1)
    n, d = 1000 + 13, 64                      # 13: the boundary 32-row block is partially filled
    rows = _unit(rng, n, d)
    q = _unit(rng, 20, d)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        bad = _unit(rng, 300, d) * 50.0        # large norms: must not leak into the max-norm statistic either
        bad[17, 3] = np.inf
        with pytest.raises(m.CgvError) as ei:
            ix.add(bad)
        assert ei.value.code == m.cgvec.CGV_ERR_NONFINITE and "not applied" in str(ei.value)
        assert len(ix) == n
        with pytest.raises(m.CgvError):
            ix.get_row(n)                      # the rejected rows are not addressable
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, rows, 10, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        more = _unit(rng, 500, d)
        ix.add(more)                           # ids continue at n
        allr = np.vstack([rows, more])
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, allr, 10, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        with pytest.raises(m.CgvError) as ei:  # update with a NaN row: rejected BEFORE the stored row is touched
            ix.update_row(5, np.full(d, np.nan, np.float32))
        assert ei.value.code == m.cgvec.CGV_ERR_NONFINITE
        assert np.array_equal(ix.get_row(5), oracle.round_trip(rows[5], odt))
        ix.truncate(n)                          # cgv_truncate: back to the first n rows
        assert len(ix) == n
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, rows, 10, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt,k", [("bf16", 1, 228), ("bf16", 1, 300), ("f32s", 0, 61), ("bf16", 1, 2048)])
def test_k_beyond_the_mfma_range_takes_the_exact_scan(oracle, dtype, odt, k):
    m = pkg()
    rng = np.random.default_rng(2)
    n, d, nq = 6000, 64, 7
    rows, q = _unit(rng, n, d), _unit(rng, nq, d)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        idx, sc = ix.search(q, k)
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        fast = k <= (228 if dtype == "bf16" else 60)
        assert ix.stats()["last_path"] == (1 if fast else 0)
        with pytest.raises(m.CgvError):
            ix.search(q, 2049)
    finally:
        ix.close()


def test_begin_without_end_reports_busy_instead_of_deadlocking():
    import torch
    m = pkg()
    rng = np.random.default_rng(3)
    ix = m.HipKnnIndex(32, dtype="bf16")
    try:
        ix.add(_unit(rng, 5000, 32))
        q = torch.from_numpy(_unit(rng, 16, 32)).cuda()
        pend = [ix.search_begin(q, 5) for _ in range(ix.max_in_flight)]
        with pytest.raises(m.CgvError) as ei:
            ix.search_begin(q, 5)
        assert ei.value.code == m.cgvec.CGV_ERR_BUSY
        with pytest.raises(m.CgvError) as ei:   # the host-pointer entry needs a context too
            ix.search(q.cpu().numpy(), 5)
        assert ei.value.code == m.cgvec.CGV_ERR_BUSY
        first = pend[0].wait()
        again = ix.search_begin(q, 5).wait()    # a released context is usable again
        assert torch.equal(first[0], again[0])
        for p in pend[1:]:
            p.wait()
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt", [("f32", 0), ("bf16", 1), ("fp16", 2)])
def test_score_ids_is_the_per_hit_rescore(oracle, dtype, odt):
    """cgv_score_ids_f32(CGV_OP_COSINE_SEQ) == search.rs:519-533 on (raw query, stored row), 0.0 for missing ids."""
    m = pkg()
    rng = np.random.default_rng(4)
    n, d, nq, mm = 3000, 100, 9, 37            # d = 100: the sequential formula has no lane structure to get wrong
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[50] = 0.0                              # zero norm -> 0.0
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ids = rng.integers(0, n, (nq, mm)).astype(np.uint64)
    ids[0, 0] = 50
    ids[1, 1] = np.uint64(2**64 - 1)           # padding
    ids[2, 2] = n + 7                          # beyond the index
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        got = ix.score_ids(q, ids, "cosine_seq")
        stored = oracle.round_trip(rows, odt)
        exp = np.zeros((nq, mm), np.float32)
        for a in range(nq):
            for b in range(mm):
                if ids[a, b] < n:
                    exp[a, b] = oracle.search_cosine(q[a], stored[int(ids[a, b])])
        assert np.array_equal(got, exp)
        assert got[0, 0] == 0.0 and got[1, 1] == 0.0 and got[2, 2] == 0.0
        got2 = ix.score_ids(q, ids, "cosine")   # the AVX2-order cosine on the same pairs
        for a, b in ((3, 3), (4, 10), (8, 36)):
            assert got2[a, b] == np.float32(oracle.cosine_adaptive(q[a], stored[int(ids[a, b])]))
    finally:
        ix.close()


def test_two_devices_in_one_process_get_their_kernel_attributes():
    """hipFuncAttributeMaxDynamicSharedMemorySize is per device: a second device's first search must work
    (VERDICT r1 weak 2). Skips below 2 GPUs."""
    m = pkg()
    if m.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    rng = np.random.default_rng(5)
    rows, q = _unit(rng, 20000, 128), _unit(rng, 64, 128)
    outs = []
    for dev in (0, 1):
        ix = m.HipKnnIndex(128, dtype="bf16", device=dev)
        ix.add(rows)
        outs.append(ix.search(q, 10))
        ix.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype,odt", [("bf16", 1), ("fp8", 3)])
def test_topic_sorted_corpus_keeps_the_fast_path(oracle, dtype, odt):
    """Insertion order must not matter for speed either: a corpus SORTED by topic (16 tight clusters, one after
    the other) with queries from every topic. If the first threshold came from the first 4096 rows (one topic),
    a query from another topic would see its whole cluster (6 % of the rows, cosine ~0.8) pass the first
    launches and overflow into the exact scan. The boot sample is spread over the corpus with the same
    golden-ratio stride as the later stages: no fallback, results bit-equal to the oracle."""
    m = pkg()
    rng = np.random.default_rng(404)
    d, ncl, per, nq, k = 256, 16, 12_500, 256, 10
    cent = rng.standard_normal((ncl, d)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    noise = rng.standard_normal((ncl * per, d)).astype(np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    rows = np.repeat(cent, per, axis=0) + np.float32(0.5) * noise          # sorted by topic
    qn = rng.standard_normal((nq, d)).astype(np.float32)
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    queries = cent[rng.integers(0, ncl, nq)] + np.float32(0.5) * qn
    ix = m.HipKnnIndex(d, metric="cosine", dtype=dtype)
    try:
        ix.add(rows)
        idx, sc = ix.search(queries, k)
        st = ix.stats()
        ri, rs = oracle.batch_top_k(queries, rows, k, metric=0, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        assert st["last_path"] == 1
        assert st["fallback_queries"] == 0, st
    finally:
        ix.close()


def test_fp8_magnitude_range_is_enforced(oracle):
    """fp8 rows / queries are accepted when their largest magnitude lies in [2^-48, 2^48] (the coarse pass
    accumulates de-scaled dot products in f32); outside it an add, an update or a search fails with
    CGV_ERR_INVALID_ARG and changes nothing - never a silently wrong answer. Inside the range, rows 2^80
    apart in magnitude still give the oracle's results."""
    m = pkg()
    rng = np.random.default_rng(9)
    n, d = 6000, 96
    rows = _unit(rng, n, d) * np.exp2(rng.integers(-40, 41, (n, 1))).astype(np.float32)
    q = _unit(rng, 8, d) * np.exp2(rng.integers(-40, 41, (8, 1))).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype="fp8")
    try:
        ix.add(rows)
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, rows, 10, dtype=3)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        for scale in (np.float32(2.0) ** 60, np.float32(2.0) ** -60):
            bad = _unit(rng, 40, d)
            bad[7] *= scale
            with pytest.raises(m.CgvError) as ei:
                ix.add(bad)
            assert ei.value.code == m.cgvec.CGV_ERR_INVALID_ARG and "not applied" in str(ei.value)
            assert len(ix) == n
            with pytest.raises(m.CgvError) as ei:
                ix.update_row(3, bad[7])
            assert ei.value.code == m.cgvec.CGV_ERR_INVALID_ARG
            assert np.array_equal(ix.get_row(3), oracle.round_trip(rows[3], 3))
            with pytest.raises(m.CgvError) as ei:
                ix.search(bad[6:8], 10)
            assert ei.value.code == m.cgvec.CGV_ERR_INVALID_ARG
        idx, sc = ix.search(q, 10)      # the index still answers, unchanged
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    finally:
        ix.close()


def test_pinned_host_buffers_are_used_in_place(oracle):
    """cgv_search_f32 reads pinned query buffers and writes pinned result buffers directly; pageable
    buffers go through staging. Every mix must give the oracle's answers - including the batches whose queries take
    the exact-scan fallback (it writes into the caller's pinned arrays too), a non-finite query (error, index intact),
    an f32 index (exact path only), and a registered (hipHostRegister) buffer.
    Runs IN THE SUITE'S PROCESS (round 3 moved it into a child after two unexplained aborts of the full suite in the test
    that followed it; HISTORY.md §9.5 has what the hunt for them found). tests/conftest.py hooks tests/c_client/abort_bt.c into
    the process, so a recurrence leaves the native stack of the raising thread in the log. CGV_PINNED_CHILD=1 restores the
    isolated form (a child that leaves through os._exit)."""
    if not os.environ.get("CGV_PINNED_CHILD"):
        _pinned_host_buffers_case(oracle)
        return
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import os, sys; sys.path.insert(0, %r); import test_gpu_robustness as t; from oracle import oracle as o; "
            "o.build(); t._pinned_host_buffers_case(o); print('pinned-case-ok', flush=True); os._exit(0)" % here)
    env = dict(os.environ, LIBC_FATAL_STDERR_="1")
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=600,
                       env=env, cwd=os.path.dirname(here))
    assert p.returncode == 0 and "pinned-case-ok" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


def _pinned_host_buffers_case(oracle):
    import torch
    m = pkg()
    rng = np.random.default_rng(91)
    n, d, nq, k = 30_000, 256, 300, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5000:5060] = rows[17] * (1 + 1e-4 * rng.standard_normal((60, 1)).astype(np.float32))   # a near-duplicate cluster
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[3] = rows[17]                                   # its query falls back to the exact scan
    for dtype, odt in (("bf16", 1), ("f32", 0), ("f32s", 0)):
        ix = m.HipKnnIndex(d, dtype=dtype)
        try:
            ix.add(rows)
            ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
            qp = torch.from_numpy(q).pin_memory()
            oi = torch.empty((nq, k), dtype=torch.int64).pin_memory()
            osc = torch.empty((nq, k), dtype=torch.float32).pin_memory()
            oi_pg = np.empty((nq, k), dtype=np.uint64)
            os_pg = np.empty((nq, k), dtype=np.float32)
            combos = [(qp.data_ptr(), oi.data_ptr(), osc.data_ptr(), lambda: (oi.numpy().view(np.uint64), osc.numpy())),
                      (q.ctypes.data, oi.data_ptr(), osc.data_ptr(), lambda: (oi.numpy().view(np.uint64), osc.numpy())),
                      (qp.data_ptr(), oi_pg.ctypes.data, os_pg.ctypes.data, lambda: (oi_pg, os_pg)),
                      (qp.data_ptr(), oi.data_ptr(), os_pg.ctypes.data, lambda: (oi.numpy().view(np.uint64), os_pg))]
            for qptr, iptr, sptr, get in combos:
                oi.zero_(); osc.zero_(); oi_pg[:] = 0; os_pg[:] = 0
                for _ in range(2):                    # twice: the flag words must come back clean for the second batch
                    ix.search_host_ptr(qptr, nq, k, iptr, sptr)
                gi, gs = get()
                assert np.array_equal(gi, ri) and np.array_equal(gs, rs), dtype
            if dtype == "bf16":
                assert ix.stats()["fallback_queries"] >= 8     # the cluster's query, every time
            bad = qp.clone().pin_memory()
            bad[7, 5] = float("nan")
            with pytest.raises(m.CgvError) as ei:
                ix.search_host_ptr(bad.data_ptr(), nq, k, oi.data_ptr(), osc.data_ptr())
            assert ei.value.code == m.cgvec.CGV_ERR_NONFINITE
            ix.search_host_ptr(qp.data_ptr(), nq, k, oi.data_ptr(), osc.data_ptr())     # and the index still answers
            assert np.array_equal(oi.numpy().view(np.uint64), ri) and np.array_equal(osc.numpy(), rs)
            # pinned queries in, DEVICE results out (what a rank of the row-sharded deployment does before the exchange)
            di, ds = ix.search_from_pinned(qp, k)
            assert np.array_equal(di.cpu().numpy().view(np.uint64), ri) and np.array_equal(ds.cpu().numpy(), rs)
            # registered pageable memory: the device alias may differ from the host address. The buffer is an anonymous
            # page-aligned mapping of its own (not a piece of the malloc heap, whose pages it would pin together with
            # whatever else lives on them) and stays mapped for the rest of the session (_REGISTERED): its address range is
            # never handed to another allocation while the HIP runtime may still remember it
            rt = torch.cuda.cudart()
            mm = mmap.mmap(-1, (q.nbytes + 4095) // 4096 * 4096)
            _REGISTERED.append(mm)
            reg = np.frombuffer(mm, dtype=np.float32, count=q.size).reshape(q.shape)
            reg[:] = q
            if hasattr(rt, "cudaHostRegister") and int(rt.cudaHostRegister(reg.ctypes.data, len(mm), 0)) == 0:
                try:
                    ix.search_host_ptr(reg.ctypes.data, nq, k, oi.data_ptr(), osc.data_ptr())
                    assert np.array_equal(oi.numpy().view(np.uint64), ri) and np.array_equal(osc.numpy(), rs)
                finally:
                    torch.cuda.synchronize()
                    assert int(rt.cudaHostUnregister(reg.ctypes.data)) == 0
        finally:
            ix.close()
