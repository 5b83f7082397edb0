"""Group commit of concurrent small callers (csrc/coalesce.h; VERDICT r5 'Next' 1).

The reference's trait-level call is ONE query (traits.rs:14 search_similar(&self, &[f32], limit); surreal_store.rs:61-85; caller
search.rs:114-117) and its multi-query caller issues B concurrent single-query searches (search.rs:358-361, try_join_all) from a
multi-thread runtime. Concurrent cgv_search_f32 calls of a few queries each are merged into one device batch; every caller must get
exactly what a lone call gives it: ids and scores bit-equal to the oracle's, its own k, its own status (a NaN query fails ITS
caller only)."""
import ctypes as C
import os
import subprocess
import threading
import uuid

import numpy as np
import pytest

from _util import ROOT, pkg

pytestmark = pytest.mark.gpu


def callers_lib():
    """tests/c_client/callers.c as a shared library: native caller threads (no interpreter lock between their calls)."""
    m = pkg()
    src = os.path.join(ROOT, "tests", "c_client", "callers.c")
    so = os.path.join(ROOT, "tests", "c_client", "libcgv_callers.so")
    libdir = os.path.dirname(m.cgvec.LIB_PATH)
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-fPIC", "-shared", "-pthread", "-I",
                               os.path.join(ROOT, "include"), src, "-o", so, "-L", libdir, "-lcgvec_hip",
                               "-Wl,-rpath,$ORIGIN/../../codegraph-rust_amd/lib"])
    m.cgvec.lib()
    L = C.CDLL(so)
    L.cgv_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_char_p]
    L.cgv_callers_run.restype = C.c_int
    return L


def run_callers(L, ix, q, k, threads, calls, warm=2, nq_per_call=1):
    ncalls = threads * calls
    oi = np.full((ncalls, nq_per_call, k), 7, dtype=np.uint64)
    os_ = np.full((ncalls, nq_per_call, k), -7, dtype=np.float32)
    lat = np.zeros(ncalls, dtype=np.float64)
    wall = C.c_double(0)
    err = C.create_string_buffer(256)
    rc = L.cgv_callers_run(ix._h, q.ctypes.data, q.shape[0], q.shape[1], k, threads, calls, warm, nq_per_call, oi.ctypes.data,
                           os_.ctypes.data, lat.ctypes.data, C.byref(wall), err)
    assert rc == 0, err.value
    return oi, os_, lat, wall.value


@pytest.mark.parametrize("dtype,odt,n,d", [("bf16", 1, 70_000, 256), ("f32", 0, 10_000, 384), ("fp8", 3, 40_000, 128)])
def test_native_callers_get_the_oracles_answers(oracle, dtype, odt, n, d):
    """64 native threads x 12 single-query calls on one index: most calls ride in shared device batches (COARSE_TOP2 for the
    MFMA indexes, the exact scan for f32 rows = BASELINE config 1's shape); every answer equals the oracle's bit for bit."""
    m = pkg()
    L = callers_lib()
    rng = np.random.default_rng(61)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows *= (0.5 + rng.random((n, 1))).astype(np.float32)
    T, calls, k = 64, 12, 10
    q = rng.standard_normal((T * calls, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        oi, os_, lat, wall = run_callers(L, ix, q, k, T, calls)
        st = ix.coalesce_stats()
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
        assert np.array_equal(oi[:, 0, :], ri)
        assert np.array_equal(os_[:, 0, :], rs)
        assert st["batches"] > 0 and st["batched_requests"] > st["batches"], st      # callers really shared batches
        assert st["max_batch_queries"] <= 64 and st["retried_alone"] == 0, st
        # switched off: every call is its own batch again, same answers
        ix.set_coalesce(0, 0, 0)
        oi2, os2, _, _ = run_callers(L, ix, q[: 8 * 4], k, 8, 4)
        st2 = ix.coalesce_stats()
        assert st2["batches"] == st["batches"] and st2["lone_calls"] == st["lone_calls"], (st, st2)
        assert np.array_equal(oi2[:, 0, :], ri[:32]) and np.array_equal(os2[:, 0, :], rs[:32])
    finally:
        ix.close()


def test_mixed_k_and_nq_and_a_nan_caller(oracle):
    """Callers with different k and nq share a batch (k = the largest, each gets its prefix); a caller whose query holds a NaN gets
    CGV_ERR_NONFINITE - it alone: its neighbours in the queue get their exact answers."""
    m = pkg()
    rng = np.random.default_rng(62)
    n, d = 50_000, 192
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype="bf16")
    T, rounds = 24, 10
    results, errors = {}, {}
    qs = {(t, r): rng.standard_normal((1 + (t + r) % 4, d)).astype(np.float32) for t in range(T) for r in range(rounds)}
    bad = {(5, 3), (5, 7), (11, 0)}
    for key in bad:
        qs[key][0, 7] = np.nan
    ks = [5, 10, 20, 50]
    gate = threading.Barrier(T)

    def caller(t):
        k = ks[t % len(ks)]
        gate.wait()
        for r in range(rounds):
            try:
                results[(t, r)] = ix.search(qs[(t, r)], k)
            except m.CgvError as e:
                errors[(t, r)] = e
    try:
        ix.add(rows)
        th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert set(errors) == bad, sorted(errors)
        for e in errors.values():
            assert e.code == m.cgvec.CGV_ERR_NONFINITE and "NaN" in str(e)
        for (t, r), (gi, gs) in results.items():
            k = ks[t % len(ks)]
            ri, rs = oracle.batch_top_k(qs[(t, r)], rows, k, dtype=1)
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs), (t, r, k)
        st = ix.coalesce_stats()
        assert st["batches"] > 0, st
        # the handle is fine afterwards
        gi, gs = ix.search(qs[(0, 0)], 10)
        ri, rs = oracle.batch_top_k(qs[(0, 0)], rows, 10, dtype=1)
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
    finally:
        ix.close()


def test_k_classes_do_not_share_a_batch(oracle):
    """A caller asking for k = 10 (one COARSE_TOP2 launch) is never dragged onto the path a k = 300 neighbour needs (the exact
    scan): batches carry one k class. Both get the oracle's answers."""
    m = pkg()
    rng = np.random.default_rng(63)
    n, d = 30_000, 128
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype="bf16")
    q = rng.standard_normal((16, d)).astype(np.float32)
    out = {}

    def caller(t):
        k = 300 if t % 4 == 0 else 10
        for r in range(4):
            out[(t, r)] = (k, ix.search(q[t:t + 1], k))
    try:
        ix.add(rows)
        th = [threading.Thread(target=caller, args=(t,)) for t in range(16)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for (t, r), (k, (gi, gs)) in out.items():
            ri, rs = oracle.batch_top_k(q[t:t + 1], rows, k, dtype=1)
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs), (t, r, k)
    finally:
        ix.close()


def test_64_threads_through_the_trait_surface(oracle):
    """cgvs_search_similar (VectorStore::search_similar, traits.rs:14 / surreal_store.rs:61-85) from 64 threads at once: the store's
    read entries hold its lock SHARED (host/store.cpp), so the callers reach cgv_search_f32 together and share device batches;
    an upsert in the middle runs exclusively and every search sees either the old or the new corpus, never a torn one."""
    m = pkg()
    rng = np.random.default_rng(64)
    n, d, k = 20_000, 384, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ids = [uuid.UUID(int=i + 1) for i in range(n)]
    vs = m.store.VectorStore(dtype="bf16")
    T, rounds = 64, 6
    q = rng.standard_normal((T * rounds, d)).astype(np.float32)
    got = {}
    gate = threading.Barrier(T)

    def caller(t):
        gate.wait()
        for r in range(rounds):
            got[(t, r)] = vs.search_similar(q[t * rounds + r], k)
    try:
        vs.store_embeddings(ids, rows)
        th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        ri, _ = oracle.batch_top_k(q, rows, k, dtype=1)
        for (t, r), hits in got.items():
            assert [h.int - 1 for h in hits] == [int(x) for x in ri[t * rounds + r]], (t, r)
        # a writer among the readers: 200 new rows, searched for afterwards
        extra = rng.standard_normal((200, d)).astype(np.float32)
        eids = [uuid.UUID(int=n + i + 1) for i in range(200)]
        got.clear()
        th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        vs.store_embeddings(eids, extra)
        for x in th:
            x.join()
        all_rows = np.concatenate([rows, extra])
        r_old, _ = oracle.batch_top_k(q, rows, k, dtype=1)
        r_new, _ = oracle.batch_top_k(q, all_rows, k, dtype=1)
        for (t, r), hits in got.items():
            g = [h.int - 1 for h in hits]
            j = t * rounds + r
            assert g == [int(x) for x in r_old[j]] or g == [int(x) for x in r_new[j]], (t, r)
        hits = vs.search_similar(extra[17], 1)
        assert hits[0].int - 1 == n + 17
    finally:
        vs.close()


def test_lone_caller_takes_the_plain_path():
    """One thread in a serial loop never waits for anybody: every call runs alone (no staging, no batch)."""
    m = pkg()
    rng = np.random.default_rng(65)
    rows = rng.standard_normal((20_000, 128)).astype(np.float32)
    ix = m.HipKnnIndex(128, dtype="bf16")
    try:
        ix.add(rows)
        for i in range(20):
            ix.search(rows[i:i + 1], 5)
        st = ix.coalesce_stats()
        assert st["lone_calls"] == 20 and st["batches"] == 0 and st["retried_alone"] == 0, st
        ix.search(rows[:9], 5)      # more than 8 queries per call: its own batch already, not counted
        assert ix.coalesce_stats()["lone_calls"] == 20
    finally:
        ix.close()
