"""Parity of the HIP path (through the C ABI) against the CPU oracle, on a real MI355X.

Bar (BASELINE.json north_star): top-k row ids bit-exact under (score desc, id asc);
scores bit-exact too, because the re-score kernel reproduces the reference's f32 lane
order (tolerance stated where it is not zero)."""
import ctypes as C

import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu

ODT = {"f32": 0, "bf16": 1, "fp16": 2, "fp8": 3, "f32s": 0}   # f32s: f32 rows + bf16 shadow -> f32 oracle
OMETRIC = {"cosine": 0, "dot": 1, "cosine_seq": 2, "cosine_scalar": 3}


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def _check(oracle, rows, queries, k, dtype, metric, idx, sc, what=""):
    ref_i, ref_s = oracle.batch_top_k(queries, rows, k, metric=OMETRIC[metric], dtype=ODT[dtype])
    assert np.array_equal(idx, ref_i), f"{what}: ids differ\n{idx[:2]}\n{ref_i[:2]}"
    assert np.array_equal(sc, ref_s), f"{what}: scores differ (max abs {np.nanmax(np.abs(sc - ref_s))})"


def _run(oracle, n, d, nq, k, dtype, metric, seed=0, unit=True, mutate=None):
    m = pkg()
    rng = np.random.default_rng(seed)
    rows = _unit(rng, n, d) if unit else (rng.standard_normal((n, d)) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    queries = _unit(rng, nq, d) if unit else rng.standard_normal((nq, d)).astype(np.float32)
    if mutate:
        mutate(rows, queries)
    ix = m.HipKnnIndex(d, metric=metric, dtype=dtype)
    try:
        ix.add(rows)
        assert len(ix) == n
        idx, sc = ix.search(queries, k)
        _check(oracle, rows, queries, k, dtype, metric, idx, sc, f"n={n} d={d} nq={nq} k={k} {dtype} {metric}")
        return ix.stats()
    finally:
        ix.close()


def test_device_visible():
    assert pkg().device_count() >= 1


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp8", "f32s"])
@pytest.mark.parametrize("d", [64, 128, 768, 100])
def test_mfma_tile_mapping_dense_scores(dtype, d, oracle):
    """Dense coarse scores of the MFMA kernel vs an fp32 matmul of the rounded inputs
    (tolerance 2e-4 abs on cosine: fp32 accumulation-order noise only). Asymmetric data
    catches row/column swaps in the MFMA C layout."""
    import torch
    m = pkg()
    rng = np.random.default_rng(5)
    n, nq = 700, 300
    rows = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, metric="cosine", dtype=dtype)
    try:
        ix.add(rows)
        qd = torch.from_numpy(queries).cuda()
        got = ix.debug_coarse_scores(qd).cpu().numpy()
        if dtype == "fp8":   # e4m3 codes under the per-row power-of-two scale
            r = torch.from_numpy(oracle.round_trip(rows, 3, fp8_codes=True)).double()
            q = torch.from_numpy(oracle.round_trip(queries, 3, fp8_codes=True)).double()
        else:
            tdt = torch.float16 if dtype == "fp16" else torch.bfloat16   # f32s: the coarse pass runs on the bf16 shadow
            r = torch.from_numpy(rows).to(tdt).double()
            q = torch.from_numpy(queries).to(tdt).double()
        ref = (q / q.norm(dim=1, keepdim=True)) @ (r / r.norm(dim=1, keepdim=True)).T
        err = np.abs(got - ref.numpy())
        assert not np.isnan(got).any(), "unwritten entries in the dense dump"
        assert err.max() < 2e-4, f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,metric", [("bf16", "cosine"), ("fp16", "cosine"), ("bf16", "dot"),
                                          ("fp16", "dot"), ("f32", "cosine"), ("f32", "dot"),
                                          ("fp8", "cosine"), ("bf16", "cosine_seq"), ("f32", "cosine_seq"),
                                          ("fp8", "cosine_seq"), ("f32s", "cosine"), ("f32s", "dot"),
                                          ("f32s", "cosine_seq"), ("f32", "cosine_scalar"), ("bf16", "cosine_scalar"),
                                          ("fp8", "cosine_scalar"), ("f32s", "cosine_scalar")])
def test_parity_small(oracle, dtype, metric):
    _run(oracle, n=1000, d=256, nq=7, k=10, dtype=dtype, metric=metric)


@pytest.mark.parametrize("dtype", ["bf16", "f32", "fp8", "f32s"])
@pytest.mark.parametrize("d", [8, 31, 33, 37, 100, 384, 1536])
def test_parity_ragged_dims(oracle, dtype, d):
    """D < 32 takes the reference's scalar branch (simd_ops.rs:281-295); D % 8 != 0 the tail."""
    _run(oracle, n=777, d=d, nq=5, k=10, dtype=dtype, metric="cosine", seed=d, unit=False)


@pytest.mark.parametrize("dtype", ["bf16", "f32", "fp16", "f32s"])
@pytest.mark.parametrize("d", [8, 31, 32, 33, 100, 768])
def test_parity_scalar_host_mode(oracle, dtype, d):
    """CGV_METRIC_COSINE_SCALAR: cosine_similarity_scalar (simd_ops.rs:257-278) for EVERY length - what
    adaptive_cosine_similarity (:281-295) computes on a host without AVX2 + FMA and on every non-x86_64 host (the aarch64
    builds of the reference). Bit-exact ids and scores against the oracle's metric 3 (cgo_cosine_scalar per row, full sort)
    at D below, at and above the AVX2 path's 32-element switch; the MFMA path nominates, the scalar formula re-scores."""
    st = _run(oracle, n=3000, d=d, nq=9, k=10, dtype=dtype, metric="cosine_scalar", seed=100 + d, unit=False)
    if dtype != "f32":
        assert st["last_path"] == 1


def test_scalar_and_avx2_orders_really_differ(oracle):
    """The two modes are different arithmetic, not aliases: at D = 768 the scalar order's scores differ from the AVX2
    order's in the last bits for most pairs (if they were equal the mode could not be told from a no-op) - and each
    device mode matches ITS oracle function through the building-block API too (CGV_OP_COSINE_SCALAR)."""
    m = pkg()
    rng = np.random.default_rng(4)
    rows = rng.standard_normal((500, 768)).astype(np.float32)
    q = rng.standard_normal(768).astype(np.float32)
    ix = m.HipKnnIndex(768, dtype="f32")
    try:
        ix.add(rows)
        a = ix.batch_similarity(q, "cosine")
        b = ix.batch_similarity(q, "cosine_scalar")
        ra = np.array([oracle.cosine_adaptive(q, r) for r in rows], dtype=np.float32)
        rb = np.array([oracle.cosine_scalar(q, r) for r in rows], dtype=np.float32)
        assert np.array_equal(a, ra) and np.array_equal(b, rb)
        assert (a != b).mean() > 0.3
    finally:
        ix.close()


def test_parity_c1_shape_f32(oracle):
    """BASELINE config C1: 10k x 384 f32 cosine, single query."""
    st = _run(oracle, n=10_000, d=384, nq=1, k=10, dtype="f32", metric="cosine", seed=1)
    assert st["last_path"] == 0


def test_parity_c1_hash_embedder_corpus(oracle):
    """The reference's own synthetic 384-d generator (search.rs:178-205) over "node_{i}"."""
    m = pkg()
    rows = np.stack([oracle.hash_embed(f"node_{i}", 384) for i in range(10_000)])
    queries = np.stack([oracle.hash_embed(t, 384) for t in ("sum two numbers", "read a file", "node_17")])
    for dtype in ("f32", "bf16"):
        ix = m.HipKnnIndex(384, dtype=dtype)
        try:
            ix.add(rows)
            idx, sc = ix.search(queries, 10)
            _check(oracle, rows, queries, 10, dtype, "cosine", idx, sc, dtype)
            assert idx[2, 0] == 17
        finally:
            ix.close()


@pytest.mark.parametrize("dtype,metric", [("bf16", "cosine"), ("fp16", "dot"), ("fp8", "cosine")])
def test_parity_medium_staged(oracle, dtype, metric):
    """70k rows -> 274 corpus tiles: exercises all three threshold stages, several
    query tiles (nq=300) and the strided sample."""
    st = _run(oracle, n=70_000, d=1536 // 4 if dtype == "fp16" else 768, nq=300, k=10, dtype=dtype, metric=metric, seed=3)
    assert st["last_path"] == 1
    assert st["fallback_queries"] == 0
    assert st["max_observed_err"] <= st["last_eps"]


def test_fp8_mixed_row_scales_staged(oracle):
    """fp8 rows and queries whose power-of-two scales differ by up to 2^40 inside every 32-row block: the
    coarse kernel applies the scales inside the MFMA (de-scaled accumulators), so all three stages and the
    norm-bound filter see the rows' true magnitudes. Bit-exact, fast path, no fallback."""
    def mutate(rows, queries):
        rng = np.random.default_rng(77)
        rows *= np.exp2(rng.integers(-20, 21, (rows.shape[0], 1))).astype(np.float32)
        queries *= np.exp2(rng.integers(-20, 21, (queries.shape[0], 1))).astype(np.float32)
    st = _run(oracle, n=70_000, d=512, nq=300, k=10, dtype="fp8", metric="cosine", seed=13, mutate=mutate)
    assert st["last_path"] == 1
    assert st["fallback_queries"] == 0
    assert st["max_observed_err"] <= st["last_eps"]


def test_reference_epilogue_variant(oracle):
    """CGV_EPI=0: the round-2 epilogue of the bf16 coarse kernel (everything at the tile boundary), kept in the MEASUREMENT
    flavour of the library (`make ABLATE=1`, libcgvec_hip_ablate.so) as the A/B reference of the interleaved one - same
    candidates, same bit-exact results, no fallback. A child process that loads that flavour (CGV_LIB_PATH): the production
    library carries neither the instantiation nor the switch."""
    import os
    import subprocess
    import sys
    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from _util import pkg
from oracle import oracle as o
o.build()
m = pkg()
rng = np.random.default_rng(21)
n, d, nq, k = 40_000, 1024, 300, 10
rows = rng.standard_normal((n, d)).astype(np.float32)
queries = rng.standard_normal((nq, d)).astype(np.float32)
ix = m.HipKnnIndex(d, metric="cosine", dtype="bf16")
ix.add(rows)
idx, sc = ix.search(queries, k)
st = ix.stats()
ri, rs = o.batch_top_k(queries, rows, k, metric=0, dtype=1)
assert np.array_equal(idx, ri) and np.array_equal(sc, rs), "results differ from the oracle"
assert st["last_path"] == 1 and st["fallback_queries"] == 0, st
print("EPI0-OK")
"""
    abl = pkg().cgvec.build_library(ablate=True)      # a no-op when __graft_entry__.build() made it (it travels with the tree)
    env = dict(os.environ, CGV_EPI="0", CGV_LIB_PATH=abl)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "EPI0-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_f32_index_with_bf16_shadow(oracle, metric):
    """CGV_DTYPE_F32_SHADOW: results are the reference's f32 arithmetic on the UNROUNDED inputs (the same
    as an f32 index), found through the MFMA coarse pass over a bf16 copy; the guarantee check carries
    the copy's rounding residual. Staged launches, un-normalised rows, update_row, get_row."""
    m = pkg()
    rng = np.random.default_rng(31)
    n, d, nq, k = 60_000, 384, 200, 10
    rows = (rng.standard_normal((n, d)) * rng.uniform(0.3, 2.0, (n, 1))).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, metric=metric, dtype="f32s")
    try:
        ix.add(rows[:25_000])
        ix.add(rows[25_000:])
        idx, sc = ix.search(q, k)
        _check(oracle, rows, q, k, "f32", metric, idx, sc, "f32 + shadow")
        st = ix.stats()
        assert st["last_path"] == 1 and st["last_kprime"] == 56
        assert st["fallback_queries"] <= nq // 10                     # the bound is loose but not useless
        assert 0 < st["max_observed_err"] <= st["last_eps"]           # bf16 rounding error, within the bound
        assert np.array_equal(ix.get_row(4321), rows[4321])           # rows are stored unrounded
        rows[77] = q[3] * np.float32(0.5)
        ix.update_row(77, rows[77])
        idx, sc = ix.search(q[:8], k)
        _check(oracle, rows, q[:8], k, "f32", metric, idx, sc, "f32 + shadow after update_row")
        if metric == "cosine":
            assert idx[3, 0] == 77
        # same answers as the plain f32 index (exact scan)
        ix2 = m.HipKnnIndex(d, metric=metric, dtype="f32")
        ix2.add(rows)
        i2, s2 = ix2.search(q[:8], k)
        ix2.close()
        assert np.array_equal(idx, i2) and np.array_equal(sc, s2)
    finally:
        ix.close()


def test_parity_sequential_cosine_metric_staged(oracle):
    """CGV_METRIC_COSINE_SEQ: the sequential cosine of search.rs:519-533 / indexer.rs:2965-2979 as the
    exact arithmetic behind the MFMA coarse pass (un-normalised rows, several stages)."""
    st = _run(oracle, n=40_000, d=384, nq=130, k=10, dtype="bf16", metric="cosine_seq", seed=14, unit=False)
    assert st["last_path"] == 1 and st["fallback_queries"] == 0
    assert st["max_observed_err"] <= st["last_eps"]


def test_parity_unnormalised_rows_and_prefetch_k(oracle):
    """Chunk-mean rows are not unit length; k=30 is search.rs:113's prefetch for limit 10."""
    _run(oracle, n=20_000, d=384, nq=40, k=30, dtype="bf16", metric="cosine", seed=9, unit=False)


@pytest.mark.parametrize("dtype", ["bf16", "f32s", "fp8"])
def test_ties_duplicates_zero_rows_zero_query(oracle, dtype):
    def mutate(rows, queries):
        rows[100:140] = rows[7]          # 40 exact duplicates of the best match for query 0
        queries[0] = rows[7]
        rows[300:310] = 0.0              # zero rows -> score 0.0 (simd_ops.rs:73-74)
        queries[1] = 0.0                 # zero query -> every score 0.0, ids 0..k-1
    st = _run(oracle, n=5000, d=128, nq=4, k=10, dtype=dtype, metric="cosine", seed=4, mutate=mutate)
    assert st["fallback_queries"] >= 1   # ties at the candidate boundary are resolved by the exact scan


def test_fewer_rows_than_k_and_single_row(oracle):
    m = pkg()
    for n in (1, 3, 9):
        for dtype in ("bf16", "f32s"):
            _run(oracle, n=n, d=64, nq=3, k=10, dtype=dtype, metric="cosine", seed=n)
    ix = m.HipKnnIndex(64)
    try:
        idx, sc = ix.search(np.ones((2, 64), np.float32), 5)     # empty index
        assert np.all(idx == np.uint64(2**64 - 1)) and np.all(np.isneginf(sc))
        i0, s0 = ix.search(np.ones((0, 64), np.float32), 5)      # nq == 0 (surreal_store.rs:62-64)
        assert i0.shape == (0, 5)
    finally:
        ix.close()


def test_incremental_add_equals_bulk_add(oracle):
    m = pkg()
    rng = np.random.default_rng(12)
    rows, queries = _unit(rng, 3000, 96), _unit(rng, 6, 96)
    ix = m.HipKnnIndex(96, dtype="bf16")
    try:
        for lo in range(0, 3000, 701):   # unaligned appends (norm-block stats must be redone)
            ix.add(rows[lo:lo + 701])
        idx, sc = ix.search(queries, 10)
        _check(oracle, rows, queries, 10, "bf16", "cosine", idx, sc, "incremental")
        assert np.array_equal(ix.get_row(1234), oracle.round_trip(rows[1234], 1))
    finally:
        ix.close()


def test_fp8_storage_semantics(oracle):
    """BASELINE config C5's storage: e4m3fn codes under a per-row power-of-two scale
    (SURVEY.md §7 'fp8 dynamic range'). get_row returns code * 2^-e; rows of wildly different
    magnitude, zero rows and duplicates keep the (score desc, id asc) order of the oracle;
    dot is refused (the per-row scale is not carried into a dot product)."""
    m = pkg()
    rng = np.random.default_rng(33)
    d = 200
    rows = _unit(rng, 6000, d) * np.exp2(rng.integers(-20, 20, (6000, 1))).astype(np.float32)
    rows[50:60] = 0.0
    rows[700:720] = rows[3] * np.float32(4.0)     # same direction, other scale -> identical codes
    queries = _unit(rng, 12, d)
    queries[0] = rows[3]
    queries[1] = 0.0
    ix = m.HipKnnIndex(d, dtype="fp8")
    try:
        ix.add(rows[:2500])
        ix.add(rows[2500:])
        idx, sc = ix.search(queries, 25)
        _check(oracle, rows, queries, 25, "fp8", "cosine", idx, sc, "fp8 semantics")
        assert idx[0, :21].tolist() == [3] + list(range(700, 720))
        for r in (0, 3, 55, 4321):
            assert np.array_equal(ix.get_row(r), oracle.round_trip(rows[r], 3))
        ix.update_row(10, rows[3])
        idx2, _ = ix.search(queries[:1], 25)
        assert idx2[0, :2].tolist() == [3, 10]
        cur = oracle.round_trip(np.vstack([rows[:10], rows[3:4], rows[11:]]), 3, fp8_codes=True)
        q2 = oracle.round_trip(queries[2], 3, fp8_codes=True)
        for op, fn in (("cosine", oracle.cosine_adaptive), ("cosine_seq", oracle.search_cosine),
                       ("cosine_distance_seq", oracle.cosine_distance)):
            ref = np.array([fn(q2, r) for r in cur[:400]], np.float32)
            assert np.array_equal(ix.batch_similarity(queries[2], op=op, limit_rows=400), ref), op
        with pytest.raises(m.CgvError):
            ix.batch_similarity(queries[2], op="dot")
    finally:
        ix.close()
    with pytest.raises(m.CgvError):
        m.HipKnnIndex(d, metric="dot", dtype="fp8")


def test_kat_parallel_operations_on_gpu(oracle):
    """simd_ops.rs:462-472: ones[256] vs ramp rows, k=10 -> rows 999..990."""
    m = pkg()
    q = np.ones((1, 256), np.float32)
    rows = (np.arange(1000)[:, None] + np.arange(256)[None, :]).astype(np.float32)
    for dtype in ("f32", "bf16"):
        ix = m.HipKnnIndex(256, dtype=dtype)
        try:
            ix.add(rows)
            idx, sc = ix.search(q, 10)
            _check(oracle, rows, q, 10, dtype, "cosine", idx, sc, "KAT " + dtype)
            if dtype == "f32":
                assert idx[0].tolist() == list(range(999, 989, -1))
        finally:
            ix.close()


def test_forced_exact_path_matches(oracle):
    m = pkg()
    rng = np.random.default_rng(21)
    rows, queries = _unit(rng, 9000, 200), _unit(rng, 9, 200)
    ix = m.HipKnnIndex(200, dtype="bf16")
    try:
        ix.add(rows)
        ix.set_force_exact(True)
        idx, sc = ix.search(queries, 17)
        assert ix.stats()["last_path"] == 0
        _check(oracle, rows, queries, 17, "bf16", "cosine", idx, sc, "forced exact")
    finally:
        ix.close()


def test_nonfinite_inputs_are_errors():
    """The reference panics on NaN (simd_ops.rs:379); across the FFI that is a status, not an abort."""
    m = pkg()
    ix = m.HipKnnIndex(32)
    try:
        ix.add(np.ones((10, 32), np.float32))
        q = np.ones((1, 32), np.float32)
        q[0, 3] = np.nan
        with pytest.raises(m.CgvError) as ei:
            ix.search(q, 3)
        assert ei.value.code == m.cgvec.CGV_ERR_NONFINITE
        with pytest.raises(m.CgvError):
            ix.search(np.ones((1, 31), np.float32), 3)      # dimension mismatch (simd_ops.rs:16-18)
        bad = np.ones((2, 32), np.float32)
        bad[1, 0] = np.inf
        with pytest.raises(m.CgvError):
            ix.add(bad)
    finally:
        ix.close()


def test_device_pointer_api_and_merge(oracle):
    import torch
    m = pkg()
    rng = np.random.default_rng(33)
    rows, queries = _unit(rng, 6000, 128), _unit(rng, 33, 128)
    k = 10
    shards = []
    try:
        outs_i, outs_s = [], []
        for r in range(3):
            lo, hi = m.shard_range(6000, r, 3)
            ix = m.HipKnnIndex(128, dtype="bf16")
            shards.append(ix)
            ix.add(torch.from_numpy(rows[lo:hi]).cuda())
            ix.set_index_base(lo)
            i, s = ix.search(torch.from_numpy(queries).cuda(), k)
            outs_i.append(i)
            outs_s.append(s)
        gi, gs = m.merge_topk(torch.stack(outs_i), torch.stack(outs_s))
        torch.cuda.synchronize()
        _check(oracle, rows, queries, k, "bf16", "cosine", gi.cpu().numpy().view(np.uint64), gs.cpu().numpy(), "3-shard merge")
    finally:
        for ix in shards:
            ix.close()


def test_golden_fixture(oracle):
    import os
    m = pkg()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_small.npz"))
    for dtype in ("f32", "bf16", "fp16"):
        ix = m.HipKnnIndex(int(g["rows"].shape[1]), dtype=dtype)
        try:
            ix.add(g["rows"])
            idx, sc = ix.search(g["queries"], int(g["k"]))
            assert np.array_equal(idx, g[f"idx_{dtype}"])
            assert np.array_equal(sc, g[f"score_{dtype}"])
        finally:
            ix.close()


def test_full_size_properties_c2():
    """BASELINE config C2 (1M x 768 bf16, batch 1024) through size-independent properties:
    a stored row queried against the corpus returns itself first with score ~1; repeated
    search is idempotent; scores are sorted; no fallback on random data."""
    import torch
    m = pkg()
    n, d, nq, k = 1_000_000, 768, 1024, 10
    gen = torch.Generator(device="cuda").manual_seed(0xC0DE6001)
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.reserve(n)
        for lo in range(0, n, 250_000):
            x = torch.randn((250_000, d), generator=gen, device="cuda")
            x = torch.nn.functional.normalize(x, dim=1)
            if lo == 0:
                probe = x[:nq].clone()
            ix.add(x)
        del x
        idx, sc = ix.search(probe, k)
        idx2, sc2 = ix.search(probe, k)
        torch.cuda.synchronize()
        assert torch.equal(idx, idx2) and torch.equal(sc, sc2)
        assert torch.equal(idx[:, 0].cpu(), torch.arange(nq))
        assert (sc[:, 0] > 0.999).all() and (sc[:, 1:] < 0.5).all()
        assert (sc[:, :-1] >= sc[:, 1:]).all()
        st = ix.stats()
        assert st["fallback_queries"] == 0 and st["max_observed_err"] <= st["last_eps"]
    finally:
        ix.close()


# (BASELINE configs C3, C4 and C5 at their full single-GPU sizes: tests/test_gpu_configs.py)


@pytest.mark.parametrize("dtype,odt", [("f32", 0), ("bf16", 1)])
@pytest.mark.parametrize("d", [24, 100, 384])
def test_building_blocks_batch_similarity(oracle, dtype, odt, d):
    """parallel_batch_similarity with each of the reference's kernels, and
    compute_distances_cpu's 'first limit rows' semantics — bit-exact."""
    m = pkg()
    rng = np.random.default_rng(d)
    rows = rng.standard_normal((300, d)).astype(np.float32)
    rows[17] = 0.0
    q = rng.standard_normal(d).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        rs, qs = oracle.round_trip(rows, odt), oracle.round_trip(q, odt)
        for op, fn in (("cosine", oracle.cosine_adaptive), ("dot", oracle.dot_avx2), ("l2", oracle.l2_avx2),
                       ("cosine_seq", oracle.search_cosine), ("cosine_distance_seq", oracle.cosine_distance)):
            got = ix.batch_similarity(q, op)
            ref = np.array([fn(qs, r) for r in rs], np.float32)
            assert np.array_equal(got, ref), op
        first = ix.batch_similarity(q, "cosine_distance_seq", limit_rows=10)      # gpu.rs:297-322
        assert first.shape == (10,) and np.array_equal(first, ix.batch_similarity(q, "cosine_distance_seq")[:10])
    finally:
        ix.close()


def test_search_baseline_o1(oracle):
    """optimization.rs:376-402: ascending cosine_distance, stable."""
    m = pkg()
    rng = np.random.default_rng(77)
    rows = rng.standard_normal((5000, 128)).astype(np.float32)
    rows[900] = rows[30]          # exact duplicate -> stable order keeps 30 before 900
    rows[55] = 0.0                # zero norm -> +inf distance, sorts last
    q = rows[30] + 0.01 * rng.standard_normal(128).astype(np.float32)
    ix = m.HipKnnIndex(128, dtype="f32")
    try:
        ix.add(rows)
        for limit in (1, 10, 40):
            gi, gd = ix.search_baseline(q, limit)
            ri, rd = oracle.search_baseline(q, rows, limit)
            assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    finally:
        ix.close()


def test_normalize_rows_on_device(oracle):
    """simd_ops.rs:189-222 / 386-419 (reciprocal multiply; zero rows untouched)."""
    m = pkg()
    rng = np.random.default_rng(3)
    for d in (7, 64, 100, 768):
        rows = (rng.standard_normal((50, d)) * 3).astype(np.float32)
        rows[4] = 0.0
        assert np.array_equal(m.cgvec.normalize_rows(rows), oracle.normalize_rows(rows)), d


def test_normalize_rows_scalar_arm_on_device(oracle):
    """simd_ops.rs:394-403 / 406-415: the arm of parallel_normalize_vectors a host without AVX2 + FMA takes - squares
    summed in order by one accumulator, `> 0.0` (a NaN sum leaves the row alone), a DIVIDE per element. Bit-exact against
    the oracle for D = 8 .. 768, ragged row counts (the kernel stages 64 x 64 tiles), zero rows and a NaN row; the two arms
    differ on ordinary data (reciprocal-multiply vs divide), so the test would notice one standing in for the other."""
    m = pkg()
    rng = np.random.default_rng(5)
    differ = 0
    for d in (1, 7, 8, 63, 64, 65, 100, 384, 768):
        for n in (1, 50, 64, 257):
            rows = (rng.standard_normal((n, d)) * 3).astype(np.float32)
            if n > 4:
                rows[4] = 0.0
                rows[2, d // 2] = np.nan
            got = m.cgvec.normalize_rows(rows, arm="scalar")
            want = oracle.normalize_rows(rows, arm="scalar")
            assert np.array_equal(got, want, equal_nan=True), (d, n)
            if n > 4:
                assert np.array_equal(got[2], rows[2], equal_nan=True)   # NaN sum: `nsq > 0.0` is false, row untouched
            differ += int(not np.array_equal(got, oracle.normalize_rows(rows, arm="avx2"), equal_nan=True))
    assert differ > 0


def test_batches_in_flight_and_concurrent_callers(oracle):
    """cgv_search_begin/_end: more batches begun than the pool holds contexts (begin blocks until a
    context frees up from another thread's end), different nq / k per batch, results equal to the
    oracle; a writer (add) issued while searches are in flight waits for them; SURVEY.md §8(b)
    threading: concurrent callers on one handle."""
    import threading
    import torch
    m = pkg()
    rng = np.random.default_rng(91)
    rows, d = _unit(rng, 40_000, 128), 128
    batches = [(_unit(rng, nq, d), k) for nq, k in ((300, 10), (17, 30), (1, 10), (513, 5), (64, 10), (256, 16))]
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.add(rows[:30_000])
        ix.add(rows[30_000:])
        refs = [oracle.batch_top_k(q, rows, k, dtype=1) for q, k in batches]
        # (1) pipelined on one thread, two in flight
        pend, outs = [], []
        for q, k in batches:
            pend.append(ix.search_begin(torch.from_numpy(q).cuda(), k))
            if len(pend) == 2:
                outs.append(pend.pop(0).wait())
        outs += [p.wait() for p in pend]
        for (gi, gs), (ri, rs) in zip(outs, refs):
            assert np.array_equal(gi.cpu().numpy().view(np.uint64), ri) and np.array_equal(gs.cpu().numpy(), rs)
        # (2) concurrent callers, each with its own torch stream
        errs, res = [], [None] * len(batches)

        def worker(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    q, k = batches[i]
                    for _ in range(3):
                        gi, gs = ix.search(torch.from_numpy(q).cuda(), k)
                    res[i] = (gi.cpu().numpy().view(np.uint64), gs.cpu().numpy())
            except Exception as e:   # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(batches))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for (gi, gs), (ri, rs) in zip(res, refs):
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        # (3) a writer while a batch is in flight: the add must wait, the batch sees the old corpus
        q, k = batches[0]
        p = ix.search_begin(torch.from_numpy(q).cuda(), k)
        extra = _unit(rng, 500, d)
        t = threading.Thread(target=lambda: ix.add(extra))
        t.start()
        gi, gs = p.wait()
        t.join()
        assert np.array_equal(gi.cpu().numpy().view(np.uint64), refs[0][0])
        assert len(ix) == 40_500
        gi2, gs2 = ix.search(q, k)
        ri2, rs2 = oracle.batch_top_k(q, np.vstack([rows, extra]), k, dtype=1)
        assert np.array_equal(gi2, ri2) and np.array_equal(gs2, rs2)
    finally:
        ix.close()


@pytest.mark.parametrize("k", [10, 7])
def test_packed_exchange_kernels(oracle, k):
    """cgv_pack_topk_dev + cgv_merge_packed_dev (the sharded path's single all-gather buffer) against the
    oracle's merge, including padding entries, cross-shard ties and an odd k (padded record rows)."""
    import torch
    m = pkg()
    rng = np.random.default_rng(k)
    g, nq = 4, 33
    idx = rng.integers(0, 1 << 40, (g, nq, k)).astype(np.uint64)
    sc = np.sort(rng.standard_normal((g, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    sc[1, :, 0] = sc[0, :, 0]                       # cross-shard score ties -> id ascending
    idx[2, 5, k - 3:] = np.uint64(2**64 - 1)        # a short shard: padding entries
    sc[2, 5, k - 3:] = -np.inf
    recs = []
    for r in range(g):
        ti = torch.from_numpy(idx[r].view(np.int64)).cuda()
        ts = torch.from_numpy(sc[r]).cuda()
        rec = m.cgvec.pack_topk(ti, ts)
        assert rec.shape == (nq, m.cgvec.packed_width(k)) and m.cgvec.packed_width(k) % 2 == 0
        recs.append(rec)
    oi, os_ = m.cgvec.merge_packed(torch.stack(recs).contiguous(), k)
    oi2, os2 = m.merge_topk(torch.from_numpy(idx.view(np.int64)).cuda(), torch.from_numpy(sc).cuda())
    for q in range(nq):
        ri, rs = oracle.merge_topk(idx[:, q, :], sc[:, q, :], k)
        assert np.array_equal(oi[q].cpu().numpy().view(np.uint64), ri) and np.array_equal(os_[q].cpu().numpy(), rs)
    assert torch.equal(oi, oi2) and torch.equal(os_, os2)
    # pinned HOST result arrays: the merge kernel writes them in place (what a rank of the sharded bench hands over)
    hi = torch.full((nq, k), -1, dtype=torch.int64).pin_memory()
    hs = torch.full((nq, k), float("nan"), dtype=torch.float32).pin_memory()
    ri2, rs2 = m.cgvec.merge_packed(torch.stack(recs).contiguous(), k, out=(hi, hs))
    torch.cuda.synchronize()
    assert ri2 is hi and rs2 is hs
    assert torch.equal(hi, oi.cpu()) and torch.equal(hs, os_.cpu())
    with pytest.raises(m.CgvError):
        m.cgvec.merge_packed(torch.stack(recs).contiguous(), k, out=(torch.empty((nq, k), dtype=torch.int64), hs))   # pageable


def test_randomised_shapes_differential(oracle):
    """Seeded sweep over shapes the fixed cases do not pin: n around tile / block boundaries, nq around
    query-tile boundaries, arbitrary D, k up to 256 (beyond the MFMA path's k' cap -> exact path),
    every dtype x metric the library accepts, duplicated and zero rows sprinkled in."""
    m = pkg()
    rng = np.random.default_rng(20260927)
    ns = [1, 2, 31, 32, 33, 255, 256, 257, 511, 512, 513, 1023, 1025, 4095, 4097, 9000, 20011]
    nqs = [1, 2, 7, 64, 255, 256, 257, 300, 513]
    combos = [("bf16", "cosine"), ("bf16", "dot"), ("fp16", "cosine"), ("fp16", "dot"), ("fp8", "cosine"),
              ("f32", "cosine"), ("f32", "dot"), ("f32s", "cosine"), ("f32s", "dot"), ("bf16", "cosine_seq"),
              ("f32s", "cosine_seq")]
    for it in range(33):
        dtype, metric = combos[it % len(combos)]
        n = int(rng.choice(ns))
        nq = int(rng.choice(nqs))
        d = int(rng.choice([1, 3, 8, 16, 31, 32, 33, 64, 65, 100, 128, 200, 384, 513]))
        k = int(rng.choice([1, 2, 5, 10, 16, 30, 57, 64, 100, 256]))
        if n * nq > 3_000_000:
            nq = max(1, 3_000_000 // n)
        rows = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        if n > 40:
            rows[n // 2] = rows[3]                    # a duplicate
            rows[n // 3] = 0.0                        # a zero row
        if nq > 2:
            q[1] = rows[min(3, n - 1)]
        ix = m.HipKnnIndex(d, metric=metric, dtype=dtype)
        try:
            cut = int(rng.integers(0, n + 1))
            ix.add(rows[:cut])
            ix.add(rows[cut:])
            idx, sc = ix.search(q, k)
            _check(oracle, rows, q, k, dtype, metric, idx, sc, f"it={it} n={n} nq={nq} d={d} k={k} {dtype} {metric}")
        finally:
            ix.close()


def test_many_queries_more_query_tiles_than_corpus_splits(oracle):
    """nq = 2100 -> 9 query tiles (a query-tile count C2-C5 do not use; nsplit_max = 28), k = 30
    (k' = 40), dot metric on un-normalised rows."""
    st = _run(oracle, n=12_000, d=96, nq=2100, k=30, dtype="bf16", metric="dot", seed=77, unit=False)
    assert st["last_path"] == 1 and st["fallback_queries"] <= 5


def _golden_stride(R):
    """csrc/plan.cpp golden_stride: P ~ 0.618 R, coprime to R (the visiting order of the corpus tiles is j * P mod R)."""
    from math import gcd
    if R <= 2:
        return 1
    P = max(1, int(R * 0.6180339887498949))
    while gcd(P, R) != 1:
        P += 1
    return P


@pytest.mark.parametrize("dtype,odt,metric", [("bf16", 1, "cosine"), ("fp8", 3, "cosine"), ("f32s", 0, "cosine"), ("fp16", 2, "dot")])
def test_emitting_sample_floor_crowded_cell_is_put_right_inside_the_final_kernel(oracle, dtype, odt, metric):
    """Round 6: the sample launch emits - per query and CELL of a sampled tile ((tile, 128-row half, rows with row % 8 < 4 or
    >= 4): 64 rows) its two best rows, and the best score it left out goes to the query's floor. Three near-copies of a query
    in ONE cell of a sampled tile leave the third out: the floor rises above the k-th exact score and the check fails on the
    floor alone. The final kernel then scores the offending cell again itself (sample_floor_repair, kernels_select.h): no
    fallback flag, no exact scan - and on the row-sharded path no provisional record and no second exchange (the 8-GPU bench met
    such a query in every fourth step). Two in a cell are both kept; three in a cell of a tile the sample does not visit are
    emitted by the ordinary threshold path. Every answer is the oracle's."""
    import torch
    m = pkg()
    rng = np.random.default_rng(77)
    n, d, nq, k = 80_000, 64, 1024, 10                       # 313 tiles, 4 query tiles -> the sample takes 64 tiles, the plan emits (>= 4 S)
    R = (n + 255) // 256
    P = _golden_stride(R)
    order = [(j * P) % R for j in range(R)]
    t_in, t_in2, t_out = order[1], order[40], order[100]      # two sampled tiles (positions < 64) and one the sample never sees
    t_last = order[63]                                        # the sample's last tile
    rows = _unit(rng, n, d)
    q = _unit(rng, nq, d)

    def plant(tile, qi, offsets, step=0.02):
        for j, off in enumerate(offsets):
            v = q[qi] + step * (j + 1) * _unit(rng, 1, d)[0]
            rows[tile * 256 + off] = v / np.linalg.norm(v)
    plant(t_in, 5, (0, 1, 2))             # three in one cell of a sampled tile (rows 0..2: first half, row % 8 < 4) -> floor violation
    plant(t_in2, 300, (130, 131))         # two in one cell: both kept
    plant(t_in2, 301, (0, 4, 128))        # three in three different cells of one sampled tile: all kept
    plant(t_out, 700, (8, 9, 10))         # three in one cell of a tile behind the sample: the threshold path emits them
    # five in one cell (second half, row % 8 >= 4) of the sample's LAST tile: three left-out rows join the keys
    plant(t_last, 900, (128 + 4, 128 + 5, 128 + 6, 128 + 7, 128 + 12), step=0.01)
    # two crowded cells of two sampled tiles for ONE query: both are scored again
    plant(order[7], 901, (32, 33, 34), step=0.01)
    plant(order[9], 901, (160 + 4, 160 + 5, 160 + 6), step=0.015)
    ix = m.HipKnnIndex(d, dtype=dtype, metric=metric)
    try:
        ix.add(rows)
        assert m.cgvec.lib().cgv_debug_last_top2_(ix._h) == 0
        f0, r0 = ix.stats()["fallback_queries"], ix.sample_repairs()
        gi, gs = ix.search(q, k)
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt, metric=OMETRIC[metric])
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        st = ix.stats()
        assert st["last_path"] == 1
        fl = (C.c_uint32 * nq)()
        L = m.cgvec.lib()
        L.cgv_debug_fbflags_.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
        assert L.cgv_debug_fbflags_(ix._h, 0, fl, nq) == 0
        # the planted queries are answered without a flag. (A random query may fail its check another way - or the same way,
        # 2e-5 per query on a corpus this small, and then it is put right too.)
        assert all(fl[i] == 0 for i in (5, 300, 301, 700, 900, 901)), [(i, hex(fl[i])) for i in range(nq) if fl[i]]
        assert st["fallback_queries"] - f0 <= 1, st
        assert 3 <= ix.sample_repairs() - r0 <= 5
        assert set(gi[5][:3].tolist()) == {t_in * 256, t_in * 256 + 1, t_in * 256 + 2}
        assert set(gi[700][:3].tolist()) == {t_out * 256 + 8, t_out * 256 + 9, t_out * 256 + 10}
        assert set(gi[900][:5].tolist()) == {t_last * 256 + o for o in (132, 133, 134, 135, 140)}
        assert set(gi[901][:6].tolist()) == {order[7] * 256 + o for o in (32, 33, 34)} | {order[9] * 256 + o for o in (164, 165, 166)}
        # the same batch again (floor words, scand and lists are reused), and without the planted triple: nothing to put right
        r1 = ix.sample_repairs()
        gi2, gs2 = ix.search(q, k)
        assert np.array_equal(gi2, ri) and np.array_equal(gs2, rs)
        assert ix.sample_repairs() - r1 == r1 - r0
        q2 = q.copy()
        q2[5] = _unit(rng, 1, d)[0]
        f1, r2 = ix.stats()["fallback_queries"], ix.sample_repairs()
        gi3, gs3 = ix.search(q2, k)
        r3 = oracle.batch_top_k(q2, rows, k, dtype=odt, metric=OMETRIC[metric])
        assert np.array_equal(gi3, r3[0]) and np.array_equal(gs3, r3[1])
        assert ix.stats()["fallback_queries"] - f1 <= 1 and ix.sample_repairs() - r2 == (r1 - r0) - 1
        # the row-sharded rank program's step (packed records, merge with the redo word): no provisional record, no second exchange
        if dtype == "bf16":
            sk = m.ShardedKnn(ix, rank=0, world=1)
            out = (torch.empty((nq, k), dtype=torch.int64).pin_memory(), torch.empty((nq, k), dtype=torch.float32).pin_memory())
            before = sk.redo_batches
            sk.step_packed(torch.from_numpy(q).pin_memory(), k, out=out, device=torch.device("cuda", 0))
            assert sk.redo_batches == before
            assert np.array_equal(out[0].numpy().view(np.uint64), ri) and np.array_equal(out[1].numpy(), rs)
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,odt", [("bf16", 1), ("fp16", 2)])
def test_emitting_sample_clustered_corpus_many_repairs(oracle, dtype, odt):
    """A store that inserts the chunks of one file next to each other: files of 8 adjacent near-duplicate rows, 1024 queries each
    aimed at a file. A file in a SAMPLED tile puts 4 of the query's best rows into each of two 64-row cells: both cells leave two
    out, the floor check fails, and the final kernel re-scores both cells itself - for about a fifth of the batch (64 of 313
    tiles are sampled). One query meets 75 exact copies of one vector in 25 sampled cells: all ties, the exact scan answers it.
    Every answer is the oracle's."""
    m = pkg()
    rng = np.random.default_rng(5)
    n, d, nq, k = 80_000, 64, 1024, 10
    files = n // 8
    base = _unit(rng, files, d)
    rows = np.repeat(base, 8, axis=0) + 0.03 * rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    pick = rng.choice(files, nq, replace=False)
    q = base[pick] + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)
    R, P = (n + 255) // 256, _golden_stride((n + 255) // 256)
    sampled = {(j * P) % R for j in range(64)}
    # query 0: 75 EXACT copies of one vector, three in each of 25 cells of sampled tiles: every score ties, nothing can be proven
    # from the candidates (and there would be more offending cells than the repair takes): the exact scan answers, ids ascending
    v = q[0] + 0.05 * _unit(rng, 1, d)[0]
    v /= np.linalg.norm(v)
    for (t, c) in [(t, c) for t in sorted(sampled)[:7] for c in range(4)][:25]:
        for j in range(3):
            rows[t * 256 + (c >> 1) * 128 + 4 * (c & 1) + 40 + j] = v      # rows of cell c (M-half, row % 8 < 4 or >= 4)
    in_sample = sum(1 for f in pick[1:] if (int(f) * 8) // 256 in sampled)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        f0, r0 = ix.stats()["fallback_queries"], ix.sample_repairs()
        gi, gs = ix.search(q, k)
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
        assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
        fixed, fb = ix.sample_repairs() - r0, ix.stats()["fallback_queries"] - f0
        assert in_sample > 100 and fixed >= 0.9 * in_sample, (in_sample, fixed, fb)
        assert 1 <= fb <= 0.1 * in_sample + 2, (in_sample, fixed, fb)      # query 0; near-ties inside a file may fail the check another way
        assert gi[0].tolist() == sorted(gi[0].tolist()) and len(set(gs[0].tolist())) == 1
    finally:
        ix.close()


@pytest.mark.parametrize("n,nq,k,d,dtype,metric", [
    (33_000, 1024, 10, 64, "bf16", "cosine"),      # R = 129 tiles = 2 S + 1: the smallest corpus whose sample emits
    (40_000, 700, 30, 96, "fp16", "dot"),          # 3 query tiles -> S = 85; k' = 40; un-normalised rows
    (70_000, 1024, 57, 64, "bf16", "cosine"),      # k' = 64: the largest the fused final selection takes
    (140_000, 300, 10, 64, "fp8", "cosine"),       # 2 query tiles -> S = 128, R = 547; k' = 32 (fp8)
    (66_000, 130, 5, 100, "f32s", "cosine"),       # one query tile -> S = 256, R = 258; ragged D; f32 rows + bf16 shadow
])
def test_emitting_sample_regime_sweep(oracle, n, nq, k, d, dtype, metric):
    """Corpora between 2 S and a few S tiles (S = sample tiles = CUs / query tiles): the sample launch emits and the launch behind
    it covers as few as S + 1 tiles - shapes the headline configs never produce (round 6 lowered the limit from 4 S to 2 S once a
    crowded sample cell no longer cost an exact scan). Planted near-duplicate runs make some queries fail the floor check."""
    m = pkg()
    rng = np.random.default_rng(n + nq)
    unit = metric == "cosine"
    rows = _unit(rng, n, d) if unit else (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    q = _unit(rng, nq, d) if unit else rng.standard_normal((nq, d)).astype(np.float32)
    for j in range(0, min(nq, 200), 2):            # runs of 6 adjacent near-copies of a query, anywhere in the corpus
        r0 = int(rng.integers(0, n - 8)) & ~7
        for t in range(6):
            v = q[j] + 0.02 * (t + 1) * _unit(rng, 1, d)[0] * np.linalg.norm(q[j])
            rows[r0 + t] = v / (np.linalg.norm(v) if unit else 1.0)
    ix = m.HipKnnIndex(d, metric=metric, dtype=dtype)
    try:
        ix.add(rows)
        emits = bool(_plan_emits(m, n, k, nq, dtype == "f32s"))
        assert emits
        gi, gs = ix.search(q, k)
        _check(oracle, rows, q, k, dtype, metric, gi, gs, f"n={n} nq={nq} k={k} d={d} {dtype} {metric}")
        st = ix.stats()
        assert st["last_path"] == 1 and st["fallback_queries"] <= max(8, nq // 10), st   # (near-ties of the planted runs, mostly fp8)
    finally:
        ix.close()


def _plan_emits(m, n, k, nq, shadow):
    L = m.cgvec.lib()
    out = (C.c_uint32 * 64)()
    L.cgv_debug_plan_.restype = C.c_uint32
    L.cgv_debug_plan_.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.c_uint32]
    n_cu = m.cgvec.device_properties(0)["multiProcessorCount"] if hasattr(m.cgvec, "device_properties") else 256
    L.cgv_debug_plan_(n, k, nq, n_cu, 1 if shadow else 0, out, 64)
    return out[1] & 0x10000


def test_emitting_sample_repairs_with_three_batches_in_flight(oracle):
    """The in-kernel repair reads the sample's per-cell floor values out of the search context's own scratch: three different
    clustered batches in flight on one index (three contexts, three streams), twice over - every batch's answer is the oracle's and
    the repairs of all of them are counted."""
    import torch
    m = pkg()
    rng = np.random.default_rng(11)
    n, d, nq, k = 80_000, 64, 1024, 10
    files = n // 8
    base = _unit(rng, files, d)
    rows = np.repeat(base, 8, axis=0) + 0.03 * rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    qs = [base[rng.choice(files, nq, replace=False)] + 0.01 * rng.standard_normal((nq, d)).astype(np.float32) for _ in range(3)]
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.add(rows)
        want = [oracle.batch_top_k(q, rows, k, dtype=1) for q in qs]
        qp = [torch.from_numpy(q).pin_memory() for q in qs]
        outs = [(torch.empty((nq, k), dtype=torch.int64).pin_memory(), torch.empty((nq, k), dtype=torch.float32).pin_memory()) for _ in range(3)]
        r0 = ix.sample_repairs()
        for rnd in range(2):
            pend = [ix.search_begin_pinned(qp[i], k, outs[i]) for i in range(3)]
            for p in reversed(pend):                       # (any end order)
                p.wait()
            for i in range(3):
                assert np.array_equal(outs[i][0].numpy().view(np.uint64), want[i][0]), (rnd, i)
                assert np.array_equal(outs[i][1].numpy(), want[i][1]), (rnd, i)
        assert ix.sample_repairs() - r0 > 6 * 100          # ~200 of 1024 queries per batch sit in sampled tiles
    finally:
        ix.close()


def test_phase_times_of_the_mfma_pipeline(oracle):
    """cgv_set_profiling(3) + cgv_get_phase_times: HIP events at the phase boundaries of the MFMA pipeline, on the stream the batch
    runs on - conversion | first threshold | emitting launches | final + publish. What bench.py's N > 1 line prints per rank."""
    m = pkg()
    rng = np.random.default_rng(3)
    n, d = 60_000, 128
    rows, q = _unit(rng, n, d), _unit(rng, 512, d)
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.add(rows)
        ix.search(q, 10)
        assert all(v == 0.0 for v in ix.phase_times_us().values())            # profiling off: zeros
        ix.set_profiling(3)
        for nq in (512, 8):                                                    # staged thresholds; one COARSE_TOP2 launch
            ix.search(q[:nq], 10)                                              # (first use of a path allocates its scratch: host time inside the span)
            gi, gs = ix.search(q[:nq], 10)
            ri, rs = oracle.batch_top_k(q[:nq], rows, 10, dtype=1)
            assert np.array_equal(gi, ri) and np.array_equal(gs, rs)
            ph, st = ix.phase_times_us(), ix.stats()
            assert ph["prep"] > 0 and ph["emitting"] > 0 and ph["final_publish"] > 0, ph
            assert (ph["first_threshold"] > 15.0) == (nq > 64), ph             # (a small batch has no threshold phase: two adjacent events)
            assert abs(sum(ph.values()) - 1e3 * st["last_total_ms"]) < 0.05 * 1e3 * st["last_total_ms"] + 5.0, (ph, st["last_total_ms"])
        ix.set_force_exact(True)
        ix.search(q[:4], 10)
        assert all(v == 0.0 for v in ix.phase_times_us().values())            # exact scan: no MFMA phases
    finally:
        ix.close()
