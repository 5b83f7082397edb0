"""the reference's int8 'optimized' scan on the GPU, through the C ABI
(include/cgvec_i8.h), against the oracle's literal restatement of optimization.rs:63-150.
Integer arithmetic + one f32 formula: the returned indices must be identical, ties included."""
import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu


def _rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


@pytest.mark.parametrize("n,d,limit", [(1000, 128, 10), (20_000, 768, 10), (9000, 100, 37), (50, 33, 80), (3, 7, 1)])
def test_search_optimized_matches_reference_policy(oracle, n, d, limit):
    m = pkg()
    rng = np.random.default_rng(n + d)
    rows = _rows(rng, n, d) * np.float32(3.0)        # some components clamp at +-1
    data = oracle.quantize_u8(rows)
    ix = m.Int8ScanIndex(d)
    try:
        ix.add_u8(data[: n // 2])
        ix.add(rows[n // 2:])                        # device-side quantize_batch
        assert len(ix) == n
        assert np.array_equal(ix.get_row(n - 1), data[n - 1])
        for qi in range(4):
            q = _rows(rng, 1, d)[0] * np.float32(2.0)
            got = ix.search_optimized(q, limit)
            ref = oracle.search_optimized_u8(q, data, limit)
            assert np.array_equal(got, ref), (qi, got[:8], ref[:8])
    finally:
        ix.close()


def test_reference_optimization_fixture_gate(oracle):
    """The reference's own fixture (tests/model_optimization_tests.rs:36-58, :347-427; inputs regenerated bit for
    bit, tests/golden/make_optimization_fixture.py): the device int8 scan and the device baseline return the
    recorded oracle lists and pass the test's gate (positional top-10 agreement >= 0.8)."""
    import os
    m = pkg()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optimization_11223.npz"))
    v, q = g["vectors"], g["vectors"][0]
    ix8 = m.Int8ScanIndex(128)
    ix = m.HipKnnIndex(128, metric="cosine", dtype="f32", device=0)
    try:
        ix8.add(v)                                   # device-side quantize_batch (optimization.rs:226-283)
        opt = ix8.search_optimized(q, 10)            # optimization.rs:63-150
        ix.add(v)
        base, dist = ix.search_baseline(q, 10)       # optimization.rs:376-418
        assert len(opt) == len(base) == 10
        assert sum(int(a == b) for a, b in zip(opt, base)) / 10.0 >= 0.8
        assert np.array_equal(opt, g["int8_idx"]) and np.array_equal(base, g["baseline_idx"])
        assert np.array_equal(dist, g["baseline_dist"])
    finally:
        ix8.close()
        ix.close()


def test_quantize_batch_matches(oracle):
    m = pkg()
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((257, 131)) * 0.7).astype(np.float32)
    x[0, :6] = [0.5 / 127, 1.5 / 127, -0.5 / 127, -2.5 / 127, 1.0, -1.0]   # round half away from zero
    x[1, :3] = [np.nan, np.inf, -np.inf]                                   # NaN -> 0, inf clamps
    assert np.array_equal(m.quantize_u8(x), oracle.quantize_u8(x))


def test_quantize_batch_4bit_arm(oracle):
    m = pkg()
    rng = np.random.default_rng(6)
    for d in (8, 33):                                    # odd dimension: the last code pairs with 0
        x = (rng.standard_normal((129, d)) * 0.8).astype(np.float32)
        x[0, :4] = [-1.0, 1.0, 0.0, 1.0 / 15.0]
        x[1, 0] = np.nan
        assert np.array_equal(m.quantize_u4(x), oracle.quantize_u4(x))


def test_ties_zero_rows_and_degenerate_queries(oracle):
    """The buffer policy is order dependent: among equal minima the most recently inserted entry is
    evicted first (optimization.rs:135-145). Duplicated rows + zero rows (skipped, :131-133)."""
    m = pkg()
    rng = np.random.default_rng(8)
    d, n = 64, 30_000
    rows = _rows(rng, n, d)
    rows[100:400] = rows[7]            # 300 copies of one row, before the 8192-row head boundary
    rows[9000:9300] = rows[7]          # and 300 more behind it
    rows[20:30] = 0.0                  # quantise to all-128 -> |v|^2 == 0 -> skipped
    rows[25_000:25_010] = 1e-4         # below half a quantisation step: also all zero
    data = oracle.quantize_u8(rows)
    ix = m.Int8ScanIndex(d)
    try:
        ix.add_u8(data)
        for q, limit in ((rows[7], 10), (rows[7], 450), (rows[7], 700), (rows[1234], 5), (-rows[7], 20)):
            got = ix.search_optimized(q, limit)
            ref = oracle.search_optimized_u8(q, data, limit)
            assert np.array_equal(got, ref), (limit, got[:12], ref[:12])
        sc = ix.scores(rows[7])
        assert np.isnan(sc[20:30]).all() and np.isnan(sc[25_000:25_010]).all() and not np.isnan(sc[:20]).any()
        assert ix.search_optimized(np.zeros(d, np.float32), 10).size == 0          # |q| == 0 (:113-115)
        assert ix.search_optimized(np.full(d, 1e-4, np.float32), 10).size == 0
        assert np.array_equal(ix.search_optimized(rows[5], 0), oracle.search_optimized_u8(rows[5], data, 0))  # limit.max(1)
        short = rows[5][: d // 2]                                                   # .take(dimension) + zero pad (:86-96)
        padded = np.concatenate([short, np.zeros(d - short.size, np.float32)])
        assert np.array_equal(ix.search_optimized(short, 10), oracle.search_optimized_u8(padded, data, 10))
    finally:
        ix.close()


def test_empty_index_and_errors():
    m = pkg()
    ix = m.Int8ScanIndex(16)
    try:
        assert ix.search_optimized(np.ones(16, np.float32), 5).size == 0
        with pytest.raises(m.CgvError):
            ix.add(np.ones((2, 15), np.float32))
        with pytest.raises(m.CgvError):
            ix.get_row(0)
    finally:
        ix.close()
