"""the reference's ScalarQuantizer / ProductQuantizer (persistent.rs:116-477) on the GPU
against the oracle's literal single-thread restatement: parameters, codes and decoded values bit-equal."""
import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbits,uniform", [(8, False), (8, True), (4, False), (16, False), (12, True)])
def test_scalar_quantizer(oracle, nbits, uniform):
    m = pkg()
    rng = np.random.default_rng(nbits)
    x = (rng.standard_normal((500, 37)) * rng.uniform(0.1, 3.0, (1, 37))).astype(np.float32)
    x[:, 5] = 0.25                                   # zero range: scale stays 1, bias 0 (non-uniform, :397-400)
    sq = m.ScalarQuantizer(37, nbits, uniform)
    try:
        with pytest.raises(m.CgvError, match="Quantizer not trained"):
            sq.encode(x[:2])
        with pytest.raises(m.CgvError, match="Cannot train on empty vector set"):
            sq.train(np.zeros((0, 37), np.float32))
        sq.train(x)
        sc, bi = sq.params()
        rsc, rbi = oracle.sq_train(x, nbits, uniform)
        assert np.array_equal(sc, rsc) and np.array_equal(bi, rbi)
        y = np.vstack([x[:50], x[:3] * 10, -x[:3] * 10])   # out-of-range values clamp
        y[0, 0] = np.nan                                   # NaN.max(0.0) == 0.0
        codes = sq.encode(y)
        assert np.array_equal(codes, oracle.sq_encode(y, nbits, rsc, rbi))
        assert np.array_equal(sq.decode(codes), oracle.sq_decode(codes, 37, nbits, rsc, rbi))
    finally:
        sq.close()


@pytest.mark.parametrize("n,dim,mm,nbits", [(600, 32, 4, 4), (300, 24, 8, 3), (40, 16, 2, 6), (500, 64, 8, 8)])
def test_product_quantizer(oracle, n, dim, mm, nbits):
    """k-means with the reference's init (first ksub sub-vectors, index i % n when ksub > n), 50 iterations,
    first-minimum assignment and sequential centroid sums."""
    m = pkg()
    rng = np.random.default_rng(n + dim)
    x = (rng.standard_normal((n, dim)) + rng.integers(0, 3, (n, 1)) * 2.0).astype(np.float32)
    x[5] = x[2]                                            # duplicates -> distance ties -> lowest centroid wins
    pq = m.ProductQuantizer(dim, mm, nbits)
    try:
        with pytest.raises(m.CgvError, match="Quantizer not trained"):
            pq.encode(x[:2])
        pq.train(x)
        cent = pq.centroids()
        ref = oracle.pq_train(x, mm, nbits)
        assert np.array_equal(cent, ref)
        y = rng.standard_normal((100, dim)).astype(np.float32)
        codes = pq.encode(y)
        assert np.array_equal(codes, oracle.pq_encode(y, ref))
        dec = pq.decode(codes)
        exp = np.concatenate([ref[s][codes[:, s]] for s in range(mm)], axis=1)
        assert np.array_equal(dec, exp)
        if nbits < 8:
            bad = codes.copy()
            bad[0, 0] = 255
            with pytest.raises(m.CgvError, match="Invalid centroid index"):
                pq.decode(bad)
    finally:
        pq.close()


def test_pq_all_rows_in_centroid_zero_stops_early(oracle):
    """Identical training vectors: every assignment is 0, `changed` stays false, one iteration (:311-313)."""
    m = pkg()
    x = np.tile(np.arange(8, dtype=np.float32), (20, 1))
    pq = m.ProductQuantizer(8, 2, 2)
    try:
        pq.train(x)
        assert np.array_equal(pq.centroids(), oracle.pq_train(x, 2, 2))
    finally:
        pq.close()


def test_pq_create_validation():
    m = pkg()
    with pytest.raises(m.CgvError, match="Dimension must be divisible by number of subquantizers"):
        m.ProductQuantizer(30, 4, 8)
    with pytest.raises(m.CgvError):
        m.ProductQuantizer(32, 4, 9)
