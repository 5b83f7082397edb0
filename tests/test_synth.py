"""SURVEY.md section 8(d) synthetic inputs: the counter-based generator keyed (seed, row, col) -> N(0, 1) f32 -> L2-normalised
in f32 (oracle side: cgo_synth_rows; device side: csrc/synth.hip, compared bit for bit in test_gpu_synth_matches_the_oracle).
The reference has no generator of its own; the Philox4x32-10 core is pinned by the published Random123 known-answer vectors."""
import zlib

import numpy as np
import pytest

SEED_CORPUS, SEED_QUERY = 0xC0DE6001, 0xC0DE6002


def test_philox4x32_10_known_answers(oracle):
    """Random123 (Salmon, Moraes, Dror, Shaw, SC'11) kat_vectors, philox4x32 with 10 rounds."""
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kats:
        assert oracle.philox4x32_10(ctr, key) == want


def test_synth_rows_contract_in_python(oracle):
    """One row restated in numpy float32 scalars (one rounding per operation) from the contract in csrc/synth.hip's header."""
    f = np.float32

    def ln(u):
        b = int(np.frombuffer(f(u).tobytes(), np.uint32)[0])
        e = (b >> 23) - 127
        m = np.frombuffer(np.uint32((b & 0x007FFFFF) | 0x3F800000).tobytes(), np.float32)[0]
        if m > f(1.41421356):
            m, e = f(m * f(0.5)), e + 1
        t = f(f(m - f(1)) / f(m + f(1)))
        t2 = f(t * t)
        p = f(0.0909090936)
        for c in (0.111111112, 0.142857149, 0.2, 0.333333343, 1.0):
            p = f(f(p * t2) + f(c))
        return f(f(f(e) * f(0.693147182)) + f(f(f(2) * t) * p))

    def pair(xa, xb):
        u = f(f(f(xa >> 9) + f(0.5)) * f(2.0 ** -23))
        v = f(f(xb >> 8) * f(2.0 ** -24))
        rad = f(np.sqrt(f(f(-2) * ln(u))))
        a4 = f(v * f(4))
        q = int(a4)
        a = f(f(a4 - f(q)) * f(1.57079637))
        a2 = f(a * a)
        ps = f(-2.50521084e-08)
        for c in (2.75573188e-06, -1.98412701e-04, 8.33333377e-03, -0.166666672, 1.0):
            ps = f(f(ps * a2) + f(c))
        sn = f(a * ps)
        pc = f(2.08767570e-09)
        for c in (-2.75573188e-07, 2.48015876e-05, -1.38888892e-03, 4.16666679e-02, -0.5, 1.0):
            pc = f(f(pc * a2) + f(c))
        c, s = [(pc, sn), (-sn, pc), (-pc, -sn), (sn, -pc)][q & 3]
        return f(rad * c), f(rad * s)

    for seed, r, dim in ((SEED_CORPUS, 5, 10), (SEED_QUERY, (1 << 32) + 3, 300)):
        z = []
        for b in range((dim + 3) // 4):
            x = oracle.philox4x32_10([b, r & 0xFFFFFFFF, r >> 32, 0], [seed & 0xFFFFFFFF, seed >> 32])
            z += list(pair(x[0], x[1])) + list(pair(x[2], x[3]))
        z = np.array(z[:dim], dtype=f)
        assert np.array_equal(oracle.synth_rows(seed, r, 1, dim, normalise=False)[0], z)
        part = [f(0)] * 64
        for c in range(dim):
            part[(c // 4) % 64] = f(part[(c // 4) % 64] + f(z[c] * z[c]))
        off = 32
        while off:
            part = [f(part[l] + part[l ^ off]) for l in range(64)]
            off //= 2
        assert np.array_equal(oracle.synth_rows(seed, r, 1, dim)[0], np.array([f(x / f(np.sqrt(part[0]))) for x in z], dtype=f))


def test_synth_rows_distribution_and_chunking(oracle):
    x = oracle.synth_rows(SEED_CORPUS, 0, 4000, 768, normalise=False)
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.std()) - 1.0) < 2e-3
    assert abs(float((x.astype(np.float64) ** 4).mean()) - 3.0) < 0.03          # a normal's fourth moment
    assert 4.5 < float(np.abs(x).max()) < 5.77                                    # |z| <= sqrt(2 * 24 ln 2) by construction
    col = x[:, :16].astype(np.float64)
    cc = np.corrcoef(col.T) - np.eye(16)
    assert float(np.abs(cc).max()) < 0.08                                         # neighbouring columns (one Philox block) uncorrelated
    y = oracle.synth_rows(SEED_CORPUS, 0, 4000, 768)
    n = np.linalg.norm(y.astype(np.float64), axis=1)
    assert float(np.abs(n - 1.0).max()) < 3e-7
    # any chunking gives the same rows; a different seed / dim does not
    assert np.array_equal(oracle.synth_rows(SEED_CORPUS, 1234, 77, 768), y[1234:1311])
    assert not np.array_equal(oracle.synth_rows(SEED_QUERY, 0, 8, 768), y[:8])
    assert np.array_equal(oracle.synth_rows(SEED_CORPUS, 0, 8, 764, normalise=False), x[:8, :764])   # cols are keyed, not streamed
    # the committed checksum of the first rows of the two streams SURVEY.md names: the data of every bench line
    assert zlib.crc32(oracle.synth_rows(SEED_CORPUS, 0, 64, 768).tobytes()) == GOLDEN_CRC["corpus"]
    assert zlib.crc32(oracle.synth_rows(SEED_QUERY, 0, 64, 768).tobytes()) == GOLDEN_CRC["query"]
    assert oracle.synth_rows(SEED_CORPUS, 0, 0, 768).shape == (0, 768)


GOLDEN_CRC = {"corpus": 146529031, "query": 70095500}


@pytest.mark.gpu
def test_gpu_synth_matches_the_oracle(oracle):
    """The device generator (cgv_synth_rows_f32_dev) against the oracle's, bit for bit: ragged dims, rows beyond 2^32, N(0, 1)
    and unit-norm forms, a C2-sized chunk; then an index built from device-generated rows answers like the oracle on
    oracle-generated rows (the CPU never sees the device's data)."""
    import importlib
    m = importlib.import_module("codegraph-rust_amd")
    for seed, row0, n, d in ((SEED_CORPUS, 0, 300, 768), (SEED_QUERY, 0, 1024, 768), (7, (1 << 32) - 5, 40, 384), (9, 11, 130, 1),
                             (9, 11, 5, 3), (3, 0, 257, 1536), (3, 999_000, 513, 100), (SEED_CORPUS, 875_000, 20_000, 768)):
        for normalise in (True, False):
            got = m.cgvec.synth_rows_dev(seed, row0, n, d, normalise=normalise).cpu().numpy()
            assert np.array_equal(got, oracle.synth_rows(seed, row0, n, d, normalise=normalise)), (seed, row0, n, d, normalise)
    n, d, nq, k = 30_000, 256, 48, 10
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        for lo in range(0, n, 8192):
            ix.add(m.cgvec.synth_rows_dev(SEED_CORPUS, lo, min(8192, n - lo), d))
        q = oracle.synth_rows(SEED_QUERY, 0, nq, d)
        idx, sc = ix.search(q, k)
        ri, rs = oracle.batch_top_k(q, oracle.synth_rows(SEED_CORPUS, 0, n, d), k, dtype=1)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    finally:
        ix.close()
