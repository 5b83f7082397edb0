"""SURVEY.md §8(f)3 on the GPU: load a corpus file in the reference's mmap format into an index
(page cache -> pinned staging -> device convert) and search it; f64 (SurrealDB column) ingest."""
import struct

import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


@pytest.mark.parametrize("dtype,odt", [("bf16", 1), ("f32", 0), ("fp8", 3)])
def test_load_mmap_then_search_equals_add(oracle, tmp_path, dtype, odt):
    m = pkg()
    rng = np.random.default_rng(17)
    n, d = 50_000, 384                                   # 77 MB: two 64-MiB staging chunks
    rows, q = _unit(rng, n, d), _unit(rng, 9, d)
    p = tmp_path / "corpus.bin"
    m.write_mmap(p, rows)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows[:1000])                              # the loader appends
        assert ix.load_mmap(p) == n
        assert len(ix) == n + 1000
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, np.vstack([rows[:1000], rows]), 10, dtype=odt)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        out = tmp_path / "saved.bin"
        ix.save_mmap(out)                                # stored values, upcast, same format
        raw = out.read_bytes()
        assert struct.unpack("<QQ", raw[:16]) == (n + 1000, d)
        back = np.frombuffer(raw[16:], dtype="<f4").reshape(n + 1000, d)
        assert np.array_equal(back[1000:1003], oracle.round_trip(rows[:3], odt))
    finally:
        ix.close()


def test_load_mmap_validation_messages(tmp_path):
    """Same checks and messages as load_from_mmap (memory.rs:310-348)."""
    m = pkg()
    ix = m.HipKnnIndex(8, dtype="f32")
    try:
        with pytest.raises(m.CgvError, match="Failed to open mmap file"):
            ix.load_mmap(tmp_path / "missing.bin")
        small = tmp_path / "small.bin"
        small.write_bytes(b"\x00" * 10)
        with pytest.raises(m.CgvError, match="Invalid mmap file: too small"):
            ix.load_mmap(small)
        wrong = tmp_path / "wrong.bin"
        m.write_mmap(wrong, np.ones((3, 16), np.float32))
        with pytest.raises(m.CgvError, match="Dimension mismatch: expected 8, found 16"):
            ix.load_mmap(wrong)
        trunc = tmp_path / "trunc.bin"
        trunc.write_bytes(struct.pack("<QQ", 4, 8) + b"\x00" * (3 * 8 * 4))
        with pytest.raises(m.CgvError, match=r"Invalid mmap file size: expected 144, got 112"):
            ix.load_mmap(trunc)
        empty = tmp_path / "empty.bin"
        empty.write_bytes(struct.pack("<QQ", 0, 8))
        assert ix.load_mmap(empty) == 0 and len(ix) == 0
    finally:
        ix.close()


def test_add_f64_column(oracle):
    m = pkg()
    rng = np.random.default_rng(23)
    rows, q = _unit(rng, 3000, 96), _unit(rng, 4, 96)
    ix = m.HipKnnIndex(96, dtype="bf16")
    try:
        ix.add_f64(rows.astype(np.float64))              # f32 -> f64 -> f32 is the identity
        idx, sc = ix.search(q, 10)
        ri, rs = oracle.batch_top_k(q, rows, 10, dtype=1)
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        wide = rng.standard_normal((10, 96))             # genuine f64 values: `as f32` = RNE
        ix2 = m.HipKnnIndex(96, dtype="f32")
        ix2.add_f64(wide)
        assert np.array_equal(ix2.get_row(3), wide[3].astype(np.float32))
        ix2.close()
    finally:
        ix.close()
