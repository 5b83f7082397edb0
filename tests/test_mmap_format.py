"""the reference's corpus file format (memory.rs:242-374): 16-byte header
{u64 vector_count, u64 dimension} + row-major f32. The writer needs no GPU."""
import os
import struct

import numpy as np

from _util import pkg


def test_write_mmap_layout(tmp_path):
    m = pkg()
    rows = np.arange(5 * 7, dtype=np.float32).reshape(5, 7) * np.float32(0.25)
    p = tmp_path / "corpus.bin"
    m.write_mmap(p, rows)
    raw = p.read_bytes()
    assert len(raw) == 16 + 5 * 7 * 4                      # memory.rs:258-261
    assert struct.unpack("<QQ", raw[:16]) == (5, 7)        # memory.rs:285-289
    assert np.array_equal(np.frombuffer(raw[16:], dtype="<f4").reshape(5, 7), rows)


def test_write_mmap_empty_writes_no_file(tmp_path):
    m = pkg()
    p = tmp_path / "none.bin"
    m.write_mmap(p, np.zeros((0, 8), np.float32))          # memory.rs:243-245
    assert not os.path.exists(p)
