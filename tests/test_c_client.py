"""A plain-C program against include/cgvec.h (tests/c_client/abi_client.c): compiles with gcc, links the
library without Python or torch in the process. On a GPU box its results must equal the oracle's;
without a GPU it must fail loudly (exit 3, "no CPU fallback")."""
import os
import struct
import subprocess

import numpy as np
import pytest

from _util import ROOT, pkg

SRC = os.path.join(ROOT, "tests", "c_client", "abi_client.c")


def _build(tmp_path, src=SRC, name="abi_client", extra=()):
    m = pkg()
    m.build_library()
    exe = str(tmp_path / name)
    libdir = os.path.dirname(m.cgvec.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", *extra, "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lcgvec_hip", "-Wl,-rpath," + libdir])
    return exe


CALLERS = os.path.join(ROOT, "tests", "c_client", "callers.c")


SHIM = os.path.join(ROOT, "tests", "c_client", "shim_replay.c")


def _write_input(path, rows, q, k):
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", rows.shape[0], rows.shape[1], q.shape[0], k))
        f.write(rows.tobytes())
        f.write(q.tobytes())


def test_c_client_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    m = pkg()
    if m.device_count() > 0:
        pytest.skip("GPU present")
    rng = np.random.default_rng(0)
    _write_input(tmp_path / "in.bin", rng.standard_normal((64, 16)).astype(np.float32),
                 rng.standard_normal((2, 16)).astype(np.float32), 5)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr


def test_pthread_client_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path, CALLERS, "callers", extra=("-pthread", "-DCALLERS_MAIN"))
    if pkg().device_count() > 0:
        pytest.skip("GPU present")
    rng = np.random.default_rng(0)
    _write_input(tmp_path / "in.bin", rng.standard_normal((64, 16)).astype(np.float32),
                 rng.standard_normal((4, 16)).astype(np.float32), 5)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "1", "4"], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr


def test_shim_replay_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path, SHIM, "shim_replay")
    if pkg().device_count() > 0:
        pytest.skip("GPU present")
    rng = np.random.default_rng(0)
    _write_input(tmp_path / "in.bin", rng.standard_normal((64, 16)).astype(np.float32),
                 rng.standard_normal((2, 16)).astype(np.float32), 5)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.txt")], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,odt", [(4, 0), (1, 1)])
def test_shim_call_sequence_replay(tmp_path, oracle, dtype, odt):
    """The Rust shim of INTEGRATION.md §3 has no logic: tests/c_client/shim_replay.c issues its exact call sequence
    (upsert in two batches, kNN by column, RE-UPSERT of a known id, kNN, get_embedding, search_similar). Expected:
    the seam's contract (surreal_store.rs:11-22, 61-85) evaluated by the oracle - ids by row, distance = 1 - cosine
    ascending, the re-upserted node found once by its NEW embedding (UPSERT), its old embedding gone."""
    exe = _build(tmp_path, SHIM, "shim_replay")
    rng = np.random.default_rng(21)
    n, d, k = 6000, 384, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((2, d)).astype(np.float32)
    q[1] = rows[7] * np.float32(-1.0) + rng.standard_normal(d).astype(np.float32) * np.float32(0.05)   # far from row 7's old value
    _write_input(tmp_path / "in.bin", rows, q, k)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.txt"), str(dtype)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = {0: [], 1: [], 2: []}
    for line in open(tmp_path / "out.txt"):
        ph, rank, row, bits = line.split()
        got[int(ph)].append((int(row), np.array([int(bits, 16)], dtype=np.uint32).view(np.float32)[0]))
    # phase 0: the first query against the original rows
    ri, rs = oracle.batch_top_k(q[:1], rows, k, dtype=odt)
    assert [r for r, _ in got[0]] == ri[0].tolist()
    assert np.array_equal(np.array([x for _, x in got[0]], np.float32), np.float32(1.0) - rs[0])
    # phases 1, 2: row 7 now holds q[1]
    rows2 = rows.copy()
    rows2[7] = q[1]
    ri, rs = oracle.batch_top_k(q[1:], rows2, k, dtype=odt)
    assert [r for r, _ in got[1]] == ri[0].tolist() and ri[0][0] == 7
    assert np.array_equal(np.array([x for _, x in got[1]], np.float32), np.float32(1.0) - rs[0])
    assert [r for r, _ in got[2]] == ri[0].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,odt", [(1, 1), (0, 0), (4, 0)])
def test_c_client_results_equal_oracle(tmp_path, oracle, dtype, odt):
    exe = _build(tmp_path)
    rng = np.random.default_rng(3)
    n, d, nq, k = 20_000, 256, 33, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    _write_input(tmp_path / "in.bin", rows, q, k)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(dtype)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert f"rows={n}" in p.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    idx = np.frombuffer(raw[: nq * k * 8], dtype=np.uint64).reshape(nq, k)
    sc = np.frombuffer(raw[nq * k * 8: nq * k * 12], dtype=np.float32).reshape(nq, k)
    back = np.frombuffer(raw[nq * k * 12:], dtype=np.float32)
    ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
    assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    assert np.array_equal(back, oracle.round_trip(rows[1], odt))


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [2, 3])
def test_c_client_sharded_handle_equals_oracle(tmp_path, oracle, n_shards):
    """ONE cgv_sharded handle over several shards (devices i % device_count: a 1-GPU box lists device 0 more than
    once, a multi-GPU box uses distinct devices and the RCCL all-gather): ids are global insertion indices and
    the results equal the single-index / oracle results."""
    exe = _build(tmp_path)
    rng = np.random.default_rng(11)
    n, d, nq, k = 30_000, 128, 65, 10      # 30k rows = 8 chunks of 4096: uneven over 3 shards
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    _write_input(tmp_path / "in.bin", rows, q, k)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "1", str(n_shards)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert f"rows={n} shards={n_shards}" in p.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    idx = np.frombuffer(raw[: nq * k * 8], dtype=np.uint64).reshape(nq, k)
    sc = np.frombuffer(raw[nq * k * 8: nq * k * 12], dtype=np.float32).reshape(nq, k)
    back = np.frombuffer(raw[nq * k * 12:], dtype=np.float32)
    ri, rs = oracle.batch_top_k(q, rows, k, dtype=1)
    assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    assert np.array_equal(back, oracle.round_trip(rows[1], 1))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,odt,threads", [(1, 1, 64), (0, 0, 16)])
def test_pthread_client_concurrent_single_query_callers(tmp_path, oracle, dtype, odt, threads):
    """The pthread variant of the plain-C client (tests/c_client/callers.c): `threads` native threads, each in a serial loop of
    SINGLE-query cgv_search_f32 calls on one index - the reference's call shape (traits.rs:14; search.rs:358-361 issues B of them
    concurrently). The library merges concurrent callers into shared device batches (csrc/coalesce.h); every caller's ids and
    scores must be the oracle's."""
    exe = _build(tmp_path, CALLERS, "callers", extra=("-pthread", "-DCALLERS_MAIN"))
    rng = np.random.default_rng(13)
    n, d, k = 60_000, 256, 10
    nq = threads * 8
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    _write_input(tmp_path / "in.bin", rows, q, k)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(dtype), str(threads)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    idx = np.frombuffer(raw[: nq * k * 8], dtype=np.uint64).reshape(nq, k)
    sc = np.frombuffer(raw[nq * k * 8: nq * k * 12], dtype=np.float32).reshape(nq, k)
    ri, rs = oracle.batch_top_k(q, rows, k, dtype=odt)
    assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
    # "coalesce: batches B requests R ..." on stderr: the callers really shared batches
    words = p.stderr.split()
    assert int(words[words.index("batches") + 1]) > 0, p.stderr
