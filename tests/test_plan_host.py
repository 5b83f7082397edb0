"""Host logic of the search pipeline that needs no GPU: the launch planner and the on-demand RCCL
loader's failure path (a missing librccl must degrade to the copy exchange, not crash)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from _util import pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(L, n, k, nq, n_cu=256, shadow=0):
    out = (C.c_uint32 * 64)()
    L.cgv_debug_plan_.restype = C.c_uint32
    L.cgv_debug_plan_.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.c_uint32]
    w = L.cgv_debug_plan_(n, k, nq, n_cu, shadow, out, 64)
    assert w >= 2 and (out[1] & 0xFFFF) == w - 2
    return out[0], list(out[2:w]), bool(out[1] & 0x10000)


@pytest.mark.parametrize("n,k,nq,shadow", [
    (1_000_000, 10, 1024, 0), (125_000, 10, 1024, 0), (250_000, 10, 1024, 0), (500_000, 10, 1024, 0),
    (1_000_000, 10, 256, 0), (1_250_000, 10, 4096, 0), (62_500_000, 10, 8192, 0), (1_000_000, 10, 1024, 1),
    (40_000, 10, 1024, 0), (100_000, 10, 300, 0), (200_000, 10, 256, 0), (4097, 10, 1, 0), (10_000, 200, 16, 0), (999_999, 228, 1024, 0), (70_000, 1, 1, 0), (300_000, 60, 7, 1),
])
def test_plan_covers_every_tile_once(n, k, nq, shadow):
    L = pkg().cgvec.lib()
    sample, counts, emits = _plan(L, n, k, nq, shadow=shadow)
    ntiles = (n + 255) // 256
    assert 0 < sample <= 256 and sample <= ntiles         # 16 / 8 / 4 group maxima per tile -> <= 1024 values per query
    # round 6: a sample that emits its own candidates is not scored again - the launches cover the tiles behind it
    assert all(c > 0 for c in counts) and sum(counts) + (sample if emits else 0) == ntiles
    kprime = min((4 * k + 16 + 7) // 8 * 8, 256) if shadow else (k + max(6, k // 8) + 7) // 8 * 8
    assert emits == (nq > 64 and kprime <= 64 and ntiles >= 2 * sample and counts[0] >= min(sample, max(1, 256 // ((nq + 255) // 256))))
    if emits:
        assert min(counts[0], max(1, 256 // ((nq + 255) // 256))) >= sample    # the first launch continues every sampled tile's lists
    assert len(counts) <= 8
    # thresholds tighten launch by launch: the rows behind a launch's threshold never shrink relative to its size
    seen = sample
    for c in counts[:-1]:
        assert c <= 64 * seen
        seen += c


def test_plan_shapes_of_the_baseline_configs():
    L = pkg().cgvec.lib()
    assert _plan(L, 1_000_000, 10, 1024) == (64, [448, 3395], True)   # C2: emitting sample, 112 k rows, the dominant launch
    assert _plan(L, 125_000, 10, 1024) == (64, [425], True)           # C2's 8-GPU shard: ONE launch behind the sample
    assert _plan(L, 4096, 10, 64)[0] == 0                              # <= 16 tiles: the dense boot stage covers the corpus
    assert _plan(L, 1_000_000, 10, 256) == (256, [3651], True)        # C4, one query tile: the sample uses every CU
    assert _plan(L, 1_000_000, 10, 64) == (256, [3907], False)        # <= 64 queries take COARSE_TOP2 anyway: the old form
    assert _plan(L, 1_000_000, 200, 1024)[2] is False                  # k' > 64: the final selection is not fused, no floor there


def test_rccl_loader_failure_is_reported_not_fatal():
    code = (
        "import ctypes as C, importlib, sys; sys.path.insert(0, %r)\n"
        "m = importlib.import_module('codegraph-rust_amd'); L = m.cgvec.lib()\n"
        "assert L.cgv_debug_rccl_lib_(b'/nonexistent/librccl-missing.so') == 1\n"
        "buf = C.create_string_buffer(512)\n"
        "ok = L.cgv_debug_rccl_probe_(buf, 512); print(ok, buf.value.decode())\n"
        "ok2 = L.cgv_debug_rccl_probe_(buf, 512); print(ok2, buf.value.decode())\n" % ROOT)
    env = dict(os.environ)   # (the library reads nothing from the environment: the name is set through the internal setter)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("0 ") and "librccl-missing" in lines[0]   # dlerror text, read once
    assert lines[1].startswith("0 ") and "librccl-missing" in lines[1]   # and still there on the second call
