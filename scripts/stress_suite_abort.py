"""Stress for the intermittent abort of the full GPU suite (HISTORY.md §9.5): the WHOLE pinned / registered host-buffer test
case, in this process, followed by what the suite ran next when it died (a fresh index, cgv_add_f32 of 3*4096+77 x 64
pageable rows, a search), in a loop. tests/c_client/abort_bt.c is hooked in, so an abort names the native stack that raised it.

  python scripts/stress_suite_abort.py [seconds] [form]
      form = heap   : the registered buffer is a piece of the malloc heap, released after the case (the round-3 form
                      under which the two aborts were seen)
             mmap   : an anonymous mapping kept for the process' lifetime (the form the test uses now)
             both   : alternate
Run it plain, under MALLOC_CHECK_=3 MALLOC_PERTURB_=165, and under AMD_LOG_LEVEL=1|3 (scripts/gpu_abort_hunt.sh)."""
import gc
import importlib
import mmap
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler  # noqa: E402

faulthandler.enable()
from _util import install_abort_bt  # noqa: E402

print("abort_bt installed:", install_abort_bt(), flush=True)
import torch  # noqa: E402

m = importlib.import_module("codegraph-rust_amd")
from oracle import oracle as o  # noqa: E402

o.build()
C = 4096
KEEP = []


def pinned_case(form, cache):
    rng = np.random.default_rng(91)
    n, d, nq, k = 30_000, 256, 300, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5000:5060] = rows[17] * (1 + 1e-4 * rng.standard_normal((60, 1)).astype(np.float32))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[3] = rows[17]
    for dtype, odt in (("bf16", 1), ("f32", 0), ("f32s", 0)):
        ix = m.HipKnnIndex(d, dtype=dtype)
        try:
            ix.add(rows)
            if odt not in cache:
                cache[odt] = o.batch_top_k(q, rows, k, dtype=odt)
            ri, rs = cache[odt]
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
                for _ in range(2):
                    ix.search_host_ptr(qptr, nq, k, iptr, sptr)
                gi, gs = get()
                assert np.array_equal(gi, ri) and np.array_equal(gs, rs), dtype
            bad = qp.clone().pin_memory()
            bad[7, 5] = float("nan")
            try:
                ix.search_host_ptr(bad.data_ptr(), nq, k, oi.data_ptr(), osc.data_ptr())
                raise AssertionError("NaN query accepted")
            except m.CgvError as e:
                assert e.code == m.cgvec.CGV_ERR_NONFINITE
            ix.search_host_ptr(qp.data_ptr(), nq, k, oi.data_ptr(), osc.data_ptr())
            assert np.array_equal(oi.numpy().view(np.uint64), ri) and np.array_equal(osc.numpy(), rs)
            di, ds = ix.search_from_pinned(qp, k)
            assert np.array_equal(di.cpu().numpy().view(np.uint64), ri) and np.array_equal(ds.cpu().numpy(), rs)
            rt = torch.cuda.cudart()
            if form == "heap":
                reg = np.ascontiguousarray(q.copy())
                addr, length = reg.ctypes.data, reg.nbytes
            else:
                mm = mmap.mmap(-1, (q.nbytes + 4095) // 4096 * 4096)
                KEEP.append(mm)
                reg = np.frombuffer(mm, dtype=np.float32, count=q.size).reshape(q.shape)
                reg[:] = q
                addr, length = reg.ctypes.data, len(mm)
            if int(rt.cudaHostRegister(addr, length, 0)) == 0:
                try:
                    ix.search_host_ptr(addr, nq, k, oi.data_ptr(), osc.data_ptr())
                    assert np.array_equal(oi.numpy().view(np.uint64), ri) and np.array_equal(osc.numpy(), rs)
                finally:
                    if form != "heap":
                        torch.cuda.synchronize()
                    rc = int(rt.cudaHostUnregister(addr))
                    if rc:
                        print("unregister rc", rc, flush=True)
            del reg
        finally:
            ix.close()


def following_test(cache):
    """tests/test_gpu_sharded.py::test_id_map_reports_block_cyclic_global_ids - where both aborts happened"""
    rng = np.random.default_rng(1)
    n, d, G, s = 3 * C + 77, 64, 3, 1
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    q = rng.standard_normal((9, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype="bf16")
    try:
        ix.add(rows)
        ix.set_id_map(C, G, s)
        idx, sc = ix.search(q, 10)
        if "idmap" not in cache:
            cache["idmap"] = o.batch_top_k(q, rows, 10, dtype=1)
        ri, rs = cache["idmap"]
        exp = ((ri // C) * G + s) * C + ri % C
        assert np.array_equal(idx, exp.astype(np.uint64)) and np.array_equal(sc, rs)
    finally:
        ix.close()


limit = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
form = sys.argv[2] if len(sys.argv) > 2 else "both"
cache = {}
t0 = time.time()
it = 0
while time.time() - t0 < limit:
    f = form if form != "both" else ("heap" if it % 2 == 0 else "mmap")
    pinned_case(f, cache)
    if it % 3 == 2:
        gc.collect()
    following_test(cache)
    it += 1
    if it % 5 == 0:
        print("iterations", it, "elapsed %.1f" % (time.time() - t0), flush=True)
print("stress done: iterations", it, "form", form, flush=True)
