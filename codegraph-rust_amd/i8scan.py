"""ctypes binding of include/cgvec_i8.h: the reference's int8 'optimized' scan
(crates/codegraph-vector/src/optimization.rs:63-150, 212-283) on the GPU."""
import ctypes as C

import numpy as np

from .cgvec import CGV_ERR_DIM_MISMATCH, CgvError, _check, lib as _base_lib

_decl = False


def lib():
    global _decl
    L = _base_lib()
    if not _decl:
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.cgv_i8_create.argtypes = [u32, i32, C.POINTER(vp)]
        L.cgv_i8_destroy.argtypes = [vp]
        L.cgv_i8_add_u8.argtypes = [vp, vp, u64]
        L.cgv_i8_add_f32.argtypes = [vp, vp, u64]
        L.cgv_i8_count.argtypes = [vp]
        L.cgv_i8_count.restype = u64
        L.cgv_i8_get_row_u8.argtypes = [vp, u64, vp]
        L.cgv_i8_search_optimized.argtypes = [vp, vp, u32, u64, vp, C.POINTER(u64)]
        L.cgv_i8_scores_f32.argtypes = [vp, vp, u32, vp]
        L.cgv_quantize_u8_f32.argtypes = [i32, vp, u64, u32, vp]
        L.cgv_quantize_u4_f32.argtypes = [i32, vp, u64, u32, vp]
        for n in ("cgv_i8_create", "cgv_i8_destroy", "cgv_i8_add_u8", "cgv_i8_add_f32", "cgv_i8_get_row_u8",
                  "cgv_i8_search_optimized", "cgv_i8_scores_f32", "cgv_quantize_u8_f32", "cgv_quantize_u4_f32"):
            getattr(L, n).restype = i32
        _decl = True
    return L


def quantize_u8(rows, device=0):
    """ModelOptimizer::quantize_batch, 8-bit arm (optimization.rs:226-283) -> uint8 [n, dim]."""
    r = np.ascontiguousarray(rows, dtype=np.float32)
    if r.ndim == 1:
        r = r[None, :]
    out = np.empty(r.shape, dtype=np.uint8)
    _check(lib().cgv_quantize_u8_f32(device, r.ctypes.data_as(C.c_void_p), r.shape[0], r.shape[1],
                                     out.ctypes.data_as(C.c_void_p)))
    return out


def quantize_u4(rows, device=0):
    """ModelOptimizer::quantize_batch, 4-bit arm (optimization.rs:248-262) -> uint8 [n, ceil(dim/2)]."""
    r = np.ascontiguousarray(rows, dtype=np.float32)
    if r.ndim == 1:
        r = r[None, :]
    out = np.empty((r.shape[0], (r.shape[1] + 1) // 2), dtype=np.uint8)
    _check(lib().cgv_quantize_u4_f32(device, r.ctypes.data_as(C.c_void_p), r.shape[0], r.shape[1],
                                     out.ctypes.data_as(C.c_void_p)))
    return out


class Int8ScanIndex:
    """OptimizationResult{optimized_data, metadata} resident in HBM."""

    def __init__(self, dim, device=0):
        self.dim = int(dim)
        h = C.c_void_p()
        _check(lib().cgv_i8_create(self.dim, int(device), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().cgv_i8_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().cgv_i8_count(self._h))

    def add_u8(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        if d.ndim != 2 or d.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"data shape {d.shape} != (*, {self.dim})")
        _check(lib().cgv_i8_add_u8(self._h, d.ctypes.data_as(C.c_void_p), d.shape[0]))

    def add(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.float32)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"rows shape {r.shape} != (*, {self.dim})")
        _check(lib().cgv_i8_add_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def get_row(self, i):
        out = np.empty(self.dim, dtype=np.uint8)
        _check(lib().cgv_i8_get_row_u8(self._h, int(i), out.ctypes.data_as(C.c_void_p)))
        return out

    def search_optimized(self, query, limit):
        """OptimizationResult::search_optimized -> row indices (uint64)."""
        q = np.ascontiguousarray(query, dtype=np.float32).ravel()
        out = np.empty(max(int(limit), 1), dtype=np.uint64)
        n = C.c_uint64(0)
        _check(lib().cgv_i8_search_optimized(self._h, q.ctypes.data_as(C.c_void_p), q.size, int(limit),
                                             out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[:n.value]

    def scores(self, query):
        q = np.ascontiguousarray(query, dtype=np.float32).ravel()
        out = np.empty(len(self), dtype=np.float32)
        _check(lib().cgv_i8_scores_f32(self._h, q.ctypes.data_as(C.c_void_p), q.size, out.ctypes.data_as(C.c_void_p)))
        return out
