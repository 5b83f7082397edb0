"""ctypes binding of include/cgvec_quant.h: the reference's ScalarQuantizer / ProductQuantizer
(crates/codegraph-vector/src/persistent.rs:116-477) trained and applied on the GPU."""
import ctypes as C

import numpy as np

from .cgvec import CGV_ERR_DIM_MISMATCH, CgvError, _check, lib as _base_lib

_decl = False


def lib():
    global _decl
    L = _base_lib()
    if not _decl:
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.cgv_sq_create.argtypes = [u32, u32, i32, i32, C.POINTER(vp)]
        L.cgv_sq_destroy.argtypes = [vp]
        L.cgv_sq_train_f32.argtypes = [vp, vp, u64]
        L.cgv_sq_params.argtypes = [vp, vp, vp]
        L.cgv_sq_bytes_per_value.argtypes = [vp]
        L.cgv_sq_bytes_per_value.restype = u32
        L.cgv_sq_encode_f32.argtypes = [vp, vp, u64, vp]
        L.cgv_sq_decode.argtypes = [vp, vp, u64, vp]
        L.cgv_pq_create.argtypes = [u32, u32, u32, i32, C.POINTER(vp)]
        L.cgv_pq_destroy.argtypes = [vp]
        L.cgv_pq_train_f32.argtypes = [vp, vp, u64]
        L.cgv_pq_centroids.argtypes = [vp, vp]
        L.cgv_pq_encode_f32.argtypes = [vp, vp, u64, vp]
        L.cgv_pq_decode.argtypes = [vp, vp, u64, vp]
        for n in ("cgv_sq_create", "cgv_sq_destroy", "cgv_sq_train_f32", "cgv_sq_params", "cgv_sq_encode_f32",
                  "cgv_sq_decode", "cgv_pq_create", "cgv_pq_destroy", "cgv_pq_train_f32", "cgv_pq_centroids",
                  "cgv_pq_encode_f32", "cgv_pq_decode"):
            getattr(L, n).restype = i32
        _decl = True
    return L


def _rows(a, dim):
    r = np.ascontiguousarray(a, dtype=np.float32)
    if r.ndim != 2 or r.shape[1] != dim:
        raise CgvError(CGV_ERR_DIM_MISMATCH, f"rows shape {r.shape} != (*, {dim})")
    return r


class ScalarQuantizer:
    """persistent.rs:331-477"""

    def __init__(self, dim, nbits=8, uniform=False, device=0):
        self.dim, self.nbits = int(dim), int(nbits)
        h = C.c_void_p()
        _check(lib().cgv_sq_create(self.dim, self.nbits, int(bool(uniform)), int(device), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().cgv_sq_destroy(self._h)
            self._h = None

    def train(self, rows):
        r = _rows(rows, self.dim)
        _check(lib().cgv_sq_train_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def params(self):
        sc, bi = np.empty(self.dim, np.float32), np.empty(self.dim, np.float32)
        _check(lib().cgv_sq_params(self._h, sc.ctypes.data_as(C.c_void_p), bi.ctypes.data_as(C.c_void_p)))
        return sc, bi

    def encode(self, rows):
        r = _rows(rows, self.dim)
        out = np.empty((r.shape[0], self.dim * int(lib().cgv_sq_bytes_per_value(self._h))), np.uint8)
        _check(lib().cgv_sq_encode_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def decode(self, codes):
        c = np.ascontiguousarray(codes, np.uint8)
        out = np.empty((c.shape[0], self.dim), np.float32)
        _check(lib().cgv_sq_decode(self._h, c.ctypes.data_as(C.c_void_p), c.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out


class ProductQuantizer:
    """persistent.rs:116-329"""

    def __init__(self, dim, m, nbits=8, device=0):
        self.dim, self.m, self.nbits = int(dim), int(m), int(nbits)
        h = C.c_void_p()
        _check(lib().cgv_pq_create(self.dim, self.m, self.nbits, int(device), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().cgv_pq_destroy(self._h)
            self._h = None

    def train(self, rows):
        r = _rows(rows, self.dim)
        _check(lib().cgv_pq_train_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def centroids(self):
        out = np.empty((self.m, 1 << self.nbits, self.dim // self.m), np.float32)
        _check(lib().cgv_pq_centroids(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def encode(self, rows):
        r = _rows(rows, self.dim)
        out = np.empty((r.shape[0], self.m), np.uint8)
        _check(lib().cgv_pq_encode_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def decode(self, codes):
        c = np.ascontiguousarray(codes, np.uint8)
        out = np.empty((c.shape[0], self.dim), np.float32)
        _check(lib().cgv_pq_decode(self._h, c.ctypes.data_as(C.c_void_p), c.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out
