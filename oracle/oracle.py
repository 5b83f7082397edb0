"""ctypes/numpy front-end of the CPU oracle (oracle/cgv_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg — never by the product package. See the header of
cgv_oracle.cpp for the reference file:line each function restates and for the
pinning status of the oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

F32, BF16, FP16, FP8 = 0, 1, 2, 3
COSINE, DOT, COSINE_SEQ, COSINE_SCALAR = 0, 1, 2, 3   # COSINE_SCALAR: simd_ops.rs:257-278 for every length (non-AVX2 host)


def build(force=False):
    src = os.path.join(_HERE, "cgv_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        u64p = C.POINTER(C.c_uint64)
        u8p = C.POINTER(C.c_uint8)
        for name in ("cgo_cosine_avx2", "cgo_cosine_scalar", "cgo_cosine_adaptive", "cgo_dot_avx2",
                     "cgo_l2_avx2", "cgo_search_cosine", "cgo_cosine_distance"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [fp, fp, C.c_size_t]
        L.cgo_normalize_avx2.restype = None
        L.cgo_normalize_avx2.argtypes = [fp, C.c_size_t]
        L.cgo_synth_rows.restype = None
        L.cgo_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, fp]
        L.cgo_philox4x32_10.restype = None
        L.cgo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.cgo_normalize_scalar.restype = None
        L.cgo_normalize_scalar.argtypes = [fp, C.c_size_t]
        L.cgo_parallel_top_k_flat.restype = C.c_int
        L.cgo_parallel_top_k_flat.argtypes = [fp, fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                              C.c_int, u64p, fp]
        L.cgo_rowset_create.restype = C.c_void_p
        L.cgo_rowset_create.argtypes = [fp, C.c_uint64, C.c_uint64]
        L.cgo_rowset_destroy.restype = None
        L.cgo_rowset_destroy.argtypes = [C.c_void_p]
        L.cgo_rowset_top_k.restype = C.c_int
        L.cgo_rowset_top_k.argtypes = [C.c_void_p, fp, C.c_uint64, C.c_int, C.c_int, u64p, fp]
        L.cgo_last_timing.restype = None
        L.cgo_last_timing.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cgo_prefetch_k.restype = C.c_uint64
        L.cgo_prefetch_k.argtypes = [C.c_uint64]
        L.cgo_normalize_scores.restype = None
        L.cgo_normalize_scores.argtypes = [fp, C.c_size_t]
        L.cgo_search_baseline.restype = C.c_uint64
        L.cgo_search_baseline.argtypes = [fp, fp, C.c_uint64, C.c_uint64, C.c_uint64, u64p, fp]
        L.cgo_simple_hash.restype = C.c_uint32
        L.cgo_simple_hash.argtypes = [C.c_char_p, C.c_size_t]
        L.cgo_hash_embed.restype = None
        L.cgo_hash_embed.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, fp]
        L.cgo_quantize_u8.restype = None
        L.cgo_quantize_u8.argtypes = [fp, C.c_size_t, u8p]
        L.cgo_search_optimized_u8.restype = C.c_uint64
        L.cgo_search_optimized_u8.argtypes = [fp, u8p, C.c_uint64, C.c_uint64, C.c_uint64, u64p]
        L.cgo_round_trip_fp8_rows.restype = None
        L.cgo_round_trip_fp8_rows.argtypes = [fp, C.c_size_t, C.c_size_t, fp, C.c_int]
        L.cgo_fp8_row_exponent.restype = C.c_int
        L.cgo_fp8_row_exponent.argtypes = [fp, C.c_size_t]
        L.cgo_round_trip.restype = None
        L.cgo_round_trip.argtypes = [fp, C.c_size_t, C.c_int, fp]
        L.cgo_f32_to_e4m3.restype = C.c_uint8
        L.cgo_f32_to_e4m3.argtypes = [C.c_float]
        L.cgo_e4m3_to_f32.restype = C.c_float
        L.cgo_e4m3_to_f32.argtypes = [C.c_uint8]
        L.cgo_f32_to_bf16.restype = C.c_uint16
        L.cgo_f32_to_bf16.argtypes = [C.c_float]
        L.cgo_merge_topk.restype = None
        L.cgo_merge_topk.argtypes = [u64p, fp, C.c_uint64, C.c_uint64, u64p, fp]
        L.cgo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _pairfn(name):
    def fn(a, b):
        a, pa = _f(a)
        b, pb = _f(b)
        if a.shape != b.shape:
            raise ValueError("dimension mismatch")  # VectorError::DimensionMismatch
        return float(getattr(lib(), name)(pa, pb, a.size))
    fn.__name__ = name
    return fn


cosine_avx2 = _pairfn("cgo_cosine_avx2")          # simd_ops.rs:15-78
cosine_scalar = _pairfn("cgo_cosine_scalar")      # simd_ops.rs:257-278
cosine_adaptive = _pairfn("cgo_cosine_adaptive")  # simd_ops.rs:281-295
dot_avx2 = _pairfn("cgo_dot_avx2")                # simd_ops.rs:149-183
l2_avx2 = _pairfn("cgo_l2_avx2")                  # simd_ops.rs:105-143
search_cosine = _pairfn("cgo_search_cosine")      # search.rs:519-533
cosine_distance = _pairfn("cgo_cosine_distance")  # optimization.rs:404-418


def normalize_avx2(v):
    """simd_ops.rs:189-222 (returns a normalised copy)."""
    v = np.array(v, dtype=np.float32, copy=True)
    lib().cgo_normalize_avx2(v.ctypes.data_as(C.POINTER(C.c_float)), v.size)
    return v


def normalize_rows(m, arm="avx2"):
    """simd_ops.rs:386-419 parallel_normalize_vectors, row by row: the AVX2 branch (:393, normalize_avx2) or the
    scalar branch a host without AVX2 + FMA / a non-x86_64 host takes (:394-403, :406-415)."""
    m = np.array(m, dtype=np.float32, copy=True)
    fn = lib().cgo_normalize_avx2 if arm == "avx2" else lib().cgo_normalize_scalar
    for r in m:
        fn(r.ctypes.data_as(C.POINTER(C.c_float)), r.size)
    return m


def parallel_top_k(query, rows, k, metric=COSINE, threads=0):
    """simd_ops.rs:361-383 -> (idx uint64[k], score f32[k]); ties: index asc.
    Pads with (UINT64_MAX, -inf) when len(rows) < k. Raises on NaN (reference panics)."""
    q, pq = _f(query)
    r, pr = _f(rows)
    n, d = r.shape
    if q.size != d:
        raise ValueError("dimension mismatch")
    idx = np.empty(k, dtype=np.uint64)
    sc = np.empty(k, dtype=np.float32)
    rc = lib().cgo_parallel_top_k_flat(pq, pr, n, d, k, metric, threads,
                                       idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       sc.ctypes.data_as(C.POINTER(C.c_float)))
    if rc < 0:
        raise FloatingPointError("NaN similarity (reference panics at simd_ops.rs:379)")
    return idx, sc


def batch_top_k(queries, rows, k, metric=COSINE, threads=0, dtype=F32):
    """N independent single-query searches (the reference's only 'batch' shape,
    search.rs:358-361) on rounded-then-upcast inputs."""
    qs = round_trip(queries, dtype, fp8_codes=True)
    rs = round_trip(rows, dtype, fp8_codes=True)
    idx = np.empty((qs.shape[0], k), dtype=np.uint64)
    sc = np.empty((qs.shape[0], k), dtype=np.float32)
    for i in range(qs.shape[0]):
        idx[i], sc[i] = parallel_top_k(qs[i], rs, k, metric, threads)
    return idx, sc


class RowSet:
    """N separately allocated f32 rows (the reference's &[Vec<f32>]); CPU-baseline layout."""

    def __init__(self, rows):
        r, pr = _f(rows)
        self.n, self.d = r.shape
        self._h = lib().cgo_rowset_create(pr, self.n, self.d)

    def top_k(self, query, k, metric=COSINE, threads=0):
        q, pq = _f(query)
        idx = np.empty(k, dtype=np.uint64)
        sc = np.empty(k, dtype=np.float32)
        rc = lib().cgo_rowset_top_k(self._h, pq, k, metric, threads,
                                    idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    sc.ctypes.data_as(C.POINTER(C.c_float)))
        if rc < 0:
            raise FloatingPointError("NaN similarity")
        return idx, sc

    def close(self):
        if self._h:
            lib().cgo_rowset_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def last_timing():
    """(score_ms, sort_ms) of the calling thread's last parallel_top_k / RowSet.top_k."""
    a, b = C.c_double(0.0), C.c_double(0.0)
    lib().cgo_last_timing(C.byref(a), C.byref(b))
    return a.value, b.value


def numa_nodes():
    try:
        return max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]))
    except OSError:
        return 1


def prefetch_k(limit):
    return int(lib().cgo_prefetch_k(limit))  # search.rs:113


def synth_rows(seed, row0, nrows, dim, normalise=True):
    """rows [row0, row0 + nrows) of the counter-based stream `seed` (Philox4x32-10 keyed
    (seed, row, col) -> N(0, 1) f32 -> L2-normalised in f32); the CPU side of the contract in csrc/synth.hip."""
    out = np.empty((nrows, dim), dtype=np.float32)
    if nrows and dim:
        lib().cgo_synth_rows(seed, row0, nrows, dim, 1 if normalise else 0, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) for x in key])
    o = (C.c_uint32 * 4)()
    lib().cgo_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def normalize_scores(s):
    s = np.array(s, dtype=np.float32, copy=True)  # search.rs:574-592
    lib().cgo_normalize_scores(s.ctypes.data_as(C.POINTER(C.c_float)), s.size)
    return s


def search_baseline(query, rows, limit):
    """optimization.rs:376-402 -> (idx, distance)."""
    q, pq = _f(query)
    r, pr = _f(rows)
    n, d = r.shape if r.ndim == 2 else (0, q.size)
    idx = np.empty(max(limit, 1), dtype=np.uint64)
    dist = np.empty(max(limit, 1), dtype=np.float32)
    m = lib().cgo_search_baseline(pq, pr, n, d, limit, idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  dist.ctypes.data_as(C.POINTER(C.c_float)))
    return idx[:m], dist[:m]


def simple_hash(text):
    b = text.encode("utf-8")
    return int(lib().cgo_simple_hash(b, len(b)))  # search.rs:535-541


def hash_embed(text, dim=384):
    b = text.encode("utf-8")
    out = np.empty(dim, dtype=np.float32)
    lib().cgo_hash_embed(b, len(b), dim, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out  # search.rs:178-205


def quantize_u8(v):
    v, pv = _f(v)
    out = np.empty(v.size, dtype=np.uint8)
    lib().cgo_quantize_u8(pv, v.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(v.shape)  # optimization.rs:212-283


def quantize_u4(v):
    v, pv = _f(v)
    n, dim = v.shape
    out = np.empty((n, (dim + 1) // 2), dtype=np.uint8)
    L = lib()
    L.cgo_quantize_u4.restype = None
    L.cgo_quantize_u4.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint8)]
    L.cgo_quantize_u4(pv, n, dim, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out  # optimization.rs:248-262


def search_optimized_u8(query, data_u8, limit):
    q, pq = _f(query)
    d = np.ascontiguousarray(data_u8, dtype=np.uint8)
    n, dim = d.shape
    idx = np.empty(max(limit, 1), dtype=np.uint64)
    m = lib().cgo_search_optimized_u8(pq, d.ctypes.data_as(C.POINTER(C.c_uint8)), n, dim, limit,
                                      idx.ctypes.data_as(C.POINTER(C.c_uint64)))
    return idx[:m]  # optimization.rs:63-150


def trigram_jaccard(a, b):
    L = lib()
    L.cgo_trigram_jaccard.restype = C.c_float
    L.cgo_trigram_jaccard.argtypes = [C.c_char_p, C.c_char_p]
    return float(L.cgo_trigram_jaccard(a.encode(), b.encode()))  # indexer.rs:2901-2932


def symbol_match_phase2(target, target_emb, names, embs, threshold=0.75):
    """indexer.rs:2790-2843 for one unresolved symbol -> (index or -1, similarity)."""
    L = lib()
    L.cgo_symbol_match_phase2.restype = C.c_int64
    L.cgo_symbol_match_phase2.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_float), C.c_uint64, C.c_float, C.POINTER(C.c_float)]
    t, pt = _f(target_emb)
    e, pe = _f(embs)
    arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
    sc = C.c_float(0.0)
    i = L.cgo_symbol_match_phase2(target.encode(), pt, len(names), arr, pe, e.shape[1], threshold, C.byref(sc))
    return int(i), float(np.float32(sc.value))


def round_trip(a, dtype, scaled_fp8=True, fp8_codes=False):
    """f32 -> storage dtype (RNE) -> f32. FP8 rows use the per-row power-of-two scale of the
    storage format (last axis = row) unless scaled_fp8=False; fp8_codes=True returns the e4m3
    codes (x * 2^e rounded) instead of the de-scaled values - the domain fp8 scores are defined in."""
    a, pa = _f(a)
    if dtype == F32:
        return a
    out = np.empty_like(a)
    if dtype == FP8 and scaled_fp8:
        d = a.shape[-1] if a.ndim else 1
        lib().cgo_round_trip_fp8_rows(pa, a.size // max(d, 1), d, out.ctypes.data_as(C.POINTER(C.c_float)),
                                      0 if fp8_codes else 1)
        return out
    lib().cgo_round_trip(pa, a.size, dtype, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def merge_topk(idx, score, k):
    """Merge G partial top-k lists (G,k) -> top-k by (score desc, idx asc)."""
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    score = np.ascontiguousarray(score, dtype=np.float32)
    g = idx.size // k
    oi = np.empty(k, dtype=np.uint64)
    os_ = np.empty(k, dtype=np.float32)
    lib().cgo_merge_topk(idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                         score.ctypes.data_as(C.POINTER(C.c_float)), g, k,
                         oi.ctypes.data_as(C.POINTER(C.c_uint64)),
                         os_.ctypes.data_as(C.POINTER(C.c_float)))
    return oi, os_


def max_threads():
    return int(lib().cgo_max_threads())


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def sq_train(v, nbits, uniform):
    """persistent.rs:355-408 -> (scales, biases)"""
    v, pv = _f(v)
    n, dim = v.shape
    sc, bi = np.empty(dim, np.float32), np.empty(dim, np.float32)
    L = lib()
    L.cgo_sq_train.restype = None
    L.cgo_sq_train.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                               C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.cgo_sq_train(pv, n, dim, nbits, int(uniform), sc.ctypes.data_as(C.POINTER(C.c_float)),
                   bi.ctypes.data_as(C.POINTER(C.c_float)))
    return sc, bi


def sq_encode(v, nbits, scales, biases):
    v, pv = _f(v)
    n, dim = v.shape
    bpv = 1 if nbits == 8 else (2 if nbits == 16 else 4)
    out = np.empty((n, dim * bpv), np.uint8)
    L = lib()
    L.cgo_sq_encode.restype = None
    L.cgo_sq_encode.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_float),
                                C.POINTER(C.c_float), C.POINTER(C.c_uint8)]
    L.cgo_sq_encode(pv, n, dim, nbits, _f(scales)[1], _f(biases)[1], _u8p(out))
    return out


def sq_decode(codes, dim, nbits, scales, biases):
    codes = np.ascontiguousarray(codes, np.uint8)
    n = codes.shape[0]
    out = np.empty((n, dim), np.float32)
    L = lib()
    L.cgo_sq_decode.restype = None
    L.cgo_sq_decode.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_float),
                                C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.cgo_sq_decode(_u8p(codes), n, dim, nbits, _f(scales)[1], _f(biases)[1], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def pq_train(v, m, nbits):
    """persistent.rs:155-187, 245-318 -> centroids [m, 2^nbits, dim/m]"""
    v, pv = _f(v)
    n, dim = v.shape
    cent = np.zeros((m, 1 << nbits, dim // m), np.float32)
    L = lib()
    L.cgo_pq_train.restype = None
    L.cgo_pq_train.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]
    L.cgo_pq_train(pv, n, dim, m, nbits, cent.ctypes.data_as(C.POINTER(C.c_float)))
    return cent


def pq_encode(v, cent):
    v, pv = _f(v)
    n, dim = v.shape
    m, ksub, _ = cent.shape
    codes = np.empty((n, m), np.uint8)
    L = lib()
    L.cgo_pq_encode.restype = None
    L.cgo_pq_encode.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_float),
                                C.POINTER(C.c_uint8)]
    L.cgo_pq_encode(pv, n, dim, m, int(ksub).bit_length() - 1, _f(cent)[1], _u8p(codes))
    return codes
