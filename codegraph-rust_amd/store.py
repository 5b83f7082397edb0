"""ctypes binding of the host-side mirror (include/cgvec_store.h): SurrealVectorStore +
SemanticSearch over the HIP kNN backend. Plumbing for tests; the logic is C++ in the library."""
import ctypes as C
import uuid

import numpy as np

from . import cgvec
from .cgvec import CgvError, _check

OR_MAX, AND_AVERAGE = 0, 1


class _Filters(C.Structure):
    _fields_ = [("languages", C.POINTER(C.c_char_p)), ("n_languages", C.c_uint32),
                ("node_types", C.POINTER(C.c_char_p)), ("n_node_types", C.c_uint32),
                ("attr_keys", C.POINTER(C.c_char_p)), ("attr_values", C.POINTER(C.c_char_p)), ("n_attrs", C.c_uint32),
                ("path_prefixes", C.POINTER(C.c_char_p)), ("n_path_prefixes", C.c_uint32)]


_bound = False


def _lib():
    global _bound
    L = cgvec.lib()
    if not _bound:
        vp, u32 = C.c_void_p, C.c_uint32
        L.cgvs_store_create.argtypes = [C.c_int, C.c_int, u32, C.POINTER(vp)]
        L.cgvs_store_create_sharded.argtypes = [C.c_int, u32, C.POINTER(C.c_int), u32, C.POINTER(vp)]
        L.cgvs_store_create_mock.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), u32, u32, C.POINTER(vp)]
        L.cgvs_mock_recorded_columns.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.cgvs_store_destroy.argtypes = [vp]
        L.cgvs_upsert_nodes.argtypes = [vp, u32, vp, vp, u32]
        L.cgvs_upsert_node_metadata.argtypes = [vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p),
                                                C.POINTER(C.c_char_p), u32]
        L.cgvs_vector_knn.argtypes = [vp, C.c_char_p, vp, u32, u32, u32, vp, vp, C.POINTER(u32)]
        L.cgvs_search_similar.argtypes = [vp, vp, u32, u32, vp, C.POINTER(u32)]
        L.cgvs_get_embedding.argtypes = [vp, vp, vp, u32, C.POINTER(u32)]
        L.cgvs_search_by_embedding.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u32)]
        L.cgvs_search_by_text.argtypes = [vp, C.c_char_p, u32, vp, vp, C.POINTER(u32)]
        L.cgvs_semantic_search.argtypes = [vp, vp, u32, C.POINTER(_Filters), u32, vp, vp, C.POINTER(u32)]
        L.cgvs_hybrid_search.argtypes = [vp, vp, u32, C.POINTER(_Filters), C.c_float, u32, vp, vp, C.POINTER(u32)]
        L.cgvs_multi_vector_search.argtypes = [vp, vp, u32, u32, C.c_int, C.POINTER(_Filters), u32, vp, vp, C.POINTER(u32)]
        L.cgvs_combine_embeddings.argtypes = [vp, u32, u32, vp]
        L.cgvs_embedding_column_for_dimension.argtypes = [u32]
        L.cgvs_embedding_column_for_dimension.restype = C.c_char_p
        L.cgvs_normalize_node_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.cgvs_parse_node_id.argtypes = [C.c_char_p, vp]
        L.cgvs_format_node_id.argtypes = [vp, C.c_char_p]
        L.cgvs_simple_hash.argtypes = [C.c_char_p]
        L.cgvs_simple_hash.restype = u32
        L.cgvs_hash_embed.argtypes = [C.c_char_p, u32, vp]
        L.cgvs_prefetch_k.argtypes = [C.c_uint64]
        L.cgvs_prefetch_k.restype = C.c_uint64
        L.cgvs_normalize_scores.argtypes = [vp, u32]
        L.cgvs_normalize_scores.restype = None
        L.cgvs_cosine_similarity.argtypes = [vp, vp, u32]
        L.cgvs_cosine_similarity.restype = C.c_float
        L.cgvs_resolver_create.argtypes = [u32, C.c_int, C.c_int, C.POINTER(vp)]
        L.cgvs_resolver_destroy.argtypes = [vp]
        L.cgvs_resolver_add_symbols.argtypes = [vp, u32, C.POINTER(C.c_char_p), vp]
        L.cgvs_resolver_count.argtypes = [vp]
        L.cgvs_resolver_count.restype = C.c_uint64
        L.cgvs_resolver_match.argtypes = [vp, u32, C.POINTER(C.c_char_p), vp, C.c_float, vp, vp]
        L.cgvs_trigram_jaccard.argtypes = [C.c_char_p, C.c_char_p]
        L.cgvs_trigram_jaccard.restype = C.c_float
        L.cgvs_symbol_name_eligible.argtypes = [C.c_char_p, C.c_char_p]
        L.cgvs_rerank_embeddings.argtypes = [C.c_int, vp, vp, u32, u32, vp, vp]
        _bound = True
    return L


def _ids_to_bytes(ids):
    return b"".join(i.bytes for i in ids)


def _cstrs(strs):
    arr = (C.c_char_p * max(len(strs), 1))(*[s.encode() for s in strs])
    return arr


def _filters(f):
    """f: dict(languages=[...]|None, node_types=[...]|None, attribute_equals={...}, path_prefixes=[...])"""
    if f is None:
        return None, []
    keep = []
    fs = _Filters()
    if f.get("languages") is not None:
        a = _cstrs(list(f["languages"])); keep.append(a)
        fs.languages, fs.n_languages = C.cast(a, C.POINTER(C.c_char_p)), len(f["languages"])
    if f.get("node_types") is not None:
        a = _cstrs(list(f["node_types"])); keep.append(a)
        fs.node_types, fs.n_node_types = C.cast(a, C.POINTER(C.c_char_p)), len(f["node_types"])
    attrs = f.get("attribute_equals") or {}
    if attrs:
        k, v = _cstrs(list(attrs.keys())), _cstrs(list(attrs.values())); keep += [k, v]
        fs.attr_keys, fs.attr_values, fs.n_attrs = C.cast(k, C.POINTER(C.c_char_p)), C.cast(v, C.POINTER(C.c_char_p)), len(attrs)
    pp = f.get("path_prefixes") or []
    if pp:
        a = _cstrs(list(pp)); keep.append(a)
        fs.path_prefixes, fs.n_path_prefixes = C.cast(a, C.POINTER(C.c_char_p)), len(pp)
    keep.append(fs)
    return C.byref(fs), keep


class VectorStore:
    """SurrealVectorStore + SemanticSearch mirror. NodeIds are uuid.UUID."""

    def __init__(self, dtype="bf16", device=0, ef_search=100, _mock=None, devices=None):
        """devices = [d0, d1, ...]: ONE store over several GPUs (cgvs_store_create_sharded; a device may repeat)."""
        self._h = C.c_void_p()
        L = _lib()
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(L.cgvs_store_create_sharded(cgvec.DTYPES[dtype], len(devices), devs, ef_search, C.byref(self._h)))
        elif _mock is not None:
            ids = _cstrs([m[0] for m in _mock])
            d = (C.c_float * max(len(_mock), 1))(*[m[1] for m in _mock])
            _check(L.cgvs_store_create_mock(C.cast(ids, C.POINTER(C.c_char_p)), d, len(_mock), ef_search, C.byref(self._h)))
        else:
            _check(L.cgvs_store_create(cgvec.DTYPES[dtype], device, ef_search, C.byref(self._h)))

    @classmethod
    def with_mock_backend(cls, results, ef_search=128):
        return cls(_mock=results, ef_search=ef_search)

    def close(self):
        if self._h:
            _lib().cgvs_store_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def recorded_columns(self):
        buf = C.create_string_buffer(4096)
        _check(_lib().cgvs_mock_recorded_columns(self._h, buf, 4096))
        return [c for c in buf.value.decode().split("\n") if c]

    def store_embeddings(self, ids, embeddings):
        e = np.ascontiguousarray(embeddings, dtype=np.float32)
        b = _ids_to_bytes(ids)
        _check(_lib().cgvs_upsert_nodes(self._h, len(ids), b, e.ctypes.data_as(C.c_void_p), e.shape[1] if e.ndim == 2 else 0))

    upsert_nodes = store_embeddings

    def upsert_node_metadata(self, node_id, language=None, node_type=None, file_path="", attributes=None):
        attributes = attributes or {}
        k, v = _cstrs(list(attributes.keys())), _cstrs(list(attributes.values()))
        _check(_lib().cgvs_upsert_node_metadata(self._h, node_id.bytes, language.encode() if language else None,
                                                node_type.encode() if node_type else None, file_path.encode(),
                                                C.cast(k, C.POINTER(C.c_char_p)), C.cast(v, C.POINTER(C.c_char_p)), len(attributes)))

    def _out(self, cap):
        return C.create_string_buffer(16 * max(cap, 1)), np.empty(max(cap, 1), np.float32), C.c_uint32(0)

    @staticmethod
    def _ids(buf, n):
        return [uuid.UUID(bytes=buf.raw[16 * i:16 * i + 16]) for i in range(n)]

    def vector_knn(self, column, query, limit, ef_search=100):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = C.create_string_buffer(48 * max(limit, 1))
        d = np.empty(max(limit, 1), np.float32)
        n = C.c_uint32(0)
        _check(_lib().cgvs_vector_knn(self._h, column.encode(), q.ctypes.data_as(C.c_void_p), q.size, limit, ef_search,
                                      ids, d.ctypes.data_as(C.c_void_p), C.byref(n)))
        return [(ids.raw[48 * i:48 * i + 48].split(b"\0")[0].decode(), float(d[i])) for i in range(n.value)]

    def search_similar(self, query, limit):
        q = np.ascontiguousarray(query, dtype=np.float32)
        buf, _, n = self._out(limit)
        _check(_lib().cgvs_search_similar(self._h, q.ctypes.data_as(C.c_void_p), q.size, limit, buf, C.byref(n)))
        return self._ids(buf, n.value)

    def get_embedding(self, node_id, cap=8192):
        out = np.empty(cap, np.float32)
        d = C.c_uint32(0)
        _check(_lib().cgvs_get_embedding(self._h, node_id.bytes, out.ctypes.data_as(C.c_void_p), cap, C.byref(d)))
        return None if d.value == 0 else out[:d.value].copy()

    def _scored(self, fn, *args, cap):
        buf, sc, n = self._out(cap)
        _check(fn(self._h, *args, buf, sc.ctypes.data_as(C.c_void_p), C.byref(n)))
        return list(zip(self._ids(buf, n.value), sc[:n.value].copy()))

    def search_by_embedding(self, query, limit):
        q = np.ascontiguousarray(query, dtype=np.float32)
        return self._scored(_lib().cgvs_search_by_embedding, q.ctypes.data_as(C.c_void_p), q.size, limit, cap=limit)

    def search_by_text(self, text, limit):
        return self._scored(_lib().cgvs_search_by_text, text.encode(), limit, cap=limit)

    def semantic_search(self, query, filters, limit):
        q = np.ascontiguousarray(query, dtype=np.float32)
        fp, keep = _filters(filters)
        return self._scored(_lib().cgvs_semantic_search, q.ctypes.data_as(C.c_void_p), q.size, fp, limit, cap=limit)

    def hybrid_search(self, query, filters, vector_weight, limit):
        q = np.ascontiguousarray(query, dtype=np.float32)
        fp, keep = _filters(filters)
        return self._scored(_lib().cgvs_hybrid_search, q.ctypes.data_as(C.c_void_p), q.size, fp, C.c_float(vector_weight),
                            limit, cap=limit)

    def multi_vector_search(self, queries, mode, filters, limit):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        fp, keep = _filters(filters)
        nq, dim = (q.shape if q.ndim == 2 else (0, 0))
        return self._scored(_lib().cgvs_multi_vector_search, q.ctypes.data_as(C.c_void_p), nq, dim, mode, fp, limit, cap=limit)


def embedding_column_for_dimension(dim):
    return _lib().cgvs_embedding_column_for_dimension(dim).decode()


def normalize_surreal_node_id(raw):
    buf = C.create_string_buffer(256)
    _check(_lib().cgvs_normalize_node_id(raw.encode(), buf, 256))
    return buf.value.decode()


def parse_node_id(text):
    out = C.create_string_buffer(16)
    _check(_lib().cgvs_parse_node_id(text.encode(), out))
    return uuid.UUID(bytes=out.raw)


def format_node_id(node_id):
    out = C.create_string_buffer(37)
    _check(_lib().cgvs_format_node_id(node_id.bytes, out))
    return out.value.decode()


def simple_hash(text):
    return int(_lib().cgvs_simple_hash(text.encode()))


def hash_embed(text, dim=384):
    out = np.empty(dim, np.float32)
    _check(_lib().cgvs_hash_embed(text.encode(), dim, out.ctypes.data_as(C.c_void_p)))
    return out


def prefetch_k(limit):
    return int(_lib().cgvs_prefetch_k(limit))


def normalize_scores(scores):
    s = np.array(scores, dtype=np.float32, copy=True)
    _lib().cgvs_normalize_scores(s.ctypes.data_as(C.c_void_p), s.size)
    return s


def cosine_similarity(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(_lib().cgvs_cosine_similarity(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.size))


def combine_embeddings(embs):
    e = np.ascontiguousarray(embs, np.float32)
    out = np.empty(e.shape[1], np.float32)
    _check(_lib().cgvs_combine_embeddings(e.ctypes.data_as(C.c_void_p), e.shape[0], e.shape[1], out.ctypes.data_as(C.c_void_p)))
    return out


def trigram_jaccard(a, b):
    """indexer.rs:2901-2932 on the lower-cased names."""
    return float(_lib().cgvs_trigram_jaccard(a.encode(), b.encode()))


def symbol_name_eligible(target, name):
    """The candidate pre-filter of ai_semantic_match_sync (indexer.rs:2804-2821)."""
    return bool(_lib().cgvs_symbol_name_eligible(target.encode(), name.encode()))


class SymbolResolver:
    """Embedding phase of the index-time symbol resolver (indexer.rs:2790-2843) over the HIP kNN:
    add the known symbols once, then match whole batches of unresolved symbols."""

    def __init__(self, dim, dtype="f32s", device=0):   # f32 rows + bf16 shadow: reference-identical AND batched
        self.dim = int(dim)
        h = C.c_void_p()
        cgvec._check(_lib().cgvs_resolver_create(self.dim, cgvec.DTYPES[dtype], int(device), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _lib().cgvs_resolver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(_lib().cgvs_resolver_count(self._h))

    def add_symbols(self, names, embeddings):
        e = np.ascontiguousarray(embeddings, dtype=np.float32)
        if e.ndim != 2 or e.shape[1] != self.dim or e.shape[0] != len(names):
            raise cgvec.CgvError(cgvec.CGV_ERR_DIM_MISMATCH, f"embeddings {e.shape} vs {len(names)} names x {self.dim}")
        arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
        cgvec._check(_lib().cgvs_resolver_add_symbols(self._h, len(names), arr, e.ctypes.data_as(C.c_void_p)))

    def match(self, targets, embeddings, threshold=0.75):
        """-> (index int64[nq] (-1 = unresolved), similarity f32[nq])"""
        e = np.ascontiguousarray(embeddings, dtype=np.float32)
        if e.ndim != 2 or e.shape[1] != self.dim or e.shape[0] != len(targets):
            raise cgvec.CgvError(cgvec.CGV_ERR_DIM_MISMATCH, f"embeddings {e.shape} vs {len(targets)} targets x {self.dim}")
        arr = (C.c_char_p * len(targets))(*[s.encode() for s in targets])
        idx = np.empty(len(targets), dtype=np.int64)
        sc = np.empty(len(targets), dtype=np.float32)
        cgvec._check(_lib().cgvs_resolver_match(self._h, len(targets), arr, e.ctypes.data_as(C.c_void_p),
                                                float(threshold), idx.ctypes.data_as(C.c_void_p),
                                                sc.ctypes.data_as(C.c_void_p)))
        return idx, sc


def rerank_embeddings(query, candidates, device=0):
    """EmbeddingReRanker::rerank (reranker.rs:113-157) on ready embeddings -> (order uint32[n], scores f32[n])."""
    q = np.ascontiguousarray(query, dtype=np.float32).ravel()
    c = np.ascontiguousarray(candidates, dtype=np.float32)
    n = c.shape[0] if c.ndim == 2 else 0
    order = np.empty(n, dtype=np.uint32)
    sc = np.empty(n, dtype=np.float32)
    if n:
        if c.shape[1] != q.size:
            raise cgvec.CgvError(cgvec.CGV_ERR_DIM_MISMATCH, f"candidate dim {c.shape[1]} != {q.size}")
        cgvec._check(_lib().cgvs_rerank_embeddings(int(device), q.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                                   n, q.size, order.ctypes.data_as(C.c_void_p),
                                                   sc.ctypes.data_as(C.c_void_p)))
    return order, sc
