"""ctypes binding of libcgvec_hip.so (C ABI in include/cgvec.h).

Plumbing only: device memory comes from PyTorch tensors (or the library's own staging
for host arrays); all compute is in the HIP library. There is no CPU fallback — if the
shared library is missing or no MI355X is visible, construction fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcgvec_hip.so")
ABLATE_LIB_PATH = os.path.join(_HERE, "lib", "libcgvec_hip_ablate.so")   # `make ABLATE=1`: + ablations, A/B knobs, traces
if os.environ.get("CGV_LIB_PATH"):      # measurement scripts load the ablate flavour instead of the production library
    LIB_PATH = os.path.abspath(os.environ["CGV_LIB_PATH"])

METRICS = {"cosine": 0, "dot": 1, "cosine_seq": 2, "cosine_scalar": 3}
DTYPES = {"f32": 0, "bf16": 1, "fp16": 2, "fp8": 3, "f32s": 4}

CGV_OK = 0
CGV_ERR_INVALID_ARG, CGV_ERR_DIM_MISMATCH, CGV_ERR_HIP, CGV_ERR_OOM = 1, 2, 3, 4
CGV_ERR_NONFINITE, CGV_ERR_OUT_OF_RANGE, CGV_ERR_INTERNAL, CGV_ERR_IO, CGV_ERR_BUSY = 5, 6, 7, 8, 9
CGV_MAX_K, CGV_FAST_MAX_K, CGV_SHARD_CHUNK_ROWS = 2048, 228, 4096
EXCHANGE = {0: "none", 1: "rccl", 2: "copy"}
PAD_IDX = np.uint64(2**64 - 1)


class CgvError(RuntimeError):
    """Mirrors CodeGraphError::Vector(String) (crates/codegraph-core/src/error.rs:17-18)."""

    def __init__(self, code, msg):
        super().__init__(f"cgvec status {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("n_rows", C.c_uint64), ("device_bytes", C.c_uint64), ("searches", C.c_uint64),
        ("queries", C.c_uint64), ("fallback_queries", C.c_uint64), ("overflow_queries", C.c_uint64),
        ("last_eps", C.c_float), ("max_observed_err", C.c_float), ("last_coarse_ms", C.c_float),
        ("last_total_ms", C.c_float), ("coarse_rows", C.c_uint64), ("last_kprime", C.c_uint32),
        ("last_path", C.c_uint32),
    ]


class ShardedStats(C.Structure):
    _fields_ = [
        ("n_rows", C.c_uint64), ("device_bytes", C.c_uint64), ("searches", C.c_uint64), ("queries", C.c_uint64),
        ("fallback_queries", C.c_uint64), ("n_shards", C.c_uint32), ("exchange", C.c_uint32),
        ("last_search_ms", C.c_float), ("last_exchange_ms", C.c_float),
    ]


def build_library(force=False, ablate=False):
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU). ablate=True: the measurement flavour
    (libcgvec_hip_ablate.so, csrc/Makefile) beside the production library."""
    src_dir = os.path.join(_HERE, "csrc")
    jobs = "-j%d" % max(1, min(9, os.cpu_count() or 1))   # one object per translation unit: they build in parallel
    subprocess.check_call(["make", "-C", src_dir, "-s", jobs] + (["ABLATE=1"] if ablate else []) + (["-B"] if force else []))
    return ABLATE_LIB_PATH if ablate else LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not built. Run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for this library.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.cgv_version.restype = u32
    L.cgv_last_error.restype = C.c_char_p
    L.cgv_device_count.restype = i32
    L.cgv_create.argtypes = [u32, i32, i32, i32, C.POINTER(vp)]
    L.cgv_destroy.argtypes = [vp]
    L.cgv_reserve.argtypes = [vp, u64]
    L.cgv_add_f32.argtypes = [vp, vp, u64]
    L.cgv_add_f32_dev.argtypes = [vp, vp, u64]
    L.cgv_add_f64.argtypes = [vp, vp, u64]
    L.cgv_load_mmap.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    L.cgv_write_mmap_f32.argtypes = [C.c_char_p, vp, u64, u32]
    L.cgv_save_mmap.argtypes = [vp, C.c_char_p]
    L.cgv_count.argtypes = [vp]
    L.cgv_count.restype = u64
    L.cgv_dim.argtypes = [vp]
    L.cgv_dim.restype = u32
    L.cgv_set_index_base.argtypes = [vp, u64]
    L.cgv_update_row_f32.argtypes = [vp, u64, vp]
    L.cgv_search_f32.argtypes = [vp, vp, u32, u32, vp, vp]
    L.cgv_search_f32_dev.argtypes = [vp, vp, u32, u32, vp, vp]
    L.cgv_search_begin_f32_dev.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u64)]
    L.cgv_search_end.argtypes = [vp, u64]
    L.cgv_max_batches_in_flight.argtypes = [vp]
    L.cgv_max_batches_in_flight.restype = u32
    L.cgv_get_row_f32.argtypes = [vp, u64, vp]
    L.cgv_batch_similarity_f32.argtypes = [vp, vp, i32, u64, vp]
    L.cgv_search_baseline_f32.argtypes = [vp, vp, u32, vp, vp, C.POINTER(u32)]
    L.cgv_normalize_rows_f32.argtypes = [i32, vp, u64, u32]
    L.cgv_normalize_rows_scalar_f32.argtypes = [i32, vp, u64, u32]
    L.cgv_synth_rows_f32_dev.argtypes = [i32, u64, u64, u64, u32, i32, vp, vp]
    L.cgv_merge_topk_dev.argtypes = [i32, vp, vp, u32, u32, u32, vp, vp, vp]
    L.cgv_host_device_alias.argtypes = [i32, vp, C.c_size_t]
    L.cgv_host_device_alias.restype = vp
    L.cgv_packed_width.argtypes = [u32]
    L.cgv_packed_width.restype = u32
    L.cgv_pack_topk_dev.argtypes = [i32, vp, vp, u32, u32, vp, vp]
    L.cgv_merge_packed_dev.argtypes = [i32, vp, u32, u32, u32, vp, vp, vp]
    L.cgv_merge_packed_flag_dev.argtypes = [i32, vp, u32, u32, u32, vp, vp, vp, vp]
    L.cgv_merge_packed_flag_dev.restype = i32
    L.cgv_search_packed_begin_f32_dev.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u64)]
    L.cgv_search_packed_begin_f32_dev.restype = i32
    L.cgv_search_packed_end.argtypes = [vp, u64, C.POINTER(i32)]
    L.cgv_search_packed_end.restype = i32
    L.cgv_set_stream.argtypes = [vp, vp]
    L.cgv_use_own_stream.argtypes = [vp]
    L.cgv_synchronize.argtypes = [vp]
    L.cgv_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.cgv_set_profiling.argtypes = [vp, i32]
    L.cgv_set_force_exact.argtypes = [vp, i32]
    L.cgv_debug_coarse_scores_dev.argtypes = [vp, vp, u32, vp]
    L.cgv_set_id_map.argtypes = [vp, u32, u32, u32]
    L.cgv_truncate.argtypes = [vp, u64]
    L.cgv_score_ids_f32.argtypes = [vp, vp, u32, i32, vp, u32, vp]
    L.cgv_sharded_create.argtypes = [u32, i32, i32, u32, C.POINTER(i32), C.POINTER(vp)]
    L.cgv_sharded_destroy.argtypes = [vp]
    L.cgv_sharded_reserve.argtypes = [vp, u64]
    L.cgv_sharded_add_f32.argtypes = [vp, vp, u64]
    L.cgv_sharded_update_row_f32.argtypes = [vp, u64, vp]
    L.cgv_sharded_get_row_f32.argtypes = [vp, u64, vp]
    L.cgv_sharded_count.argtypes = [vp]
    L.cgv_sharded_count.restype = u64
    L.cgv_sharded_n_shards.argtypes = [vp]
    L.cgv_sharded_n_shards.restype = u32
    L.cgv_sharded_shard.argtypes = [vp, u32]
    L.cgv_sharded_shard.restype = vp
    L.cgv_sharded_search_f32.argtypes = [vp, vp, u32, u32, vp, vp]
    L.cgv_sharded_search_begin_f32.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u64)]
    L.cgv_sharded_search_end.argtypes = [vp, u64]
    L.cgv_sharded_max_batches_in_flight.argtypes = [vp]
    L.cgv_sharded_max_batches_in_flight.restype = u32
    L.cgv_sharded_exchange.argtypes = [vp]
    L.cgv_sharded_set_exchange.argtypes = [vp, i32]
    L.cgv_sharded_force_exchange.argtypes = [vp, i32]
    L.cgv_sharded_force_exchange.restype = i32
    L.cgv_set_spin_us.argtypes = [vp, u32]
    L.cgv_set_spin_us.restype = i32
    L.cgv_set_coalesce.argtypes = [vp, u32, u32, u32]
    L.cgv_set_coalesce.restype = i32
    L.cgv_get_coalesce_stats.argtypes = [vp, C.POINTER(u64)]
    L.cgv_get_coalesce_stats.restype = i32
    L.cgv_get_small_batch_stats.argtypes = [vp, C.POINTER(u64)]
    L.cgv_get_phase_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.cgv_get_phase_times.restype = i32
    L.cgv_get_small_batch_stats.restype = i32
    L.cgv_get_sample_repair_stats.argtypes = [vp, C.POINTER(u64)]
    L.cgv_get_sample_repair_stats.restype = i32
    L.cgv_alloc_pinned.argtypes = [C.c_size_t]
    L.cgv_alloc_pinned.restype = vp
    L.cgv_free_pinned.argtypes = [vp]
    L.cgv_free_pinned.restype = i32
    L.cgv_debug_rccl_lib_.argtypes = [C.c_char_p]
    L.cgv_debug_rccl_lib_.restype = i32
    L.cgv_sharded_get_stats.argtypes = [vp, C.POINTER(ShardedStats)]
    for name in ("cgv_set_id_map", "cgv_truncate", "cgv_score_ids_f32", "cgv_sharded_create", "cgv_sharded_destroy",
                 "cgv_sharded_reserve", "cgv_sharded_add_f32", "cgv_sharded_update_row_f32", "cgv_sharded_get_row_f32",
                 "cgv_sharded_search_f32", "cgv_sharded_search_begin_f32", "cgv_sharded_search_end",
                 "cgv_sharded_exchange", "cgv_sharded_set_exchange", "cgv_sharded_get_stats"):
        getattr(L, name).restype = i32
    for name in ("cgv_pack_topk_dev", "cgv_merge_packed_dev", "cgv_add_f64", "cgv_load_mmap", "cgv_write_mmap_f32", "cgv_save_mmap", "cgv_create", "cgv_destroy", "cgv_reserve", "cgv_add_f32", "cgv_add_f32_dev",
                 "cgv_set_index_base", "cgv_update_row_f32", "cgv_search_f32", "cgv_search_f32_dev", "cgv_search_begin_f32_dev", "cgv_search_end", "cgv_get_row_f32",
                 "cgv_merge_topk_dev", "cgv_batch_similarity_f32", "cgv_search_baseline_f32", "cgv_normalize_rows_f32", "cgv_normalize_rows_scalar_f32", "cgv_synth_rows_f32_dev", "cgv_set_stream", "cgv_use_own_stream", "cgv_synchronize", "cgv_get_stats",
                 "cgv_set_profiling", "cgv_set_force_exact", "cgv_debug_coarse_scores_dev"):
        getattr(L, name).restype = i32
    _lib = L
    return L


def _check(rc):
    if rc != CGV_OK:
        raise CgvError(rc, lib().cgv_last_error().decode("utf-8", "replace"))


def device_count():
    return int(lib().cgv_device_count())


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class PendingSearch:
    """A batch in flight; keeps the query / output tensors alive until wait()."""

    def __init__(self, index, ticket, q, idx, sc):
        self._ix, self._t, self._q, self._idx, self._sc = index, ticket, q, idx, sc

    def wait(self):
        if self._t:
            t, self._t = self._t, 0
            _check(lib().cgv_search_end(self._ix._h, C.c_uint64(t)))
        self._q = None
        return self._idx, self._sc

    def __del__(self):
        try:
            if self._t and self._ix._h:
                lib().cgv_search_end(self._ix._h, C.c_uint64(self._t))
        except Exception:
            pass


class HipKnnIndex:
    """One device-resident corpus shard + batched kNN over it.

    Host mirror of what the reference's `SurrealVectorBackend` / `VectorStore` seam needs from
    a backend (crates/codegraph-vector/src/surreal_store.rs:11-22; crates/codegraph-core/src/
    traits.rs:11-16): add rows (store_embeddings/upsert_nodes), search (search_similar/
    vector_knn), get_row (get_embedding/get_node_embedding). Row id = insertion index.
    """

    def __init__(self, dim, metric="cosine", dtype="bf16", device=0):
        self._h = C.c_void_p()
        self.dim, self.metric, self.dtype, self.device = int(dim), metric, dtype, int(device)
        _check(lib().cgv_create(self.dim, METRICS[metric], DTYPES[dtype], self.device, C.byref(self._h)))

    # -- lifecycle ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().cgv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().cgv_count(self._h))

    def reserve(self, n):
        _check(lib().cgv_reserve(self._h, int(n)))

    def set_index_base(self, base):
        _check(lib().cgv_set_index_base(self._h, int(base)))

    def set_stream(self, stream_ptr):
        """Use an external hipStream_t; 0/None is HIP's legacy default stream (PyTorch's default)."""
        _check(lib().cgv_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def use_own_stream(self):
        _check(lib().cgv_use_own_stream(self._h))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_profiling(self, on=True):
        """False / 0: off; True / 1: events around the dominant coarse launch; 2: also around the whole pipeline."""
        _check(lib().cgv_set_profiling(self._h, int(on)))

    def set_force_exact(self, on=True):
        _check(lib().cgv_set_force_exact(self._h, 1 if on else 0))

    def set_spin_us(self, us):
        """How long the end of a search polls its stream before it blocks (cgv_set_spin_us; default 3000, 0 = block at once)."""
        _check(lib().cgv_set_spin_us(self._h, int(us)))

    def set_coalesce(self, max_batch_queries=64, max_batches_in_flight=1, window_us=250):
        """Group commit of concurrent small search calls (cgv_set_coalesce; the defaults are the library's:
        CGV_COALESCE_* in include/cgvec.h); max_batch_queries = 0 switches it off."""
        _check(lib().cgv_set_coalesce(self._h, int(max_batch_queries), int(max_batches_in_flight), int(window_us)))

    def coalesce_stats(self):
        out = (C.c_uint64 * 8)()
        _check(lib().cgv_get_coalesce_stats(self._h, out))
        names = ("batches", "batched_requests", "batched_queries", "lone_calls", "retried_alone", "max_batch_queries", "window_waits")
        return {n: int(out[i]) for i, n in enumerate(names)}

    def phase_times_us(self):
        """Profiling level 3: device microseconds of the last finished search's phases (cgv_get_phase_times)."""
        out = (C.c_float * 4)()
        _check(lib().cgv_get_phase_times(self._h, out))
        return dict(zip(("prep", "first_threshold", "emitting", "final_publish"), (round(float(x), 2) for x in out)))

    def small_batch_stats(self):
        out = (C.c_uint64 * 4)()
        _check(lib().cgv_get_small_batch_stats(self._h, out))
        return {"searches": int(out[0]), "failed_queries": int(out[1]), "repaired_by_cell_rescan": int(out[2]), "exact_scans": int(out[3])}

    def sample_repairs(self):
        """Queries of large batches whose check failed on one cell of the emitting sample and that the final kernel put right by
        itself (cgv_get_sample_repair_stats); not part of stats()['fallback_queries']."""
        out = (C.c_uint64 * 1)()
        _check(lib().cgv_get_sample_repair_stats(self._h, out))
        return int(out[0])

    def synchronize(self):
        _check(lib().cgv_synchronize(self._h))

    def stats(self):
        s = Stats()
        _check(lib().cgv_get_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    # -- data --------------------------------------------------------------------
    def add(self, rows):
        if _is_torch(rows):
            import torch
            if rows.dim() != 2 or rows.shape[1] != self.dim:
                raise CgvError(CGV_ERR_DIM_MISMATCH, f"expected [n,{self.dim}] rows, got {tuple(rows.shape)}")
            if not rows.is_cuda:
                return self.add(rows.detach().float().numpy())
            r = rows.detach().to(torch.float32).contiguous()
            self.use_torch_stream()
            _check(lib().cgv_add_f32_dev(self._h, C.c_void_p(r.data_ptr()), r.shape[0]))
            return
        r = np.ascontiguousarray(rows, dtype=np.float32)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"expected [n,{self.dim}] rows, got {r.shape}")
        _check(lib().cgv_add_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def add_f64(self, rows):
        """Rows as float64 (the SurrealDB embedding_<dim> column type), narrowed `as f32` on device."""
        r = np.ascontiguousarray(rows, dtype=np.float64)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"rows shape {r.shape} != (*, {self.dim})")
        _check(lib().cgv_add_f64(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def load_mmap(self, path):
        """Append the rows of a corpus file in the reference's mmap format (memory.rs:310-374)."""
        n = C.c_uint64(0)
        _check(lib().cgv_load_mmap(self._h, os.fsencode(path), C.byref(n)))
        return int(n.value)

    def save_mmap(self, path):
        _check(lib().cgv_save_mmap(self._h, os.fsencode(path)))

    def update_row(self, i, row):
        r = np.ascontiguousarray(row, dtype=np.float32)
        if r.size != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"row dim {r.size} != {self.dim}")
        _check(lib().cgv_update_row_f32(self._h, int(i), r.ctypes.data_as(C.c_void_p)))

    def get_row(self, i):
        out = np.empty(self.dim, dtype=np.float32)
        _check(lib().cgv_get_row_f32(self._h, int(i), out.ctypes.data_as(C.c_void_p)))
        return out

    @property
    def max_in_flight(self):
        """How many search_begin() results one thread may hold before it must wait() on one."""
        return int(lib().cgv_max_batches_in_flight(self._h))

    def search_begin(self, queries, k):
        """Enqueue one batch (CUDA tensor [nq, dim]) and return a PendingSearch; .wait() gives
        (idx, score) CUDA tensors. Up to the handle's context-pool depth batches overlap on the
        device (cgv_search_begin_f32_dev / cgv_search_end)."""
        import torch
        k = int(k)
        if not (_is_torch(queries) and queries.is_cuda):
            raise CgvError(CGV_ERR_INVALID_ARG, "search_begin takes a CUDA tensor")
        if queries.dim() != 2 or queries.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {tuple(queries.shape)} != {self.dim}")
        q = queries.detach().to(torch.float32).contiguous()
        nq = q.shape[0]
        self.use_torch_stream()  # order after the producer of `queries`
        idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        sc = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        t = C.c_uint64(0)
        if nq and k:
            _check(lib().cgv_search_begin_f32_dev(self._h, C.c_void_p(q.data_ptr()), nq, k,
                                                  C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr()), C.byref(t)))
        return PendingSearch(self, t.value, q, idx, sc)

    def search_begin_pinned(self, q_pinned, k, out):
        """Host-in / host-out batch in flight: `q_pinned` (pinned CPU f32 [nq, dim]) is read in place over PCIe by the
        conversion kernel, the last kernel writes ids / scores straight into the pinned CPU tensors out = (int64 [nq, k],
        float32 [nq, k]); valid after .wait(). The same work as cgv_search_f32 on pinned buffers, split so that up to
        max_in_flight batches overlap on the device (cgv_search_begin_f32_dev on the buffers' device aliases / cgv_search_end).
        The tensors must stay alive and untouched until wait()."""
        import torch
        k = int(k)
        oi, os_ = out
        if not (_is_torch(q_pinned) and not q_pinned.is_cuda and q_pinned.is_pinned() and q_pinned.dtype == torch.float32 and
                q_pinned.is_contiguous() and q_pinned.dim() == 2 and q_pinned.shape[1] == self.dim):
            raise CgvError(CGV_ERR_INVALID_ARG, f"search_begin_pinned takes a contiguous pinned CPU f32 tensor [nq, {self.dim}]")
        nq = q_pinned.shape[0]
        for t, dt in ((oi, torch.int64), (os_, torch.float32)):
            if tuple(t.shape) != (nq, k) or t.dtype != dt or not t.is_contiguous() or t.is_cuda or not t.is_pinned():
                raise CgvError(CGV_ERR_INVALID_ARG, "search_begin_pinned: out = contiguous pinned CPU [nq, k] int64 / float32 tensors")
        self.use_own_stream()   # nothing of the caller's to order after: host memory in, host memory out
        t = C.c_uint64(0)
        if nq and k:
            _check(lib().cgv_search_begin_f32_dev(self._h, C.c_void_p(self.device_alias(q_pinned)), nq, k,
                                                  C.c_void_p(self.device_alias(oi)), C.c_void_p(self.device_alias(os_)), C.byref(t)))
        return PendingSearch(self, t.value, q_pinned, oi, os_)

    def search(self, queries, k):
        """queries [nq, dim] -> (idx uint64 [nq,k], score f32 [nq,k]); numpy in -> numpy out,
        CUDA tensor in -> CUDA tensors out (idx as int64 view of the uint64 ids)."""
        k = int(k)
        if _is_torch(queries) and queries.is_cuda:
            import torch
            if queries.dim() != 2 or queries.shape[1] != self.dim:
                raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {tuple(queries.shape)} != {self.dim}")
            q = queries.detach().to(torch.float32).contiguous()
            nq = q.shape[0]
            self.use_torch_stream()  # same stream as the producer of `queries`
            idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            sc = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            if nq and k:
                _check(lib().cgv_search_f32_dev(self._h, C.c_void_p(q.data_ptr()), nq, k,
                                                C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr())))
            return idx, sc
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {q.shape} != {self.dim}")
        nq = q.shape[0]
        idx = np.empty((nq, k), dtype=np.uint64)
        sc = np.empty((nq, k), dtype=np.float32)
        if nq and k:
            _check(lib().cgv_search_f32(self._h, q.ctypes.data_as(C.c_void_p), nq, k,
                                        idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        return idx, sc

    def search_from_pinned(self, q_pinned, k):
        """Pinned HOST queries (a torch CPU tensor in pinned memory) -> (idx, score) CUDA tensors: the conversion kernel
        reads the batch over PCIe in place (no separate H2D copy); what a rank of the row-sharded deployment does before
        the exchange of its partial top-k."""
        import torch
        if not (_is_torch(q_pinned) and not q_pinned.is_cuda and q_pinned.is_pinned()):
            raise CgvError(CGV_ERR_INVALID_ARG, "search_from_pinned takes a pinned CPU tensor")
        if q_pinned.dim() != 2 or q_pinned.shape[1] != self.dim or q_pinned.dtype != torch.float32 or not q_pinned.is_contiguous():
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query batch {tuple(q_pinned.shape)} {q_pinned.dtype} != [nq, {self.dim}] f32")
        nq, k = q_pinned.shape[0], int(k)
        dev = torch.device("cuda", self.device)
        self.use_torch_stream()
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        if nq and k:
            # the address the DEVICE reads the pinned batch at (equal to the host address for hipHostMalloc memory, possibly
            # different for hipHostRegister-ed memory): resolved by the library over the whole range
            alias = lib().cgv_host_device_alias(self.device, C.c_void_p(q_pinned.data_ptr()), nq * self.dim * 4)
            if not alias:
                raise CgvError(CGV_ERR_INVALID_ARG, "search_from_pinned: the batch is not (wholly) pinned / registered host memory")
            _check(lib().cgv_search_f32_dev(self._h, C.c_void_p(alias), nq, k,
                                            C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr())))
        return idx, sc

    def device_alias(self, t):
        """Device-visible address of a pinned CPU tensor's storage (cgv_host_device_alias), or the data pointer of a CUDA tensor."""
        if t.is_cuda:
            return t.data_ptr()
        alias = lib().cgv_host_device_alias(self.device, C.c_void_p(t.data_ptr()), t.numel() * t.element_size())
        if not alias:
            raise CgvError(CGV_ERR_INVALID_ARG, "the tensor is not (wholly) pinned / registered host memory")
        return alias

    def search_packed_begin(self, queries, k, rec, stream=None):
        """One rank's share of a batch without a host join before the exchange (cgv_search_packed_begin_f32_dev): the shard
        search is enqueued, its top-k are packed into `rec` (CUDA int32 [nq, packed_width(k)]) on the library's stream, and
        torch's CURRENT stream is made to wait for them - enqueue the all-gather + merge_packed(..., redo=flag) next and
        synchronise once. `queries`: CUDA tensor or pinned CPU tensor [nq, dim] f32. Returns the ticket for
        search_packed_end()."""
        import torch
        nq, k = queries.shape[0], int(k)
        if queries.dim() != 2 or queries.shape[1] != self.dim or queries.dtype != torch.float32 or not queries.is_contiguous():
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query batch {tuple(queries.shape)} {queries.dtype} != [nq, {self.dim}] f32")
        if not rec.is_cuda or rec.dtype != torch.int32 or rec.numel() != nq * packed_width(k) or not rec.is_contiguous():
            raise CgvError(CGV_ERR_INVALID_ARG, "rec must be a contiguous CUDA int32 tensor of nq * packed_width(k) words")
        if stream is None:        # the search orders after what torch queued so far on its CURRENT stream (the producer of a CUDA batch)
            stream = torch.cuda.current_stream(rec.device).cuda_stream
        # (stream given: the raw handle of the stream the batch runs on - the caller has ordered it behind the producer itself and
        #  spares the interpreter the stream context around this call: the first kernel leaves ~10 us earlier, ShardedKnn)
        self.set_stream(stream)
        t = C.c_uint64(0)
        if nq and k:
            _check(lib().cgv_search_packed_begin_f32_dev(self._h, C.c_void_p(self.device_alias(queries)), nq, k,
                                                         C.c_void_p(rec.data_ptr()), C.c_void_p(stream), C.byref(t)))
        return t.value

    def search_packed_end(self, ticket):
        """Second half: the search's status (a NaN query fails here); True when provisional records were replaced by final
        ones (the exact scan ran) - the caller repeats the exchange then."""
        r = C.c_int(0)
        _check(lib().cgv_search_packed_end(self._h, C.c_uint64(int(ticket)), C.byref(r)))
        return bool(r.value)

    def search_host_ptr(self, q_ptr, nq, k, idx_ptr, score_ptr):
        """cgv_search_f32 on raw HOST pointers (any mix of pinned and pageable buffers): the caller owns the memory."""
        _check(lib().cgv_search_f32(self._h, C.c_void_p(int(q_ptr)), int(nq), int(k), C.c_void_p(int(idx_ptr)),
                                    C.c_void_p(int(score_ptr))))

    OPS = {"cosine": 0, "dot": 1, "l2": 2, "cosine_seq": 3, "cosine_distance_seq": 4, "cosine_scalar": 6}

    def batch_similarity(self, query, op="cosine", limit_rows=0):
        """parallel_batch_similarity / compute_distances_cpu: op(query, row i) for the first rows."""
        q = np.ascontiguousarray(query, dtype=np.float32)
        if q.size != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {q.size} != {self.dim}")
        n = min(limit_rows, len(self)) if limit_rows else len(self)
        out = np.empty(n, dtype=np.float32)
        _check(lib().cgv_batch_similarity_f32(self._h, q.ctypes.data_as(C.c_void_p), self.OPS[op], int(limit_rows),
                                              out.ctypes.data_as(C.c_void_p)))
        return out

    def score_ids(self, queries, ids, op="cosine_seq"):
        """op(query q, stored row ids[q][j]) for [nq, m] LOCAL ids in one launch (cgv_score_ids_f32): the
        per-hit re-score of SemanticSearch::search_by_embedding without a host round trip per hit."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        i = np.ascontiguousarray(ids, dtype=np.uint64)
        if q.ndim != 2 or q.shape[1] != self.dim or i.ndim != 2 or i.shape[0] != q.shape[0]:
            raise CgvError(CGV_ERR_DIM_MISMATCH, "queries [nq, dim] and ids [nq, m] expected")
        out = np.zeros(i.shape, dtype=np.float32)
        _check(lib().cgv_score_ids_f32(self._h, q.ctypes.data_as(C.c_void_p), q.shape[0], self.OPS[op],
                                       i.ctypes.data_as(C.c_void_p), i.shape[1], out.ctypes.data_as(C.c_void_p)))
        return out

    def set_id_map(self, chunk_rows, n_shards, shard):
        _check(lib().cgv_set_id_map(self._h, int(chunk_rows), int(n_shards), int(shard)))

    def truncate(self, n):
        _check(lib().cgv_truncate(self._h, int(n)))

    def search_baseline(self, query, limit):
        """ModelOptimizer::search_baseline -> (row ids, distances)."""
        q = np.ascontiguousarray(query, dtype=np.float32)
        idx = np.empty(max(limit, 1), dtype=np.uint64)
        dist = np.empty(max(limit, 1), dtype=np.float32)
        n = C.c_uint32(0)
        _check(lib().cgv_search_baseline_f32(self._h, q.ctypes.data_as(C.c_void_p), int(limit), idx.ctypes.data_as(C.c_void_p),
                                             dist.ctypes.data_as(C.c_void_p), C.byref(n)))
        return idx[:n.value], dist[:n.value]

    def debug_coarse_scores(self, queries):
        """Dense approximate (MFMA) scores [nq, n] as a CUDA tensor — test hook."""
        import torch
        q = queries.detach().to(torch.float32).contiguous()
        out = torch.full((q.shape[0], len(self)), float("nan"), dtype=torch.float32, device=q.device)
        self.use_torch_stream()
        _check(lib().cgv_debug_coarse_scores_dev(self._h, C.c_void_p(q.data_ptr()), q.shape[0],
                                                 C.c_void_p(out.data_ptr())))
        return out


class ShardedIndex:
    """ONE handle over several devices in one process (cgv_sharded_*, csrc/sharded.hip): global row id =
    insertion index, rows dealt block-cyclically to the shards, one exchange of the per-shard top-k
    (RCCL all-gather when the devices are distinct, device copies when a device is listed twice)."""

    def __init__(self, dim, devices, metric="cosine", dtype="bf16"):
        self._h = C.c_void_p()
        self.dim, self.metric, self.dtype = int(dim), metric, dtype
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        _check(lib().cgv_sharded_create(self.dim, METRICS[metric], DTYPES[dtype], len(devices), devs, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().cgv_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().cgv_sharded_count(self._h))

    @property
    def n_shards(self):
        return int(lib().cgv_sharded_n_shards(self._h))

    @property
    def exchange(self):
        return EXCHANGE[int(lib().cgv_sharded_exchange(self._h))]

    def set_exchange(self, kind):
        _check(lib().cgv_sharded_set_exchange(self._h, {"rccl": 1, "copy": 2}[kind]))

    def force_exchange(self, on=True):
        """A handle over ONE shard: run pack -> exchange (one-rank ncclAllGather) -> merge anyway (cgv_sharded_force_exchange)."""
        _check(lib().cgv_sharded_force_exchange(self._h, 1 if on else 0))

    def reserve(self, n):
        _check(lib().cgv_sharded_reserve(self._h, int(n)))

    def add(self, rows):
        if _is_torch(rows):
            rows = rows.detach().float().cpu().numpy()
        r = np.ascontiguousarray(rows, dtype=np.float32)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"expected [n,{self.dim}] rows, got {r.shape}")
        _check(lib().cgv_sharded_add_f32(self._h, r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def update_row(self, i, row):
        r = np.ascontiguousarray(row, dtype=np.float32)
        _check(lib().cgv_sharded_update_row_f32(self._h, int(i), r.ctypes.data_as(C.c_void_p)))

    def get_row(self, i):
        out = np.empty(self.dim, dtype=np.float32)
        _check(lib().cgv_sharded_get_row_f32(self._h, int(i), out.ctypes.data_as(C.c_void_p)))
        return out

    def shard_counts(self):
        L = lib()
        return [int(L.cgv_count(C.c_void_p(L.cgv_sharded_shard(self._h, i)))) for i in range(self.n_shards)]

    def set_force_exact(self, flag):
        """Every shard answers by its exact full scan (cgv_set_force_exact on the shard handles): the anchor of the
        full-size checks - the fast path's merged answer must equal the merged exact scans, bit for bit."""
        L = lib()
        for i in range(self.n_shards):
            _check(L.cgv_set_force_exact(C.c_void_p(L.cgv_sharded_shard(self._h, i)), 1 if flag else 0))

    def shard_handle(self, i):
        return C.c_void_p(lib().cgv_sharded_shard(self._h, i))

    def shard_stats(self, i):
        s = Stats()
        _check(lib().cgv_get_stats(C.c_void_p(lib().cgv_sharded_shard(self._h, i)), C.byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def stats(self):
        s = ShardedStats()
        _check(lib().cgv_sharded_get_stats(self._h, C.byref(s)))
        d = {f: getattr(s, f) for f, _ in ShardedStats._fields_}
        d["exchange"] = EXCHANGE[d["exchange"]]
        return d

    def search(self, queries, k):
        if _is_torch(queries):
            queries = queries.detach().float().cpu().numpy()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {q.shape} != {self.dim}")
        nq, k = q.shape[0], int(k)
        idx = np.empty((nq, k), dtype=np.uint64)
        sc = np.empty((nq, k), dtype=np.float32)
        if nq and k:
            _check(lib().cgv_sharded_search_f32(self._h, q.ctypes.data_as(C.c_void_p), nq, k,
                                                idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        return idx, sc


    @property
    def max_in_flight(self):
        return int(lib().cgv_sharded_max_batches_in_flight(self._h))

    def search_begin(self, queries, k):
        """First half of a batch (cgv_sharded_search_begin_f32): returns a handle whose .wait() is the second half
        and yields (ids, scores). The query buffer is copied before the call returns."""
        q = np.ascontiguousarray(queries.detach().float().cpu().numpy() if _is_torch(queries) else queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise CgvError(CGV_ERR_DIM_MISMATCH, f"query dim {q.shape} != {self.dim}")
        nq, k = q.shape[0], int(k)
        idx = np.empty((nq, k), dtype=np.uint64)
        sc = np.empty((nq, k), dtype=np.float32)
        t = C.c_uint64(0)
        _check(lib().cgv_sharded_search_begin_f32(self._h, q.ctypes.data_as(C.c_void_p), nq, k, idx.ctypes.data_as(C.c_void_p),
                                                  sc.ctypes.data_as(C.c_void_p), C.byref(t)))
        owner = self

        class _Pending:
            def wait(self_inner):
                _check(lib().cgv_sharded_search_end(owner._h, t.value))
                return idx, sc
        return _Pending()


def normalize_rows(rows, device=0, arm="avx2"):
    """parallel_normalize_vectors on device; returns a normalised copy. arm = "avx2" (simd_ops.rs:189-222, what an
    AVX2 + FMA host runs) or "scalar" (simd_ops.rs:394-415, every other host)."""
    r = np.array(rows, dtype=np.float32, copy=True, order="C")
    if r.ndim == 1:
        r = r[None, :]
    fn = lib().cgv_normalize_rows_f32 if arm == "avx2" else lib().cgv_normalize_rows_scalar_f32
    _check(fn(device, r.ctypes.data_as(C.c_void_p), r.shape[0], r.shape[1]))
    return r


def synth_rows_dev(seed, row0, nrows, dim, normalise=True, device=0, out=None):
    """rows [row0, row0 + nrows) of the counter-based stream `seed` as a float32 DEVICE tensor
    (cgv_synth_rows_f32_dev on torch's current stream); `out` = an existing [nrows, dim] float32 cuda tensor to fill."""
    import torch
    if out is None:
        out = torch.empty((nrows, dim), dtype=torch.float32, device=f"cuda:{device}")
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (nrows, dim)
    _check(lib().cgv_synth_rows_f32_dev(device, seed, row0, nrows, dim, 1 if normalise else 0, out.data_ptr(),
                                        torch.cuda.current_stream(out.device).cuda_stream))
    return out


def write_mmap(path, rows):
    """MemoryOptimizer::save_to_mmap (memory.rs:242-307) for host rows."""
    r = np.ascontiguousarray(rows, dtype=np.float32)
    if r.ndim != 2:
        raise CgvError(CGV_ERR_INVALID_ARG, "rows must be [n, dim]")
    _check(lib().cgv_write_mmap_f32(os.fsencode(path), r.ctypes.data_as(C.c_void_p), r.shape[0], r.shape[1]))


def merge_topk(idx, score, device=None):
    """[g, nq, k] CUDA tensors (idx int64 view of uint64 ids) -> merged [nq, k]."""
    import torch
    if device is None:
        device = idx.device.index or 0
    g, nq, k = idx.shape
    idx = idx.contiguous()
    score = score.contiguous()
    oi = torch.empty((nq, k), dtype=torch.int64, device=idx.device)
    os_ = torch.empty((nq, k), dtype=torch.float32, device=idx.device)
    stream = torch.cuda.current_stream(idx.device).cuda_stream
    _check(lib().cgv_merge_topk_dev(device, C.c_void_p(idx.data_ptr()), C.c_void_p(score.data_ptr()), g, nq, k,
                                    C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(stream)))
    return oi, os_


def packed_width(k):
    return int(lib().cgv_packed_width(int(k)))


def pack_topk(idx, score):
    """[nq, k] CUDA results -> one int32 record buffer [nq, packed_width(k)] (ids | scores)."""
    import torch
    nq, k = idx.shape
    rec = torch.empty((nq, packed_width(k)), dtype=torch.int32, device=idx.device)
    stream = torch.cuda.current_stream(idx.device).cuda_stream
    _check(lib().cgv_pack_topk_dev(idx.device.index or 0, C.c_void_p(idx.contiguous().data_ptr()),
                                   C.c_void_p(score.contiguous().data_ptr()), nq, k, C.c_void_p(rec.data_ptr()),
                                   C.c_void_p(stream)))
    return rec


def merge_packed(gathered, k, out=None, redo=None):
    """[g, nq, packed_width(k)] int32 (the all-gather output) -> merged ([nq, k] int64 ids, f32 scores).
    out = (ids, scores): contiguous [nq, k] int64 / float32 tensors to fill - CUDA tensors, or PINNED CPU tensors,
    which the merge kernel writes in place (valid once the stream has been synchronised).
    redo: a one-element int32 tensor (CUDA or pinned CPU, zeroed by the caller) the kernel sets to 1 when any rank's list of
    any query is PROVISIONAL (cgv_search_packed_begin_f32_dev) - the exchange of the batch must then be repeated."""
    import torch
    g, nq, _ = gathered.shape
    if out is not None:
        oi, os_ = out
        for t, dt in ((oi, torch.int64), (os_, torch.float32)):
            if tuple(t.shape) != (nq, k) or t.dtype != dt or not t.is_contiguous() or not (t.is_cuda or t.is_pinned()):
                raise CgvError(CGV_ERR_INVALID_ARG, "merge_packed: out tensors must be contiguous [nq, k] int64 / float32, CUDA or pinned")
    else:
        oi = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
        os_ = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
    stream = torch.cuda.current_stream(gathered.device).cuda_stream
    _check(lib().cgv_merge_packed_flag_dev(gathered.device.index or 0, C.c_void_p(gathered.data_ptr()), g, nq, k,
                                           C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()),
                                           C.c_void_p(redo.data_ptr() if redo is not None else None), C.c_void_p(stream)))
    return oi, os_
