"""codegraph-rust_amd — MI355X-native brute-force kNN for CodeGraph's codegraph-vector path.

The product is the C-ABI library `lib/libcgvec_hip.so` (sources in csrc/, header in
include/cgvec.h). This package is the thin Python plumbing used by tests and bench.py.
The directory name is not a Python identifier: load it with
`importlib.import_module("codegraph-rust_amd")`.
"""
from . import cgvec  # noqa: F401
from .cgvec import (CgvError, HipKnnIndex, PendingSearch, ShardedIndex, build_library, device_count, merge_packed, merge_topk,  # noqa: F401
                    pack_topk, write_mmap)
from .sharded import ShardedKnn, shard_range  # noqa: F401
from . import store  # noqa: F401
from .i8scan import Int8ScanIndex, quantize_u4, quantize_u8  # noqa: F401
from .quant import ProductQuantizer, ScalarQuantizer  # noqa: F401
