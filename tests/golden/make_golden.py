"""Generates tests/golden/knn_small.npz with the CPU oracle (oracle/cgv_oracle.cpp).

The reference is Rust and cannot be executed in this environment, so these vectors are
produced by the oracle restatement of its arithmetic AFTER that restatement was pinned
against the reference's own known-answer tests (tests/test_oracle_kats.py). The file is
data only: inputs + expected outputs. Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402


def main():
    rng = np.random.default_rng(0xC0DE)
    n, d, nq, k = 512, 96, 8, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows *= rng.uniform(0.3, 2.0, (n, 1)).astype(np.float32)   # un-normalised rows
    rows[40] = rows[11]                                        # one exact duplicate (tie)
    rows[77] = 0.0                                             # one zero row
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    queries[3] = rows[11]
    out = {"rows": rows, "queries": queries, "k": np.int64(k)}
    for name, dt in (("f32", o.F32), ("bf16", o.BF16), ("fp16", o.FP16)):
        i, s = o.batch_top_k(queries, rows, k, metric=o.COSINE, dtype=dt, threads=1)
        out[f"idx_{name}"], out[f"score_{name}"] = i, s
        i, s = o.batch_top_k(queries, rows, k, metric=o.DOT, dtype=dt, threads=1)
        out[f"dot_idx_{name}"], out[f"dot_score_{name}"] = i, s
    # the reference's own KAT inputs with the derivable answers
    q = np.ones(256, np.float32)
    ramp = (np.arange(1000)[:, None] + np.arange(256)[None, :]).astype(np.float32)
    out["kat_parallel_idx"], out["kat_parallel_score"] = o.parallel_top_k(q, ramp, 10, threads=1)
    out["kat_cos_1to8"] = np.float32(o.cosine_avx2(np.arange(1, 9, dtype=np.float32),
                                                   np.arange(8, 0, -1).astype(np.float32)))
    out["hash_embed_node_17"] = o.hash_embed("node_17", 384)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_small.npz"), **out)
    print("wrote knn_small.npz")


if __name__ == "__main__":
    main()
