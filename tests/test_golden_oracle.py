"""The oracle must keep reproducing the committed golden vectors (tests/golden/)."""
import os

import numpy as np


def test_oracle_reproduces_golden(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_small.npz"))
    k = int(g["k"])
    for name, dt in (("f32", oracle.F32), ("bf16", oracle.BF16), ("fp16", oracle.FP16)):
        i, s = oracle.batch_top_k(g["queries"], g["rows"], k, metric=oracle.COSINE, dtype=dt)
        assert np.array_equal(i, g[f"idx_{name}"]) and np.array_equal(s, g[f"score_{name}"])
        i, s = oracle.batch_top_k(g["queries"], g["rows"], k, metric=oracle.DOT, dtype=dt)
        assert np.array_equal(i, g[f"dot_idx_{name}"]) and np.array_equal(s, g[f"dot_score_{name}"])
    assert g["kat_parallel_idx"].tolist() == list(range(999, 989, -1))
    assert abs(float(g["kat_cos_1to8"]) - 120.0 / 204.0) < 1e-6
    assert np.array_equal(oracle.hash_embed("node_17", 384), g["hash_embed_node_17"])
    # multi-threaded == single-threaded (the parallel sort must not change results)
    i1, s1 = oracle.batch_top_k(g["queries"], g["rows"], k, threads=1)
    i8, s8 = oracle.batch_top_k(g["queries"], g["rows"], k, threads=8)
    assert np.array_equal(i1, i8) and np.array_equal(s1, s8)
