#!/usr/bin/env python3
"""How exact is the coarse score of an fp8 index (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 per instruction)?  Dense coarse scores
(cgv_debug_coarse_scores_dev) against the float64 cosine of the STORED values (cgv_get_row_f32), by pattern and dimension:
the error in units of u = 2^-24 of |q||c| and against the library's eps."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("codegraph-rust_amd")
U = 2.0 ** -24


def probe(d, dtype, rows, q, tag):
    ix = m.HipKnnIndex(d, dtype=dtype)
    ix.add(rows)
    n = rows.shape[0]
    stored = np.stack([ix.get_row(i) for i in range(n)]).astype(np.float64)
    # queries are rounded the same way: store them in a second index and read them back
    iq = m.HipKnnIndex(d, dtype=dtype)
    iq.add(q)
    qs = np.stack([iq.get_row(i) for i in range(q.shape[0])]).astype(np.float64)
    iq.close()
    coarse = ix.debug_coarse_scores(torch.from_numpy(q).cuda()).cpu().numpy().astype(np.float64)[:, :n]
    ix.search(q[:1], 1)
    eps = ix.stats()["last_eps"]
    ix.close()
    ex = (qs @ stored.T) / (np.linalg.norm(qs, axis=1)[:, None] * np.linalg.norm(stored, axis=1)[None, :])
    err = np.abs(coarse - ex)
    i = np.unravel_index(np.argmax(err), err.shape)
    print(f"{dtype:5s} D={d:4d} {tag:28s} max|err|={err.max():.3e} = {err.max() / U:8.1f} u = {err.max() / eps:6.3f} eps  "
          f"(at cos={ex[i]:+.4f}); mean|err|={err.mean() / U:7.2f} u; eps={eps:.3e}", flush=True)


def main():
    rng = np.random.default_rng(1)
    for d in (64, 128, 256, 768):
        n, nq = 8192, 128
        base = rng.standard_normal((nq, d)).astype(np.float32)
        rnd = rng.standard_normal((n, d)).astype(np.float32)
        near = np.concatenate([base + 0.01 * rng.standard_normal((nq, d)).astype(np.float32), rnd[nq:]])
        pos = np.abs(rnd)
        for dtype in ("fp8", "bf16"):
            probe(d, dtype, rnd, base, "random rows, random queries")
            probe(d, dtype, near, base, "near-duplicates (cos ~ 1)")
            probe(d, dtype, pos, np.abs(base), "all-positive components")
            heavy = (rng.standard_normal((n, d)) * np.exp(1.5 * rng.standard_normal((n, d)))).astype(np.float32)
            probe(d, dtype, heavy, (rng.standard_normal((nq, d)) * np.exp(1.5 * rng.standard_normal((nq, d)))).astype(np.float32),
                  "heavy-tailed magnitudes")
            sparse = rnd * (rng.random((n, d)) < 0.1)
            sparse[:, 0] = 1.0
            probe(d, dtype, sparse.astype(np.float32), base, "sparse rows (10 % non-zero)")
            ones = np.ones((n, d), dtype=np.float32)
            ones[:, ::2] = 0.5
            probe(d, dtype, ones, np.ones((nq, d), dtype=np.float32), "constants 1 / 0.5 (exact sums)")


if __name__ == "__main__":
    main()
