"""GPU diagnostic: dense MFMA coarse scores vs fp64 reference, with an error-structure
summary (which tile rows/cols are wrong) — used to debug the LDS swizzle / MFMA layout."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("codegraph-rust_amd")


def run(n, nq, d, dtype):
    rng = np.random.default_rng(1)
    rows = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, metric="cosine", dtype=dtype)
    ix.add(rows)
    got = ix.debug_coarse_scores(torch.from_numpy(queries).cuda()).cpu().numpy()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    r = torch.from_numpy(rows).to(tdt).double()
    q = torch.from_numpy(queries).to(tdt).double()
    ref = ((q / q.norm(dim=1, keepdim=True)) @ (r / r.norm(dim=1, keepdim=True)).T).numpy()
    err = np.abs(got - ref)
    nan = np.isnan(got)
    bad = (err > 2e-4) | nan
    print(f"[diag] n={n} nq={nq} d={d} {dtype}: max_err={np.nanmax(err):.3e} nan={nan.sum()} bad={bad.sum()}/{bad.size}")
    if bad.any():
        qi, ri = np.nonzero(bad)
        print("  bad query idx mod 32 hist:", np.bincount(qi % 32, minlength=32).tolist())
        print("  bad row   idx mod 32 hist:", np.bincount(ri % 32, minlength=32).tolist())
        print("  bad query idx //32 hist:", np.bincount(qi // 32).tolist())
        print("  bad row   idx //32 hist:", np.bincount(ri // 32).tolist())
        print("  first bad:", [(int(a), int(b), float(got[a, b]), float(ref[a, b])) for a, b in list(zip(qi, ri))[:8]])
        # is it a permutation? check whether got[q, r] matches ref[q', r'] for simple swaps
        if got.shape[0] == got.shape[1]:
            print("  transposed match:", float(np.nanmax(np.abs(got - ref.T))))
    ix.close()
    return not bad.any()


if __name__ == "__main__":
    ok = True
    for (n, nq, d, dt) in [(256, 256, 64, "bf16"), (700, 300, 768, "bf16"), (513, 1, 128, "fp16"), (5000, 600, 384, "bf16")]:
        ok &= run(n, nq, d, dt)
    print("[diag] ALL OK" if ok else "[diag] FAILURES")
