#!/usr/bin/env python3
"""Where does the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) drop small products? One K = 64 block: one huge product
H = 448 * 448 and 63 equal small ones t * t (t a power of two: every product and the true sum are exactly representable in f32).
The coarse score times the (f64) norms gives the accumulator back: the share of the 63 small products that arrived, by t^2 / H."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("codegraph-rust_amd")


def main():
    d = 64
    for sign in (1.0, -1.0):
        for j in range(4, -10, -1):
            t = 2.0 ** j
            row = np.full((1, d), t, dtype=np.float32)
            row[0, 0] = 448.0
            rows = np.repeat(row, 256, axis=0)
            q = row.copy()
            q[0, 1:] *= sign            # small products all negative when sign = -1
            ix = m.HipKnnIndex(d, dtype="fp8")
            ix.add(rows)
            stored = ix.get_row(0).astype(np.float64)
            iq = m.HipKnnIndex(d, dtype="fp8")
            iq.add(q)
            qs = iq.get_row(0).astype(np.float64)
            iq.close()
            coarse = float(ix.debug_coarse_scores(torch.from_numpy(q).cuda()).cpu().numpy()[0, 0])
            ix.close()
            acc = coarse * np.linalg.norm(stored) * np.linalg.norm(qs)
            H = stored[0] * qs[0]
            small = float(np.sum(stored[1:] * qs[1:]))
            got = acc - H
            print(f"sign {sign:+.0f} t=2^{j:+d} small/H = 2^{np.log2(abs(stored[1] * qs[1]) / H):6.1f}  sum of 63 small = {small:.6g}  arrived = {got:.6g} "
                  f"({got / small if small else float('nan'):.4f})  rel err of acc = {abs(acc - H - small) / (H + abs(small)):.3e}", flush=True)


if __name__ == "__main__":
    main()
