"""Regenerates the INPUTS of the reference's one fixture with search semantics and records the oracle's answers.

  /root/reference/crates/codegraph-vector/tests/model_optimization_tests.rs
    :36-58   generate_optimization_vectors(count, dimension, seed)  - Rust DefaultHasher = SipHash-1-3, zero keys
    :347-427 test_end_to_end_optimization_pipeline: vectors = generate(1000, 128, 11223), query = vectors[0],
             int8 `search_optimized(query, 10)` vs `search_baseline(query, &vectors, 10)`,
             positional agreement of the two top-10 lists >= 0.8.

The vectors are bit-for-bit what the Rust test computes (siphash13.py restates the hasher and the f32 steps; its
SipHash is pinned on published vectors in tests/test_oracle_kats.py). The expected index lists are the CPU
oracle's (oracle/cgv_oracle.cpp: optimization.rs:63-150, 212-283, 376-418) - the reference itself cannot run here.
Writes tests/golden/optimization_11223.npz (data only).   python tests/golden/make_optimization_fixture.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import siphash13  # noqa: E402
from oracle import oracle as o  # noqa: E402

COUNT, DIM, SEED, LIMIT = 1000, 128, 11223, 10


def main():
    v = siphash13.generate_optimization_vectors(COUNT, DIM, SEED)
    q = v[0]
    base_idx, base_dist = o.search_baseline(q, v, LIMIT)
    data = o.quantize_u8(v)
    opt_idx = o.search_optimized_u8(q, data, LIMIT)
    agree = sum(int(a == b) for a, b in zip(opt_idx, base_idx)) / float(len(opt_idx))
    np.savez_compressed(os.path.join(HERE, "optimization_11223.npz"), vectors=v, baseline_idx=base_idx,
                        baseline_dist=base_dist, int8_idx=opt_idx, quantized_sha256=np.frombuffer(
                            hashlib.sha256(data.tobytes()).digest(), dtype=np.uint8))
    print("vectors sha256", hashlib.sha256(v.tobytes()).hexdigest())
    print("baseline", base_idx.tolist())
    print("int8    ", opt_idx.tolist())
    print("agreement", agree)


if __name__ == "__main__":
    main()
