"""Rust's `std::collections::hash_map::DefaultHasher` restated in Python, and on top of it the input generator of
the reference's one fixture with search semantics:
/root/reference/crates/codegraph-vector/tests/model_optimization_tests.rs:36-58 (`generate_optimization_vectors`).

DefaultHasher::new() is SipHash-1-3 (one compression round per 8-byte word, three finalisation rounds) with the
keys k0 = k1 = 0; `u64::hash` / `usize::hash` (64-bit target) feed the value's 8 little-endian bytes, so hashing
`seed` then `i` is SipHash-1-3 of the 16-byte concatenation. `sip_hash(c, d, ...)` is the generic SipHash-c-d: the
unit tests pin it on the published SipHash-2-4 vectors (Aumasson & Bernstein, appendix A) and on the first vector of
the SipHash-1-3 table, then use c = 1, d = 3.

Test infrastructure only (tests/, fixture generation); nothing in the product imports it.
"""
import struct

import numpy as np

_M = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & _M


def sip_hash(c, d, k0, k1, data):
    v0 = k0 ^ 0x736F6D6570736575
    v1 = k1 ^ 0x646F72616E646F6D
    v2 = k0 ^ 0x6C7967656E657261
    v3 = k1 ^ 0x7465646279746573

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & _M
        v1 = _rotl(v1, 13) ^ v0
        v0 = _rotl(v0, 32)
        v2 = (v2 + v3) & _M
        v3 = _rotl(v3, 16) ^ v2
        v0 = (v0 + v3) & _M
        v3 = _rotl(v3, 21) ^ v0
        v2 = (v2 + v1) & _M
        v1 = _rotl(v1, 17) ^ v2
        v2 = _rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(data)
    for off in range(0, n - n % 8, 8):
        (m,) = struct.unpack_from("<Q", data, off)
        v3 ^= m
        for _ in range(c):
            v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    tail = data[n - n % 8:]
    b = (n & 0xFF) << 56
    for i, byte in enumerate(tail):
        b |= byte << (8 * i)
    v3 ^= b
    for _ in range(c):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= b
    v2 ^= 0xFF
    for _ in range(d):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


def default_hasher_u64s(*words):
    """DefaultHasher::new(); w.hash(&mut h) for each 64-bit word; h.finish()."""
    return sip_hash(1, 3, 0, 0, b"".join(struct.pack("<Q", w & _M) for w in words))


def u64_as_f32(x):
    """Rust `x as f32` for a u64: round to nearest, ties to even."""
    if x == 0:
        return np.float32(0.0)
    bits = x.bit_length()
    if bits <= 24:
        return np.float32(x)
    shift = bits - 24
    mant, rem, half = x >> shift, x & ((1 << shift) - 1), 1 << (shift - 1)
    if rem > half or (rem == half and (mant & 1)):
        mant += 1
    return np.float32(float(mant) * 2.0 ** shift)   # mant <= 2^24, exact in double and in f32


def generate_optimization_vectors(count, dimension, seed):
    """model_optimization_tests.rs:36-58, f32 arithmetic step by step (one rounding per operation)."""
    out = np.empty((count, dimension), dtype=np.float32)
    denom = np.float32(2.0 ** 64)          # u64::MAX as f32 rounds up to 2^64
    half, two = np.float32(0.5), np.float32(2.0)
    for i in range(count):
        h = default_hasher_u64s(seed, i)
        for j in range(dimension):
            v = u64_as_f32(default_hasher_u64s(h, j))
            out[i, j] = (np.float32(v / denom) - half) * two
    return out
