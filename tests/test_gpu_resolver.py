"""SURVEY.md §8(f)2 on the GPU: |unresolved| x |known| similarity + thresholded argmax behind the
name filter, against the oracle's literal per-symbol loop (indexer.rs:2790-2843)."""
import importlib

import numpy as np
import pytest

from _util import pkg

pytestmark = pytest.mark.gpu

WORDS = ["parse", "file", "node", "graph", "vector", "index", "search", "embed", "hash", "map", "config", "store",
         "query", "result", "token", "chunk", "edge", "symbol", "resolve", "cache"]


def _names(rng, n):
    out = []
    for i in range(n):
        w = rng.choice(WORDS, size=rng.integers(1, 4))
        s = "_".join(w) if i % 3 else "".join(x.capitalize() for x in w)
        out.append(s + (str(i % 7) if i % 5 == 0 else ""))
    return out


@pytest.mark.parametrize("dtype", ["f32", "f32s", "bf16"])
def test_resolver_matches_reference_loop(oracle, dtype):
    pkg()
    st = importlib.import_module("codegraph-rust_amd.store")
    rng = np.random.default_rng(41)
    n, nq, d = 6000, 150, 128
    names = _names(rng, n)
    embs = rng.standard_normal((n, d)).astype(np.float32)
    targets = _names(rng, nq)
    # targets are noisy copies of known symbols (similarity around the 0.75 threshold), some unrelated
    src = rng.integers(0, n, nq)
    noise = rng.uniform(0.3, 1.4, (nq, 1)).astype(np.float32)
    temb = embs[src] + noise * rng.standard_normal((nq, d)).astype(np.float32)
    for q in range(0, nq, 2):
        targets[q] = names[src[q]]                      # same name -> passes the filter
    for q in range(0, nq, 10):
        embs[(src[q] + 1) % n] = embs[src[q]]           # an exact duplicate embedding under another name
    r = st.SymbolResolver(d, dtype=dtype)
    try:
        r.add_symbols(names[:2500], embs[:2500])
        r.add_symbols(names[2500:], embs[2500:])
        assert len(r) == n
        idx, sc = r.match(targets, temb, 0.75)
        odt = {"f32": 0, "f32s": 0, "bf16": 1}[dtype]
        se, te = oracle.round_trip(embs, odt), oracle.round_trip(temb, odt)
        hits = 0
        for q in range(nq):
            ri, rs = oracle.symbol_match_phase2(targets[q], te[q], names, se, 0.75)
            assert idx[q] == ri, (q, targets[q], idx[q], ri)
            if ri >= 0:
                hits += 1
                assert sc[q] == np.float32(rs)
        assert 10 < hits < nq - 10                      # both outcomes exercised
    finally:
        r.close()


def test_resolver_escalates_past_ineligible_candidates(oracle):
    """More than 32 (and more than 256) better-scoring but name-ineligible symbols in front of the
    eligible one: the walk widens to CGV_MAX_K, then to the exact scores of every symbol."""
    pkg()
    st = importlib.import_module("codegraph-rust_amd.store")
    rng = np.random.default_rng(5)
    d = 64
    base = rng.standard_normal(d).astype(np.float32)
    for n_bad in (40, 300):
        names = [f"zzzzzzzz{i:04d}" for i in range(n_bad)] + ["parse_file_impl"] + [f"other{i}" for i in range(500)]
        embs = np.vstack([base + 0.01 * rng.standard_normal((n_bad, d)).astype(np.float32),
                          base + 0.3 * rng.standard_normal((1, d)).astype(np.float32),
                          rng.standard_normal((500, d)).astype(np.float32)])
        r = st.SymbolResolver(d, dtype="f32")
        try:
            r.add_symbols(names, embs)
            idx, sc = r.match(["parse_file"], base[None, :], 0.75)
            ri, rs = oracle.symbol_match_phase2("parse_file", base, names, embs, 0.75)
            assert ri == n_bad and idx[0] == ri and sc[0] == np.float32(rs)
            idx2, _ = r.match(["qqqq"], base[None, :], 0.75)            # nothing eligible at all
            assert idx2[0] == -1 == oracle.symbol_match_phase2("qqqq", base, names, embs, 0.75)[0]
        finally:
            r.close()


def test_rerank_embeddings_matches_reference_order(oracle):
    """reranker.rs:113-157: sequential cosine of the query vs <= ~100 candidate embeddings, stable sort desc."""
    pkg()
    st = importlib.import_module("codegraph-rust_amd.store")
    rng = np.random.default_rng(9)
    q = rng.standard_normal(384).astype(np.float32)
    c = rng.standard_normal((100, 384)).astype(np.float32)
    c[40] = c[7]                       # tie: candidate order is kept
    c[55] = 0.0                        # zero norm -> 0.0 (reranker.rs:104-106)
    order, sc = st.rerank_embeddings(q, c)
    ref = np.array([oracle.search_cosine(q, r) for r in c], np.float32)
    ro = np.argsort(-ref, kind="stable")
    assert np.array_equal(order, ro.astype(np.uint32)) and np.array_equal(sc, ref[ro])
    assert list(order).index(7) + 1 == list(order).index(40)
    o0, s0 = st.rerank_embeddings(q, np.zeros((0, 384), np.float32))
    assert o0.size == 0 and s0.size == 0
