"""Host-side mirror (include/cgvec_store.h) against the reference's own unit tests for the
seam and against the oracle — no GPU needed (the mock backend is the reference's MockBackend)."""
import uuid

import numpy as np
import pytest

from _util import pkg


def S():
    return pkg().store


def test_strips_table_prefix_from_ids():
    # crates/codegraph-vector/src/surreal_store.rs:137-142
    assert S().normalize_surreal_node_id("nodes:018f3b7d-a82d-4f40-9127-2db4beefabcd") == "018f3b7d-a82d-4f40-9127-2db4beefabcd"


def test_keeps_clean_ids_intact():
    # surreal_store.rs:144-148
    raw = "018f3b7d-a82d-4f40-9127-2db4beefabcd"
    assert S().normalize_surreal_node_id(raw) == raw


def test_search_similar_uses_surreal_backend():
    # surreal_store.rs:150-165: MockBackend, 2560-d query -> column "embedding_2560", id round-trips
    u = "018f3b7d-a82d-4f40-9127-2db4beefabcd"
    st = S().VectorStore.with_mock_backend([(f"nodes:{u}", 0.42)], ef_search=128)
    res = st.search_similar(np.zeros(2560, np.float32), 3)
    assert len(res) == 1 and str(res[0]) == u
    assert st.recorded_columns() == ["embedding_2560"]
    assert st.search_similar(np.zeros(0, np.float32), 3) == []      # surreal_store.rs:62-64
    assert st.search_similar(np.zeros(384, np.float32), 0) == []
    st.close()


def test_invalid_id_from_backend_is_a_vector_error():
    # surreal_store.rs:75-80
    st = S().VectorStore.with_mock_backend([("nodes:not-a-uuid", 0.1)])
    with pytest.raises(pkg().CgvError) as ei:
        st.search_similar(np.zeros(384, np.float32), 3)
    assert "Invalid node id 'nodes:not-a-uuid' returned by Surreal search" in str(ei.value)
    st.close()


def test_embedding_column_for_dimension():
    # crates/codegraph-graph/src/surrealdb_storage.rs:1932-1952
    for d in (384, 768, 1024, 1536, 2048, 2560, 3072, 4096):
        assert S().embedding_column_for_dimension(d) == f"embedding_{d}"
    assert S().embedding_column_for_dimension(100) == "embedding_2048"   # unsupported -> warn + 2048


def test_uuid_round_trip_and_forms():
    u = uuid.UUID("018f3b7d-a82d-4f40-9127-2db4beefabcd")
    assert S().parse_node_id(str(u)) == u
    assert S().parse_node_id(u.hex) == u
    assert S().parse_node_id("{" + str(u) + "}") == u
    assert S().parse_node_id("urn:uuid:" + str(u)) == u
    assert S().format_node_id(u) == str(u)
    with pytest.raises(pkg().CgvError):
        S().parse_node_id("018f3b7d-a82d-4f40-9127")


def test_hash_embedder_matches_oracle(oracle):
    # search.rs:178-205, 535-541 (and the copies listed in SURVEY.md §8(a13))
    for text in ("sum two numbers", "node_17", "", "fn add(x: i32, y: i32) -> i32 { x + y }"):
        assert S().simple_hash(text) == oracle.simple_hash(text)
        for dim in (8, 384, 768):
            assert np.array_equal(S().hash_embed(text, dim), oracle.hash_embed(text, dim))


def test_rescore_arithmetic_matches_oracle(oracle):
    rng = np.random.default_rng(2)
    for n in (3, 31, 384, 770):
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        assert S().cosine_similarity(a, b) == oracle.search_cosine(a, b)      # search.rs:519-533
    assert S().cosine_similarity(np.zeros(8, np.float32), np.ones(8, np.float32)) == 0.0
    # KATs: rag/context_retriever.rs:505-512, rag/result_ranker.rs:598-605, ml/features.rs:481-489
    e1, e2 = np.array([1, 0, 0], np.float32), np.array([0, 1, 0], np.float32)
    assert abs(S().cosine_similarity(e1, e1) - 1.0) < 1e-6 and abs(S().cosine_similarity(e1, e2)) < 1e-6


def test_prefetch_and_minmax_match_oracle(oracle):
    for lim in (0, 1, 5, 10, 100, 2**62):
        assert S().prefetch_k(lim) == oracle.prefetch_k(lim)                   # search.rs:113
    for s in ([0.2, 0.5, 0.8], [0.3, 0.3], [1.0], [-0.5, 0.25, 0.25, 0.9]):
        assert np.array_equal(S().normalize_scores(s), oracle.normalize_scores(s))   # search.rs:574-592


def test_combine_embeddings():
    # search.rs:232-266: mean, then divide by the norm
    rng = np.random.default_rng(8)
    e = rng.standard_normal((5, 64)).astype(np.float32)
    got = S().combine_embeddings(e)
    acc = np.zeros(64, np.float32)
    for r in e:
        acc = (acc + r).astype(np.float32)
    acc = (acc / np.float32(5)).astype(np.float32)
    nsq = np.float32(0)
    for x in acc:
        nsq = np.float32(nsq + np.float32(x * x))
    assert np.array_equal(got, (acc / np.sqrt(nsq)).astype(np.float32))


def test_store_needs_a_gpu():
    m = pkg()
    if m.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(m.CgvError) as ei:
        S().VectorStore(dtype="bf16")
    assert "no CPU fallback" in str(ei.value)
