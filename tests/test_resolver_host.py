"""Host logic of the symbol resolver mirror (indexer.rs:2804-2821, 2901-2932): no GPU needed."""
import importlib

import pytest

from _util import pkg


@pytest.mark.parametrize("a,b", [("parse_file", "parseFile"), ("HashMap", "hash_map"), ("ab", "ab"), ("ab", "abc"),
                                 ("", "x"), ("naïve_fn", "naive_fn"), ("get_node_embedding", "get_embedding"),
                                 ("x", "extremely_long_symbol_name")])
def test_trigram_jaccard_and_filter_match_oracle(oracle, a, b):
    st = importlib.import_module("codegraph-rust_amd.store")
    assert st.trigram_jaccard(a, b) == oracle.trigram_jaccard(a, b)
    la, lb = len(a.lower().encode()), len(b.lower().encode())
    ratio_ok = la > 0 and lb > 0 and min(la / lb, lb / la) >= 0.5
    assert st.symbol_name_eligible(a, b) == (ratio_ok and oracle.trigram_jaccard(a, b) >= 0.2)


def test_resolver_needs_gpu():
    m = pkg()
    if m.device_count() > 0:
        pytest.skip("GPU present")
    st = importlib.import_module("codegraph-rust_amd.store")
    with pytest.raises(m.CgvError, match="no CPU fallback"):
        st.SymbolResolver(64)
