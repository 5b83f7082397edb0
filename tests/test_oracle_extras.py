"""Independent pure-Python restatements (numpy float32 scalars: one rounding per operation) of the
reference code behind the newer oracle functions, on tiny inputs: pins oracle/cgv_oracle.cpp for
  optimization.rs:63-150, 212-283, 338-343   (int8 scan, 8-/4-bit quantize_batch)
  persistent.rs:116-477                      (ScalarQuantizer, ProductQuantizer)
  indexer.rs:2790-2843, 2901-2932            (symbol resolver phase 2)
No GPU."""
import numpy as np

F = np.float32


def _round_half_away(x):
    return int(np.floor(abs(float(x)) + 0.5) * (1 if x >= 0 else -1))


def _py_quant_u8(v):
    out = np.empty(v.shape, np.uint8)
    for i, x in np.ndenumerate(v):
        c = F(-1.0) if x < -1 else (F(1.0) if x > 1 else F(x))
        q = 0 if np.isnan(c) else _round_half_away(F(c * F(127.0)))
        out[i] = max(-127, min(127, q)) + 128
    return out


def _py_search_optimized(query, data, limit):
    limit = max(limit, 1)
    n, dim = data.shape
    q8 = _py_quant_u8(np.asarray(query, np.float32)).astype(np.int32) - 128
    nq = F(0.0)
    for v in q8:
        nq = F(nq + F(v) * F(v))
    nq = F(np.sqrt(nq))
    if nq == 0:
        return []
    best = []
    for idx in range(n):
        row = data[idx].astype(np.int32) - 128
        dot = int((row * q8).sum())
        nv = int((row * row).sum())
        if nv == 0:
            continue
        score = F(F(dot) / F(nq * F(np.sqrt(F(nv)))))
        if len(best) < limit:
            best.append((idx, score))
            if len(best) == limit:
                best.sort(key=lambda t: t[1])          # Python's sort is stable, like sort_by
        elif score > best[0][1]:
            best[0] = (idx, score)
            best.sort(key=lambda t: t[1])
    best.sort(key=lambda t: -t[1])
    return [i for i, _ in best]


def test_int8_scan_and_quantisers_against_python(oracle):
    rng = np.random.default_rng(1)
    rows = (rng.standard_normal((300, 24)) * 0.6).astype(np.float32)
    rows[40:60] = rows[7]                                   # ties exercise the buffer policy
    rows[5] = 0.0
    rows[9, :3] = [np.nan, 2.0, -2.0]
    data = oracle.quantize_u8(rows)
    assert np.array_equal(data, _py_quant_u8(rows))
    for q, limit in ((rows[7], 5), (rows[7], 30), (rows[100], 1), (rows[3] * 0, 4)):
        assert oracle.search_optimized_u8(q, data, limit).tolist() == _py_search_optimized(q, data, limit)
    u4 = oracle.quantize_u4(rows[:, :7])                    # odd dimension
    for r in range(20):
        for j in range(0, 7, 2):
            qs = []
            for t in (0, 1):
                if j + t < 7:
                    x = rows[r, j + t]
                    c = F(-1) if x < -1 else (F(1) if x > 1 else F(x))
                    nrm = F(F(c + F(1)) / F(2))
                    qs.append(0 if np.isnan(nrm) else max(0, min(15, _round_half_away(F(nrm * F(15))))))
                else:
                    qs.append(0)
            assert u4[r, j // 2] == (qs[0] & 15) | ((qs[1] & 15) << 4)


def test_scalar_quantizer_against_python(oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((50, 6)).astype(np.float32)
    x[:, 2] = 1.5
    for nbits, uniform in ((8, False), (4, True), (16, False)):
        sc, bi = oracle.sq_train(x, nbits, uniform)
        levels = F(1 << nbits)
        if uniform:
            lo, hi = F(x.min()), F(x.max())
            assert np.all(sc == F(levels / F(hi - lo))) and np.all(bi == lo)
        else:
            for d in range(6):
                lo, hi = F(x[:, d].min()), F(x[:, d].max())
                rng_ = F(hi - lo)
                assert (sc[d], bi[d]) == ((F(levels / rng_), lo) if rng_ > 0 else (F(1), F(0)))
        codes = oracle.sq_encode(x[:5], nbits, sc, bi)
        bpv = 1 if nbits == 8 else (2 if nbits == 16 else 4)
        for r in range(5):
            for d in range(6):
                norm = F(F(x[r, d] - bi[d]) * sc[d])
                q = int(min(max(norm, F(0)), F((1 << nbits) - 1)))      # as u32: truncation
                got = int.from_bytes(codes[r, d * bpv:(d + 1) * bpv].tobytes(), "little")
                assert got == q
        dec = oracle.sq_decode(codes, 6, nbits, sc, bi)
        for r in range(5):
            for d in range(6):
                q = int.from_bytes(codes[r, d * bpv:(d + 1) * bpv].tobytes(), "little")
                assert dec[r, d] == F(F(F(q) / sc[d]) + bi[d])


def _py_dist(a, b):
    s = F(0)
    for x, y in zip(a, b):
        d = F(x - y)
        s = F(s + F(d * d))
    return F(np.sqrt(s))


def test_product_quantizer_against_python(oracle):
    rng = np.random.default_rng(3)
    n, dim, m, nbits = 40, 6, 2, 2
    x = (rng.standard_normal((n, dim)) + rng.integers(0, 2, (n, 1)) * 3).astype(np.float32)
    ksub, dsub = 1 << nbits, dim // m
    cent = np.zeros((m, ksub, dsub), np.float32)
    for sub in range(m):
        v = x[:, sub * dsub:(sub + 1) * dsub]
        c = np.stack([v[i % n].copy() for i in range(ksub)])
        for _ in range(50):
            assign, changed = [], False
            for r in range(n):
                best, bd = 0, F(np.inf)
                for k_ in range(ksub):
                    d = _py_dist(v[r], c[k_])
                    if d < bd:
                        bd, best = d, k_
                changed |= best != 0
                assign.append(best)
            for k_ in range(ksub):
                rows_k = [r for r in range(n) if assign[r] == k_]
                if rows_k:
                    acc = np.zeros(dsub, np.float32)
                    for r in rows_k:
                        acc = (acc + v[r]).astype(np.float32)
                    c[k_] = (acc / F(len(rows_k))).astype(np.float32)
            if not changed:
                break
        cent[sub] = c
    ref = oracle.pq_train(x, m, nbits)
    assert np.array_equal(ref, cent)
    codes = oracle.pq_encode(x[:10], ref)
    for r in range(10):
        for sub in range(m):
            d = [_py_dist(x[r, sub * dsub:(sub + 1) * dsub], cent[sub, k_]) for k_ in range(ksub)]
            assert codes[r, sub] == int(np.argmin(np.array(d, np.float32)))   # first minimum


def test_symbol_matcher_against_python(oracle):
    def tri(s):
        s = s.lower()
        return {s} if 0 < len(s) < 3 else {s[i:i + 3] for i in range(len(s) - 2)}

    def jac(a, b):
        ta, tb = tri(a), tri(b)
        if not ta or not tb:
            return F(0)
        inter = F(len(ta & tb))
        uni = F(F(len(ta) + len(tb)) - inter)
        return F(0) if uni == 0 else F(inter / uni)

    rng = np.random.default_rng(4)
    names = ["parse_file", "ParseFile", "parse_files_impl", "hash_map", "x", "graph_store", "file_parser"]
    embs = rng.standard_normal((len(names), 16)).astype(np.float32)
    for a in names:
        for b in names:
            assert F(oracle.trigram_jaccard(a, b)) == jac(a, b)
    target = "parse_file_x"
    temb = (embs[0] + 0.2 * rng.standard_normal(16)).astype(np.float32)
    embs[2] = temb * F(1.5)                                  # cosine ~1 but must pass the name filter too
    best, best_s = -1, F(0)
    for i, nm in enumerate(names):
        la, lb = F(len(target)), F(len(nm))
        if not (min(F(la / lb), F(lb / la)) >= F(0.5)) or not (jac(target, nm) >= F(0.2)):
            continue
        s = F(oracle.search_cosine(temb, embs[i]))
        if s > F(0.75) and (best < 0 or s > best_s):
            best, best_s = i, s
    assert oracle.symbol_match_phase2(target, temb, names, embs, 0.75) == (best, float(best_s) if best >= 0 else 0.0)
