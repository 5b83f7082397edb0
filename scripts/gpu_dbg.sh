#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -20
import importlib, sys, numpy as np, os
sys.path.insert(0, '.')
m = importlib.import_module('codegraph-rust_amd')
from oracle import oracle as o
rng = np.random.default_rng(0)
def run(dtype, odt, n, d, nq, env=None):
    rows = rng.standard_normal((n, d)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = m.HipKnnIndex(d, dtype=dtype); ix.add(rows)
    idx, sc = ix.search(q, 10); st = ix.stats(); ix.close()
    ri, rs = o.batch_top_k(q, rows, 10, dtype=odt)
    print(dtype, n, d, nq, 'ids', np.array_equal(idx, ri), 'scores', np.array_equal(sc, rs), 'fb', st['fallback_queries'], flush=True)
run('bf16', 1, 20000, 256, 64)
os.environ['X']='1'
run('fp8', 3, 5000, 256, 16)
run('fp8', 3, 20000, 768, 300)
PY
CGV_COARSE=w4 python - <<'PY' 2>&1 | tail -5
import importlib, sys, numpy as np
sys.path.insert(0, '.')
m = importlib.import_module('codegraph-rust_amd')
from oracle import oracle as o
rng = np.random.default_rng(0)
n,d,nq=20000,256,64
rows = rng.standard_normal((n, d)).astype(np.float32); q = rng.standard_normal((nq, d)).astype(np.float32)
ix = m.HipKnnIndex(d, dtype='bf16'); ix.add(rows); idx, sc = ix.search(q, 10); st=ix.stats(); ix.close()
ri, rs = o.batch_top_k(q, rows, 10, dtype=1)
print('bf16 w4', np.array_equal(idx, ri), np.array_equal(sc, rs), st['fallback_queries'])
PY
