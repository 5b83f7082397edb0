#!/bin/bash
export TMPDIR=/tmp
echo "== RS correctness"; CGV_RS=1 timeout 300 python scripts/diag_coarse.py 2>&1 | grep -v amdgpu.ids | tail -5
CGV_RS=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x 2>&1 | tail -4
for rep in 1 2; do
for V in "CGV_X=0" "CGV_RS=1"; do
echo -n "[$V] "; env $V timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('qps', r['value'], 'coarse_ms', r['roofline']['avg_launch_ms'], 'TF', r['roofline']['achieved'], 'recall', r.get('recall_at_10'), r.get('score_bit_exact_rate'))"
done; done
