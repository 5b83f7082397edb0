#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python scripts/diag_coarse.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x 2>&1 | tail -4
for X in 0 4; do
echo -n "CGV_DBG=$X: "; CGV_DBG=$X timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('qps', r['value'], 'coarse_ms', r['roofline']['avg_launch_ms'], 'TF', r['roofline']['achieved'], 'step_ms', r['ms_per_step'], 'fb', r['pipeline']['fallback_queries'])"
done
