#!/bin/bash
set -u
export TMPDIR=/tmp
for X in 0 64 128 192 320; do
echo "== LD_EXTRA=$X"; CGV_LD_EXTRA=$X timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['achieved'])"
done
