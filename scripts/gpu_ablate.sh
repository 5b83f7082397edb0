#!/bin/bash
export TMPDIR=/tmp
for X in 0 4 5 13 21 29; do
echo -n "CGV_DBG=$X (1=noDMA 4=noEPI 8=noBAR 16=noFRAG): "; CGV_DBG=$X timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('coarse_ms', r['roofline']['avg_launch_ms'])"
done
