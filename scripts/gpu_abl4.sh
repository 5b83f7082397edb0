#!/bin/bash
# timing-only ablations of the W4 coarse kernel (results are wrong for masks != 0): main-launch ms per mask
export TMPDIR=/tmp
for m in ${MASKS:-0 1 5 9 13 3 11}; do
  CGV_ABLATE_W4=$m timeout 200 python bench.py --workload ${WL:-c2} --steps ${STEPS:-10} --warmup 3 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('${WL:-c2} w4 abl $m coarse_ms',ro['avg_launch_ms'],ro['achieved'],'step_ms',r['ms_per_step'])"
done
