#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for m in ${SCHEDS:-0 4}; do
  CGV_W4_SCHED=$m timeout 200 python bench.py --steps 20 --warmup 3 --cpu-seconds ${CPUSEC:-0} --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('w4 sched $m coarse_ms',ro['avg_launch_ms'],ro['achieved'],'step_ms',r['ms_per_step'],'recall',r.get('recall_at_10'))"
done; done
