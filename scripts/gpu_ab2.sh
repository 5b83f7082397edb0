#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
for rep in 1 2; do
for v in "CGV_W8_LAYOUT=0" "CGV_W8_LAYOUT=1" "CGV_COARSE=w4"; do
  env $v timeout 200 python bench.py --steps 20 --warmup 3 --cpu-seconds ${CPUSEC:-0} --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$v coarse_ms',ro['avg_launch_ms'],ro['achieved'],'step_ms',r['ms_per_step'],'dev_ms',r['pipeline']['device_ms_last_step'],'recall',r.get('recall_at_10'))"
done; done
