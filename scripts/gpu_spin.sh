#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 3000; do
  CGV_SPIN_US=$v timeout 200 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('spin_us $v value',r['value'],'step_ms',r['ms_per_step'],'median',r['median_ms_per_step'],'dev_ms',r['pipeline']['device_ms_last_step'])"
done; done
