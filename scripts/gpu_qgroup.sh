#!/bin/bash
export TMPDIR=/tmp
for wl in c5mini c3shard c2; do
for v in 0 -1; do
  if [ $v = -1 ]; then unset CGV_QGROUP; else export CGV_QGROUP=$v; fi
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-seconds ${CPUSEC:-0} --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$wl qgroup=$v value',r['value'],'step_ms',r['ms_per_step'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],ro['unit'],'frac',ro['frac'],'fb',r['pipeline']['fallback_queries'],'recall',r.get('recall_at_10'))"
done; done
