#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for m in 0 64; do
  if [ $m = 0 ]; then unset CGV_ABLATE; else export CGV_ABLATE=$m; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('c2 abl $m value',r['value'],'median',r['median_qps'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],'fb',r['pipeline']['fallback_queries'],'recall',r.get('recall_at_10'))"
done; done
