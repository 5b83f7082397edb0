#!/bin/bash
# every bench workload once (no CPU leg): one line each
export TMPDIR=/tmp
for wl in c2 c4 c3shard c5mini c2f32 c2shard8; do
timeout 600 python bench.py --workload $wl --steps ${STEPS:-30} --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$wl','qps',r['value'],'ms',r['ms_per_step'],'coarse_ms',ro['avg_launch_ms'],ro['bound'],ro['achieved'],ro['unit'],'frac',ro['frac'],'fb',r['pipeline']['fallback_queries'])"
done
