#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
for rep in 1 2; do
timeout 300 python bench.py --workload c5mini --steps 10 --warmup 3 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('c5mini value',r['value'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],'fb',r['pipeline']['fallback_queries'])"
done
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
