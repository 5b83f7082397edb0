#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "small or ragged or medium or ties or randomised" -p no:cacheprovider 2>&1 | tail -3
for cfg in "c2 w8" "c2 w4" "c5mini w8" "c5mini w4" "c3shard w8" "c3shard w4"; do
  set -- $cfg
  CGV_COARSE=$2 timeout 300 python bench.py --workload $1 --steps 10 --warmup 3 --cpu-seconds ${CPUSEC:-0} --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$1 $2 value',r['value'],'step_ms',r['ms_per_step'],'dev_ms',r['pipeline']['device_ms_last_step'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],ro['unit'],'frac',ro['frac'],'fb',r['pipeline']['fallback_queries'],'recall',r.get('recall_at_10'))"
done
