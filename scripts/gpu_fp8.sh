#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py -m gpu -q -x -k "fp8 or randomised" -p no:cacheprovider 2>&1 | tail -4
for v in default w8; do
  if [ $v = default ]; then unset CGV_COARSE; else export CGV_COARSE=$v; fi
  timeout 300 python bench.py --workload c5mini --steps 10 --warmup 3 --cpu-seconds ${CPUSEC:-0} --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('c5mini $v value',r['value'],'step_ms',r['ms_per_step'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],ro['unit'],'frac',ro['frac'],'fb',r['pipeline']['fallback_queries'],'recall',r.get('recall_at_10'))"
done
unset CGV_COARSE
for m in 1 3 9; do
  CGV_ABLATE_W4=$m timeout 300 python bench.py --workload c5mini --steps 5 --warmup 2 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('c5mini abl $m coarse_ms',ro['avg_launch_ms'],ro['achieved'])"
done
