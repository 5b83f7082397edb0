#!/bin/bash
export TMPDIR=/tmp
for W in c4 c3shard; do
echo "== $W"; timeout 900 python bench.py --workload $W --steps 10 --warmup 2 --cpu-seconds 6 --cpu-max-queries 8 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('qps', r['value'], 'step_ms', r['ms_per_step'], 'roof', r['roofline'] and (r['roofline']['avg_launch_ms'], r['roofline']['achieved']), 'pipe', r['pipeline'], 'cpu', r.get('cpu_baseline') and r['cpu_baseline']['value'], 'recall', r.get('recall_at_10'), r.get('ordered_match_rate'), r.get('score_bit_exact_rate'))"
done
