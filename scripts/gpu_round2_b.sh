#!/bin/bash
# round-2 call B: full -m gpu suite, rocprofv3 stats + PMC passes for four workloads, the remaining bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=8 ) > gpurun_out/r2b_pytest.log 2>&1
tail -15 gpurun_out/r2b_pytest.log
bash scripts/collect_profiles.sh r02 c2 c5mini c4 c3shard > gpurun_out/r2b_profiles.log 2>&1
grep -A12 "^== " gpurun_out/r2b_profiles.log | grep -v "^\"void at::\|rocclr" | cut -c1-220
for wl in c5shard c2f32 c2shard8; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r2b_bench_$wl.json
  python -c "
import json; r=json.load(open('gpurun_out/r2b_bench_$wl.json')); ro=r['roofline']; print('$wl value',r['value'],'step_ms',r['ms_per_step'],'pipelined',r['pipelined_qps'],'coarse_ms',ro['avg_launch_ms'],ro['achieved'],ro['unit'],'frac',ro['frac'],'fb',r['pipeline']['fallback_queries'])"
done
