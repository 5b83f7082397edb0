#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -15
for wl in c2 c4 c3shard; do
timeout 600 python bench.py --workload $wl --steps 40 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_$wl.json | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['workload'][:3],'qps',r['value'],'ms',r['ms_per_step'],'coarse_ms',r['roofline']['avg_launch_ms'],'frac',r['roofline']['frac'],'fb',r['pipeline']['fallback_queries'])"
done
