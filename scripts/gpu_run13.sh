#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -15
for d in 1 2 3; do
timeout 600 python bench.py --steps 60 --warmup 5 --depth $d --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_c2_d$d.json | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('depth',r['config']['batches_in_flight'],'qps',r['value'],'ms',r['ms_per_step'],'coarse_ms',r['roofline']['avg_launch_ms'],'frac',r['roofline']['frac'])"
done
