#!/bin/bash
export TMPDIR=/tmp
for wl in c2shard8 c2; do for d in 1 2 3; do
timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --cpu-seconds 0 --depth $d 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$wl depth $d','qps',r['value'],'ms',r['ms_per_step'],'coarse_ms',r['roofline']['avg_launch_ms'],'dev_ms',r['pipeline']['device_ms_last_step'])"
done; done
