#!/bin/bash
# the exact scan of a few queries as ONE kernel (kernels_exact_small.h): the -m gpu suite, then BASELINE config 1 on the device
# (c1: one query per call) with the kernel on / off in one process, and its kernel timeline
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5m; O=gpurun_out/r5m
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python scripts/ab.py --workload c1 --variants "staged:exact_small=0;fused:exact_small=1" --rounds 5 --steps 200 > $O/ab_c1.txt 2>&1; grep -h variant $O/ab_c1.txt | cut -c1-400
timeout 600 python bench.py --workload c1 2> $O/c1.err | tail -1 > $O/c1_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5m/c1_bench.json').read().strip().splitlines()[-1])
print('c1', d.get('value'), d.get('ms_per_step'), d.get('median_ms_per_step'), json.dumps(d.get('latency'))[:900], d.get('error'))
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/lat_c1 -o p -- python $R/scripts/lat_loop.py --workload c1 --nq 1 --iters 300 > $R/$O/lat_c1.log 2>&1
cd $R; f=$(find $O/lat_c1 -name 'p_kernel_stats.csv' | head -1); head -12 $f | cut -c1-200; tail -3 $O/lat_c1.log
