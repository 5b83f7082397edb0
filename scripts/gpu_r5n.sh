#!/bin/bash
# small MFMA batches: the final kernel publishes the flags itself (knob self_publish), the 4-way merge is parallel: suite, the
# single-query latency of the 125 k-row shard / C2's corpus, and the C2 headline step (the final kernel carries 4 more words)
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5n; O=gpurun_out/r5n
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-300
for wl in c2shard8 c2; do
  timeout 600 python bench.py --workload $wl --cpu-seconds 0 2> $O/$wl.err | tail -1 > $O/${wl}_bench.json
  python - <<PY
import json
d=json.loads(open('$O/${wl}_bench.json').read().strip().splitlines()[-1]); L=d.get('latency') or {}
print('$wl', d.get('value'), d.get('ms_per_step'), d.get('median_ms_per_step'), 'launch', (d.get('roofline') or {}).get('avg_launch_ms'), 'piped_host', (d.get('pipelined_host') or {}).get('ms_per_batch'), d.get('error'))
print('   nq1', L.get('nq1'), 'nq8', (L.get('nq8') or {}).get('pageable_us'), 'nq32', (L.get('nq32') or {}).get('pageable_us'))
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/lat_s8 -o p -- python $R/scripts/lat_loop.py --workload c2shard8 --nq 1 --iters 300 > $R/$O/lat_s8.log 2>&1
cd $R; f=$(find $O/lat_s8 -name 'p_kernel_stats.csv' | head -1); grep "cgv::" $f | cut -c1-110 | head -6; tail -2 $O/lat_s8.log | cut -c1-200
