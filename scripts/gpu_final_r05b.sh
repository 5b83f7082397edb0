#!/bin/bash
# end of round 5, after the one-kernel exact scan / self-publishing small batches: the -m gpu suite in the driver's form and the
# lines whose numbers those changes touch or must not touch (default C2 line, one rank's shard, the one-rank RCCL step, config 1),
# + the kernels of a single-query call on config 1 and on the 125 k-row shard
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
TAG=${1:-r05b}
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/profiles_$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.txt | tail -1
timeout 600 python bench.py 2> $O/default.err | tail -1 > $O/default_bench.json
timeout 600 python bench.py --workload c2shard8 --cpu-seconds 0 2>/dev/null | tail -1 > $O/c2shard8_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --force-dist --workload c2shard8 2>/dev/null | grep '^{' | tail -1 > $O/c2shard8_force_dist_bench.json
timeout 300 python bench.py --workload c1 2>/dev/null | tail -1 > $O/c1_bench.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*_bench.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'unreadable',e); continue
    r=d.get('roofline') or {}; L=d.get('latency') or {}
    print(f.split('/')[-1], d.get('value'), 'ms/step', d.get('ms_per_step'), 'median', d.get('median_ms_per_step'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'clock', (r.get('limits') or {}).get('clock_ghz'),
          'piped_host', (d.get('pipelined_host') or {}).get('ms_per_batch'), 'piped', (d.get('pipelined') or {}).get('ms_per_batch'), 'exact', (d.get('exact_check') or {}).get('ordered_match_rate'), 'recall', d.get('recall_at_10'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'err', d.get('error'))
    if L: print('    nq1', (L.get('nq1') or {}).get('pageable_us'), 'nq8', (L.get('nq8') or {}).get('pageable_us'), 'nq32', (L.get('nq32') or {}).get('pageable_us'), 'store', (L.get('cgvs_search_similar_nq1') or {}).get('median_us'))
PY
cd /tmp
for wl in c1 c2shard8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O -o lat_$wl -- python $R/scripts/lat_loop.py --workload $wl --nq 1 --iters 300 > $R/$O/lat_$wl.log 2>&1
  grep -v "^W2026\|^E2026" $R/$O/lat_$wl.log | tail -1; rm -f $R/$O/lat_${wl}_kernel_trace.csv
done
cd $R; rm -f $O/*_agent_info.csv $O/*_domain_stats.csv
