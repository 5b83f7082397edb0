#!/bin/bash
# bench.py with GPU_MAX_HW_QUEUES=8 (its new default): the default line, the shard, the one-rank RCCL step, the gloo dry run with 2 ranks
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r5q; mkdir -p $O
timeout 600 python bench.py 2> $O/default.err | tail -1 > $O/default_bench.json
timeout 600 python bench.py --workload c2shard8 --cpu-seconds 0 2>/dev/null | tail -1 > $O/c2shard8_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --force-dist --workload c2shard8 2>/dev/null | grep '^{' | tail -1 > $O/c2shard8_force_dist_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --force-dist 2>/dev/null | grep '^{' | tail -1 > $O/c2_force_dist_bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --dist-backend gloo --steps 30 --warmup 3 --pipelined-steps 30 2>/dev/null | grep '^{' | tail -1 > $O/c2_gloo2_dryrun_bench.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*_bench.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'unreadable',e); continue
    r=d.get('roofline') or {}
    print(f.split('/')[-1], d.get('value'), 'ms/step', d.get('ms_per_step'), 'median', d.get('median_ms_per_step'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'traffic', r.get('traffic'),
          'piped_host', (d.get('pipelined_host') or {}).get('ms_per_batch'), (d.get('pipelined_host') or {}).get('same_results_as_serial_step'), 'piped', (d.get('pipelined') or {}).get('ms_per_batch'), 'exact', (d.get('exact_check') or {}).get('ordered_match_rate'), 'recall', d.get('recall_at_10'), 'env', (d.get('config') or {}).get('runtime_env'), 'err', d.get('error'))
PY
