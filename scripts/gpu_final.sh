#!/bin/bash
# round-end collection: the whole GPU suite (driver form), profiles (kernel stats + timeline + 5 PMC passes, source-stamped) for
# four workloads, plain bench lines for the rest, the one-handle sharded path (C2 over 2 / 4 shards, C3 WHOLE over 8), the
# one-rank dry run of the multi-GPU bench
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.txt | cut -c1-200
grep -c "abort_bt: native" gpurun_out/pytest_gpu_$TAG.txt
bash scripts/collect_profiles.sh $TAG c2 c4 c3shard c5mini > gpurun_out/collect_$TAG.log 2>&1
tail -70 gpurun_out/collect_$TAG.log | cut -c1-220
O=gpurun_out/profiles_$TAG
for wl in c2shard8 c2f32; do
  timeout 900 python bench.py --workload $wl --cpu-seconds 0 2>/dev/null | tail -1 > $O/${wl}_bench.json
  cut -c1-200 $O/${wl}_bench.json
done
for g in 2 4; do
  timeout 300 python bench.py --workload c2 --sharded-handle $g --steps 50 2>/dev/null | tail -1 > $O/c2_sharded${g}_bench.json
done
timeout 600 python bench.py --workload c3 --sharded-handle 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/c3_sharded8_bench.json; cut -c1-300 $O/c3_sharded8_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist --cpu-seconds 0 2>/dev/null | tail -1 > $O/c2_force_dist_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --force-dist --workload c2shard8 --cpu-seconds 0 2>/dev/null | tail -1 > $O/c2shard8_force_dist_bench.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/profiles_'"$TAG"'/*_bench.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step', d.get('ms_per_step'), 'median', d.get('median_ms_per_step'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'),
          'pipelined', (d.get('pipelined') or {}).get('ms_per_batch'), 'pcie_inclusive', (d.get('pcie_inclusive_serial') or {}).get('ms_per_step'),
          'xch', (d.get('multi_gpu') or {}).get('exchange_ms'), d.get('last_exchange_ms'), 'recall', d.get('recall_at_10'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
