#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_store.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x -k "packed or rank or exchange or sharded" 2>&1 | tail -8
for i in 1 2; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --cpu-seconds 0 2>gpurun_out/fd.err | tail -1 > gpurun_out/force_dist_bench.json
python - <<'PY'
import json; j=json.load(open('gpurun_out/force_dist_bench.json')); print('force-dist', j['value'], j['ms_per_step'], j['median_ms_per_step'], j['roofline']['avg_launch_ms'], j['multi_gpu'])
PY
done
timeout 300 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('plain', j['value'], j['ms_per_step'], j['median_ms_per_step'], j['roofline']['avg_launch_ms'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --workload c2shard8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-dist c2shard8', j['value'], j['ms_per_step'], j['median_ms_per_step'], j['roofline']['avg_launch_ms'], j['multi_gpu']['exchange_ms'])"
timeout 300 python bench.py --workload c2shard8 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('plain c2shard8', j['value'], j['ms_per_step'], j['median_ms_per_step'], j['roofline']['avg_launch_ms'])"
