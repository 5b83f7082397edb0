#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PYTEST_ARGS="--durations=8" bash scripts/gpu_tests.sh | tail -30
bash scripts/gpu_final.sh r03
timeout 900 python bench.py --workload c5shard --cpu-seconds 0 --steps 5 --warmup 2 --pipelined-steps 4 2>gpurun_out/c5shard.err | tail -1 > gpurun_out/profiles_r03/c5shard_bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --force-dist --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/profiles_r03/c2_force_dist_bench.json
