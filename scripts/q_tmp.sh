#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash scripts/gpu_final.sh r03
timeout 900 python bench.py --workload c5shard --cpu-seconds 0 --steps 5 --warmup 2 --pipelined-steps 4 2>gpurun_out/c5shard.err | tail -1 > gpurun_out/profiles_r03/c5shard_bench.json
cut -c1-400 gpurun_out/profiles_r03/c5shard_bench.json
