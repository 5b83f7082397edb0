#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r03c
timeout 300 python bench.py --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 > gpurun_out/profiles_r03c/c2_probe_bench.json
timeout 600 python bench.py --workload c5shard --cpu-seconds 0 --steps 5 --warmup 2 --pipelined-steps 4 2>/dev/null | tail -1 > gpurun_out/profiles_r03c/c5shard_bench.json
