#!/bin/bash
# sample size / number of staged launches around the planner's choice (knobs sample_tiles, plan_launches): is the plan at its optimum?
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r5l
export TMPDIR=/tmp
timeout 300 python $R/scripts/ab.py --workload c2shard8 --variants "auto:;s32:sample_tiles=32;s16:sample_tiles=16;s48:sample_tiles=48;m2:plan_launches=2" --rounds 3 --steps 20 > $R/gpurun_out/r5l/ab_c2shard8.txt 2>&1
grep -h "variant" $R/gpurun_out/r5l/ab_c2shard8.txt | tail -6
timeout 300 python $R/scripts/ab.py --workload c2 --variants "auto:;s128:sample_tiles=128;s32:sample_tiles=32;m3:plan_launches=3;m1:plan_launches=1" --rounds 3 --steps 15 > $R/gpurun_out/r5l/ab_c2.txt 2>&1
grep -h "variant" $R/gpurun_out/r5l/ab_c2.txt | tail -6
timeout 300 python $R/scripts/ab.py --workload c2shard4 --variants "auto:;s32:sample_tiles=32;m1:plan_launches=1;m2:plan_launches=2" --rounds 3 --steps 15 > $R/gpurun_out/r5l/ab_c2shard4.txt 2>&1
grep -h "variant\|Error\|error" $R/gpurun_out/r5l/ab_c2shard4.txt | tail -6
