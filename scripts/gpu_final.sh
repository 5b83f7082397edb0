#!/bin/bash
# round-end collection: profiles (kernel stats + timeline + 5 PMC passes, source-stamped) for four workloads, plain bench
# lines for the rest, the one-handle sharded path, and the bf16 clock ablation
export TMPDIR=/tmp
TAG=${1:-r03}
bash scripts/collect_profiles.sh $TAG c2 c4 c3shard c5mini > gpurun_out/collect_$TAG.log 2>&1
# same-box reference: the stage loop of the round's start (dynamic issue side, epi=9) against the final one, interleaved
timeout 300 python scripts/ab.py --workload c2 --variants "dynamic_issue_side:epi=9;final:epi=1" --rounds 3 --steps 12 2>/dev/null > gpurun_out/profiles_$TAG/c2_same_box_ab.txt
cat gpurun_out/profiles_$TAG/c2_same_box_ab.txt | cut -c1-330
tail -60 gpurun_out/collect_$TAG.log | cut -c1-220
for wl in c2shard8 c2f32; do
  timeout 900 python bench.py --workload $wl --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/profiles_$TAG/${wl}_bench.json
  cut -c1-200 gpurun_out/profiles_$TAG/${wl}_bench.json
done
for g in 2 4; do
  timeout 300 python bench.py --workload c2 --sharded-handle $g 2>/dev/null | tail -1 > gpurun_out/profiles_$TAG/c2_sharded${g}_bench.json
done
