#!/bin/bash
# round-end collection: profiles for four workloads + plain bench lines for the rest
export TMPDIR=/tmp
bash scripts/collect_profiles.sh r02 c2 c5mini c4 c3shard > gpurun_out/collect_r02.log 2>&1
for wl in c5shard c2shard8 c2f32; do
  timeout 900 python bench.py --workload $wl --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/profiles_r02/${wl}_bench.json
  cut -c1-200 gpurun_out/profiles_r02/${wl}_bench.json
done
