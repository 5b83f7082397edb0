#!/bin/bash
# A/B of the coarse kernel variants on C2 (and optionally other workloads): smoke first, then bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -5
for v in w8 w4; do
  for wl in ${WLS:-c2}; do
    CGV_COARSE=$v timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --cpu-seconds ${CPUSEC:-0} 2>/dev/null | tail -1 > gpurun_out/ab_${v}_${wl}.json
    python - <<PY
import json
r=json.load(open("gpurun_out/ab_${v}_${wl}.json")); ro=r["roofline"]
print("$v $wl value",r["value"],"median_qps",r["median_qps"],"pipelined",r["pipelined_qps"],"coarse_ms",ro["avg_launch_ms"],ro["achieved"],ro["unit"],"frac",ro["frac"],"fb",r["pipeline"]["fallback_queries"], "recall", r.get("recall_at_10"))
PY
  done
done
