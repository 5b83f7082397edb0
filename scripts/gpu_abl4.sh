#!/bin/bash
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so   # the measurement flavour (make ABLATE=1): ablation masks, knobs, traces
# timing-only ablations of the one-wave-per-SIMD coarse kernels (results are wrong for masks != 0): main-launch ms per mask.
# bf16 (WL=c2, needs CGV_COARSE=w4): masks 1 3 5 9 11 17; fp8 (WL=c5mini, the default kernel): masks 1 3 9
export TMPDIR=/tmp
for m in ${MASKS:-0 1 5 9 3 11}; do
  CGV_COARSE=${VARIANT:-w4} CGV_ABLATE_W4=$m timeout 200 python bench.py --workload ${WL:-c2} --steps ${STEPS:-10} --warmup 3 --cpu-seconds 0 --pipelined-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('${WL:-c2} w4 abl $m coarse_ms',ro['avg_launch_ms'],ro['achieved'],'step_ms',r['ms_per_step'])"
done
