#!/bin/bash
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so   # the measurement flavour (make ABLATE=1): ablation masks, knobs, traces
# Timing-only ablations of the bf16 coarse kernel on C2 (CGV_ABLATE mask, kernels_coarse.h):
# 1 no epilogue, 2 no DMA, 4 no barrier, 8 no fragment reads, 16 no vmcnt wait, 32 global_load-lds form
# of the DMA (correct results). Results are wrong
# for any non-zero mask (fallback counts explode) - only the coarse launch time is meaningful.
export TMPDIR=/tmp
for abl in 0 1 2 4 8 10 15 16 32 0; do
CGV_ABLATE=$abl timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --depth 1 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('mask $abl','coarse_ms',r['roofline']['avg_launch_ms'],'ms_per_step',r['ms_per_step'])"
done
