#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/diag_scale.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/diag_scale.log
