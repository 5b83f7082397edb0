#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in v2 v1; do
export CGV_COARSE=$V
echo "===== variant $V"
timeout 300 python scripts/diag_coarse.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_$V.log
echo "== bench c2"; timeout 900 python bench.py --steps 20 --warmup 3 --cpu-seconds 4 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/bench_c2_$V.log | cut -c1-1400
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$V -o c2 -- python $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
cd $R; python scripts/trace_timeline.py gpurun_out/prof_$V/c2_kernel_trace.csv | tee gpurun_out/timeline_$V.txt
done
