#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for X in 0 1 2 3; do
echo "== CGV_SDBG=$X"
cd /tmp && CGV_SDBG=$X timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s$X -o c2 -- python $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
cd $R; python scripts/trace_timeline.py gpurun_out/prof_s$X/c2_kernel_trace.csv | grep -E "select|rescore"
done
