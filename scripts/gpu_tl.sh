#!/bin/bash
# kernel timeline of one workload at a given depth: gpu_tl.sh <workload> <depth>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tl_$1_d$2; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --workload $1 --steps 6 --warmup 3 --cpu-seconds 0 --depth $2 > $OUT/log.txt 2>&1
cd $R; python scripts/trace_timeline.py $OUT/t_kernel_trace.csv | tee $OUT/timeline.txt
