#!/bin/bash
# end of round: the -m gpu suite in the driver's form once more, and the "after" trace of the N = 1 pipelined_host leg (kernels + memory copies)
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r5r; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.txt | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O -o nd -- python $R/bench.py --workload c2shard8 --steps 5 --warmup 3 --pipelined-steps 60 --callers 0 --latency 0 --check-queries 0 --cpu-seconds 0 > $R/$O/nd.log 2>&1
grep '^{' $R/$O/nd.log | tail -1 | cut -c1-200; ls $R/$O
