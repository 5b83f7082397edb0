#!/bin/bash
# round 4, call H: the full GPU suite N more times in a row (the pinned case in-process, abort_bt hooked in) - does the round-3
# abort recur? - and the kernel timeline of the join-free N > 1 step (one rank)
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r4h; mkdir -p $O
for i in $(seq 1 ${RUNS:-8}); do
  timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu_$i.txt 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -E 'passed|failed|error' $O/pytest_gpu_$i.txt | tail -1) abort_bt=$(grep -c 'abort_bt: native' $O/pytest_gpu_$i.txt)"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/fd -o fd -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 $R/bench.py --gpus 1 --force-dist --workload c2shard8 --cpu-seconds 0 --check-queries 0 --steps 20 --warmup 3 --pipelined-steps 0 --settle-ms 50 > $R/$O/fd.log 2>&1
cd $R
f=$(find $O/fd -name "*kernel_trace.csv" | head -1); echo "trace: $f"
python scripts/trace_timeline.py $f > $O/c2shard8_force_dist_timeline.txt 2>&1; cat $O/c2shard8_force_dist_timeline.txt | cut -c1-200
