#!/bin/bash
# do the batches of the `pipelined_host` legs overlap on the device? kernel traces: N = 1 leg with 8 HW queues, and the one-rank RCCL form
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd /tmp
GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o nd8 -- python $R/bench.py --workload c2shard8 --steps 5 --warmup 3 --pipelined-steps 60 --latency 0 --check-queries 0 --cpu-seconds 0 > $O/nd8.log 2>&1
tail -1 $O/nd8.log | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o fd -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29547 $R/bench.py --gpus 1 --force-dist --workload c2shard8 --steps 5 --warmup 3 --pipelined-steps 60 --latency 0 --check-queries 0 --cpu-seconds 0 > $O/fd.log 2>&1
grep '^{' $O/fd.log | tail -1 | cut -c1-200
ls $O; find $O -name "*kernel_trace.csv" | head
