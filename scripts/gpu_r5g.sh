#!/bin/bash
# round 5: per-query statistics instead of 1024 contended atomics in the final kernel - suite, phase stamps, C2 / shard lines
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5g; O=gpurun_out/r5g
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
for wl in c2 c2shard8; do
  CGV_TRACE=1 timeout 300 python scripts/ab.py --workload $wl --variants "new:" --rounds 1 --steps 8 --trace > $O/trace_$wl.txt 2>/dev/null; echo "== $wl"; tail -11 $O/trace_$wl.txt | cut -c1-250
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O -o c2 -- python $R/bench.py --workload c2 --cpu-seconds 0 --pipelined-steps 0 --latency 0 --check-queries 0 > $R/$O/c2_rocprof.log 2>&1
cd $R; python scripts/trace_timeline.py $O/c2_kernel_trace.csv | tail -14
python bench.py --workload c2shard8 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2shard8', d['ms_per_step'], d['median_ms_per_step'], d['pipelined_host']['ms_per_batch'], d['latency']['nq1'])"
rm -f $O/*_agent_info.csv $O/*_domain_stats.csv $O/c2_kernel_trace.csv
