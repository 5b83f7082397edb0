#!/bin/bash
# round-3 second GPU call: zero-copy host buffers, flag publication by the last kernel, faster final kernel, 16-byte prep
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
V="zc0:zero_copy=0;zc_in:zero_copy=1;zc_io:;pace_on:pace=2;legacy_zc0:plan_legacy=1,zero_copy=0"
timeout 300 python scripts/ab.py --workload c2 --variants "$V" --rounds 3 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err
timeout 200 python scripts/ab.py --workload c2shard8 --variants "zc0:zero_copy=0;zc_in:zero_copy=1;zc_io:" --rounds 3 --steps 12 > $O/ab_c2shard8.txt 2>$O/ab_c2shard8.err
timeout 300 python scripts/ab.py --workload c4 --variants "zc0:zero_copy=0;zc_in:zero_copy=1;zc_io:" --rounds 3 --steps 12 > $O/ab_c4.txt 2>$O/ab_c4.err
CGV_TRACE=1 timeout 200 python scripts/ab.py --workload c2 --variants "new:" --rounds 1 --steps 5 --trace > $O/trace_c2.txt 2>$O/trace_c2.err
cat $O/ab_*.txt $O/trace_c2.txt
for wl in c2 c2shard8; do
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$wl -o t -- python $R/bench.py --workload $wl --steps 6 --warmup 3 --cpu-seconds 0 --pipelined-steps 0 > $O/tl_$wl.log 2>&1
  cd $R; python scripts/trace_timeline.py $O/tl_$wl/t_kernel_trace.csv > $O/${wl}_timeline.txt 2>&1; cat $O/${wl}_timeline.txt
done
timeout 300 python bench.py --workload c2 --cpu-seconds 0 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json; cut -c1-400 $O/bench_c2.json
