#!/bin/bash
# round-3 third experiment: epilogue variants of the bf16 coarse kernel, publish kernel, reduced re-score set
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; rm -rf $O; mkdir -p $O
cd $R
V="epi0:epi=0;epi1:epi=1;epi2:epi=2;epi3:epi=3"
timeout 400 python scripts/ab.py --workload c2 --variants "$V" --rounds 4 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err
timeout 200 python scripts/ab.py --workload c2shard8 --variants "$V" --rounds 4 --steps 12 > $O/ab_c2shard8.txt 2>$O/ab_c2shard8.err
timeout 300 python scripts/ab.py --workload c3shard --variants "$V" --rounds 2 --steps 6 > $O/ab_c3shard.txt 2>$O/ab_c3shard.err
CGV_TRACE=1 timeout 200 python scripts/ab.py --workload c2 --variants "new:" --rounds 1 --steps 5 --trace > $O/trace_c2.txt 2>$O/trace_c2.err
cat $O/ab_*.txt $O/trace_c2.txt
for wl in c2; do
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$wl -o t -- python $R/bench.py --workload $wl --steps 6 --warmup 3 --cpu-seconds 0 --pipelined-steps 0 > $O/tl_$wl.log 2>&1
  cd $R; python scripts/trace_timeline.py $O/tl_$wl/t_kernel_trace.csv > $O/${wl}_timeline.txt 2>&1; cat $O/${wl}_timeline.txt
done
CGV_EPI=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py tests/test_gpu_robustness.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -8 > $O/pytest_epi3.txt; cat $O/pytest_epi3.txt
