#!/bin/bash
# round-3 first GPU call: baseline vs the sample-launch plan, kernel timelines, phase stamps, bf16 clock ablation
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; rm -rf $O; mkdir -p $O
cd $R
V="legacy:plan_legacy=1;new:;new_m1:plan_launches=1;new_m3:plan_launches=3;new_s32:sample_tiles=32"
timeout 300 python scripts/ab.py --workload c2 --variants "$V" --rounds 3 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err
timeout 200 python scripts/ab.py --workload c2shard8 --variants "legacy:plan_legacy=1;new:;new_m2:plan_launches=2;new_s32:sample_tiles=32;new_s16:sample_tiles=16" --rounds 3 --steps 12 > $O/ab_c2shard8.txt 2>$O/ab_c2shard8.err
timeout 300 python scripts/ab.py --workload c4 --variants "legacy:plan_legacy=1;new:;new_m2:plan_launches=2" --rounds 3 --steps 12 > $O/ab_c4.txt 2>$O/ab_c4.err
timeout 300 python scripts/ab.py --workload c3shard --variants "legacy:plan_legacy=1;new:;new_m2:plan_launches=2;new_s64:sample_tiles=64" --rounds 2 --steps 8 > $O/ab_c3shard.txt 2>$O/ab_c3shard.err
CGV_TRACE=1 timeout 200 python scripts/ab.py --workload c2 --variants "new:" --rounds 1 --steps 5 --trace --dev-resident > $O/trace_c2.txt 2>$O/trace_c2.err
cat $O/ab_*.txt $O/trace_c2.txt
# kernel timelines (new plan)
for wl in c2 c2shard8; do
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$wl -o t -- python $R/bench.py --workload $wl --steps 6 --warmup 3 --cpu-seconds 0 --pipelined-steps 0 > $O/tl_$wl.log 2>&1
  cd $R; python scripts/trace_timeline.py $O/tl_$wl/t_kernel_trace.csv > $O/${wl}_timeline.txt 2>&1; cat $O/${wl}_timeline.txt
done
# parity: the whole GPU suite on the new plan
timeout 1200 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
# bf16 clock ablation of the C2 main launch: full, no epilogue, + DMA only for the first ring, MFMA only on real data, MFMA only on zeros
WL=c2 MASKS="0 1 65 197 15" bash scripts/gpu_clock.sh > $O/clock_c2.txt 2>&1; cat $O/clock_c2.txt
