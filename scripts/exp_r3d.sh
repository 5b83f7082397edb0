#!/bin/bash
# round-3 experiment d: default = interleaved epilogue; NT hint for single-query-tile launches (C4); event overhead; final kernel trims
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python scripts/ab.py --workload c2 --variants "epi0:epi=0;new:;noprof:profiling=0" --rounds 4 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err
timeout 300 python scripts/ab.py --workload c4 --variants "nt_off:epi=3;nt_on:;noprof:profiling=0" --rounds 4 --steps 12 > $O/ab_c4.txt 2>$O/ab_c4.err
timeout 200 python scripts/ab.py --workload c2shard8 --variants "new:;noprof:profiling=0" --rounds 4 --steps 12 > $O/ab_c2shard8.txt 2>$O/ab_c2shard8.err
CGV_TRACE=1 timeout 200 python scripts/ab.py --workload c2 --variants "new:" --rounds 1 --steps 5 --trace > $O/trace_c2.txt 2>$O/trace_c2.err
cat $O/ab_*.txt $O/trace_c2.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
for wl in c2 c4; do
  timeout 300 python bench.py --workload $wl --cpu-seconds 0 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json; cut -c1-300 $O/bench_$wl.json
done
