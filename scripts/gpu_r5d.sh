#!/bin/bash
# round 5, fourth call: v_accvgpr_read issue-rate microbenchmark (fp8 epilogue), regression tests of the small-batch path after
# the floor / iota changes, single-query latency lines
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
scripts/ubench/accread > $O/accread.txt 2>&1; cat $O/accread.txt
timeout 900 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_guarantee.py -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-300
for wl in c1 c2shard8 c2; do python scripts/lat_loop.py --workload $wl --nq 1 --iters 300 2>/dev/null | tail -1; done
