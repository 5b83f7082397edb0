#!/bin/bash
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -x -q -p no:cacheprovider --timeout 600 > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_subset.txt | cut -c1-200
for wl in c2 c2shard8; do
  timeout 300 python scripts/ab.py --workload $wl --variants "classic:fuse_sample=0;fused:fuse_sample=1" --rounds 3 --steps 30 > $O/ab_fuse_$wl.txt 2>$O/ab_fuse_$wl.err; tail -2 $O/ab_fuse_$wl.txt | cut -c1-330
done
timeout 300 python scripts/ab.py --workload c3shard --variants "classic:fuse_sample=0;fused:fuse_sample=1" --rounds 2 --steps 10 > $O/ab_c3shard.txt 2>$O/ab_c3shard.err; tail -2 $O/ab_c3shard.txt | cut -c1-330
