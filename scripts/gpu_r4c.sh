#!/bin/bash
# round 4, call C: the fence-free rendezvous of the fused launch (A/B against the three-launch form), static priority probe,
# the join-free sharded handle (tests + C2 / C3 through one handle)
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout 900 python -X faulthandler -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py tests/test_gpu_store.py tests/test_c_client.py -m gpu -x -q -p no:cacheprovider --timeout 600 > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_subset.txt | cut -c1-200
for wl in c2 c2shard8; do
  timeout 300 python scripts/ab.py --workload $wl --variants "classic:fuse_sample=0;fused:fuse_sample=1" --rounds 3 --steps 30 > $O/ab_fuse_$wl.txt 2>$O/ab_fuse_$wl.err; tail -2 $O/ab_fuse_$wl.txt | cut -c1-330
done
timeout 300 python scripts/ab.py --workload c2 --variants "classic:fuse_sample=0;prio:fuse_sample=0,epi=65" --rounds 4 --steps 30 > $O/ab_prio_c2.txt 2>$O/ab_prio_c2.err; tail -2 $O/ab_prio_c2.txt | cut -c1-330
timeout 300 python scripts/ab.py --workload c3shard --variants "classic:fuse_sample=0;fused:fuse_sample=1;prio:fuse_sample=0,epi=65" --rounds 2 --steps 10 > $O/ab_c3shard.txt 2>$O/ab_c3shard.err; tail -3 $O/ab_c3shard.txt | cut -c1-330
timeout 300 python bench.py --workload c2 --sharded-handle 2 --steps 50 > $O/c2_sharded2_bench.json 2>$O/sh2.err; cut -c1-700 $O/c2_sharded2_bench.json
timeout 600 python bench.py --workload c3 --sharded-handle 8 --steps 10 --warmup 3 > $O/c3_sharded8_bench.json 2>$O/sh8.err; cut -c1-1200 $O/c3_sharded8_bench.json; tail -3 $O/sh8.err
