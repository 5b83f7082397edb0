#!/bin/bash
# round-3 experiment f: early inverse-norm fetch in the hit path (epi bit 2 = round-2 form), 256-tile sample for C4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python scripts/ab.py --workload c2 --variants "late:epi=5;early:epi=1;m1_late:epi=5,plan_launches=1;m1_early:plan_launches=1" --rounds 4 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err
timeout 300 python scripts/ab.py --workload c4 --variants "s64:sample_tiles=64;s256:;s256_late:epi=5" --rounds 4 --steps 12 > $O/ab_c4.txt 2>$O/ab_c4.err
timeout 300 python scripts/ab.py --workload c3shard --variants "late:epi=5;early:" --rounds 2 --steps 6 > $O/ab_c3shard.txt 2>$O/ab_c3shard.err
timeout 200 python scripts/ab.py --workload c2shard8 --variants "late:epi=5;early:" --rounds 4 --steps 12 > $O/ab_c2shard8.txt 2>$O/ab_c2shard8.err
cat $O/ab_*.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py tests/test_gpu_robustness.py tests/test_gpu_configs.py::test_c4_full_size_fp16_dot -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
