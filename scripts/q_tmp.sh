#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python scripts/ab.py --workload c2 --variants "si1:epi=17;u4:epi=1" --rounds 4 --steps 12 2>$O/ab_c2.err | tee $O/ab_c2.txt
timeout 300 python scripts/ab.py --workload c4 --variants "si1:epi=17;u4:epi=1" --rounds 4 --steps 12 2>$O/ab_c4.err | tee $O/ab_c4.txt
timeout 300 python scripts/ab.py --workload c3shard --variants "si1:epi=17;u4:epi=1" --rounds 2 --steps 6 2>$O/ab_c3.err | tee $O/ab_c3shard.txt
timeout 200 python scripts/ab.py --workload c2shard8 --variants "si1:epi=17;u4:epi=1" --rounds 4 --steps 12 2>$O/ab_c2s.err | tee $O/ab_c2shard8.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py tests/test_gpu_robustness.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "not c5_full" 2>&1 | tail -6 | tee $O/pytest.txt
