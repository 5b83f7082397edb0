#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scripts/ab.py --workload c5mini --variants "gap0:epi=33;late:epi=1" --rounds 3 --steps 6 2>$O/ab_c5.err | tee $O/ab_c5mini.txt
timeout 300 python scripts/ab.py --workload c2 --variants "gap0:epi=33;late:epi=1" --rounds 4 --steps 12 2>$O/ab_c2.err | tee $O/ab_c2.txt
timeout 300 python scripts/ab.py --workload c4 --variants "gap0:epi=33;late:epi=1" --rounds 4 --steps 12 2>$O/ab_c4.err | tee $O/ab_c4.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py tests/test_gpu_robustness.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "not c5_full" 2>&1 | tail -6 | tee $O/pytest.txt
