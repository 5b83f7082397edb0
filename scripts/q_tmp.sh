#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python scripts/ab.py --workload c5mini --variants "dyn:epi=9;si:epi=1" --rounds 3 --steps 6 2>$O/ab_c5.err | tee $O/ab_c5mini.txt
PYTEST_ARGS="--durations=15" bash scripts/gpu_tests.sh | tail -45
