#!/bin/bash
# round 5, fifth call: fp8 epilogue over two k-steps (EPI2) A/B on C5-mini, corrected accvgpr_read microbenchmark
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
scripts/ubench/accread > $O/accread.txt 2>&1; cat $O/accread.txt
timeout 900 python scripts/ab.py --workload c5mini --variants "epi1:epi=1;epi2:epi=513;epi1b:epi=1" --rounds 3 --steps 6 > $O/ab_c5mini.txt 2> $O/ab_c5mini.err; echo "rc=$?"; cut -c1-330 $O/ab_c5mini.txt; tail -3 $O/ab_c5mini.err | cut -c1-200
