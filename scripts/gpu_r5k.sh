#!/bin/bash
# two workgroups per CU (kernels_coarse_wg2.h, knob epi bit 10) against the production kernel: identical results? launch / step
# time? cycles / clock / MFMA busy of the C2 main launch (the row VERDICT r4 'Next' 8 asks for)
set -x
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r5k
export TMPDIR=/tmp
for wl in c2 c3shard c2shard8; do
  timeout 300 python $R/scripts/ab.py --workload $wl --variants "base:;wg2:epi=1025" --rounds 3 --steps 15 > $R/gpurun_out/r5k/ab_$wl.txt 2>&1
  grep -h "coarse_wg2_kernel\|variant" $R/gpurun_out/r5k/ab_$wl.txt | tail -4
done
EPIS="1 1025" bash $R/scripts/gpu_clock_variants.sh > $R/gpurun_out/r5k/clock.txt 2>&1
grep "^epi" $R/gpurun_out/r5k/clock.txt
