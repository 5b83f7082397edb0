#!/bin/bash
# Round profile collection on the GPU box: rocprofv3 kernel stats of the default bench
# command + PMC passes (separate runs, --kernel-trace only) for the dominant kernel.
# Outputs under gpurun_out/profiles_rNN/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > $OUT/bench_c2.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o c2 -- python $R/bench.py --cpu-seconds 0 > $R/$OUT/rocprof_stats.log 2>&1
pmc() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "coarse_kernel" --output-format csv -d $R/$OUT -o $name -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $R/$OUT/$name.log 2>&1; }
pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pmc p2 FETCH_SIZE TCC_HIT_sum
pmc p3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc p4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
cd $R
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python scripts/trace_timeline.py $OUT/c2_kernel_trace.csv > $OUT/timeline.txt
cat $OUT/bench_c2.json | cut -c1-400; head -12 $OUT/c2_kernel_stats.csv | cut -c1-160; cat $OUT/timeline.txt; tail -8 $OUT/pmc_summary.txt
rm -f $OUT/*_agent_info.csv $OUT/*.log $OUT/p*_kernel_trace.csv $OUT/c2_domain_stats.csv
