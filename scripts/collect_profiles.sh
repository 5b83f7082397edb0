#!/bin/bash
# Round profile collection on the GPU box: rocprofv3 kernel stats of the default bench command (serial steps =
# depth 1, so the CSV alone gives per-launch durations) + PMC passes (separate runs, --kernel-trace only) for the
# dominant kernel of each workload. Outputs under gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/.
#   collect_profiles.sh <tag> [workloads...]      (default workloads: c2)
set -u
TAG=${1:-r03}; shift
WLS=${*:-c2}
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u > $R/$OUT/counters_available.txt
for wl in $WLS; do
  K="coarse"
  python $R/bench.py --workload $wl $( [ $wl = c2 ] || echo --cpu-seconds 0 ) 2>/dev/null | tail -1 > $R/$OUT/${wl}_bench.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o $wl -- python $R/bench.py --workload $wl --cpu-seconds 0 --pipelined-steps 0 --latency 0 --check-queries 0 --coalesced-threads 0 --callers 0 > $R/$OUT/${wl}_rocprof_stats.log 2>&1
  pmc() { name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$K" --output-format csv -d $R/$OUT/$wl -o $name -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 --pipelined-steps 0 --latency 0 --check-queries 0 --coalesced-threads 0 --callers 0 > $R/$OUT/${wl}_$name.log 2>&1; }
  mkdir -p $R/$OUT/$wl
  pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
  pmc p2 FETCH_SIZE TCC_HIT_sum
  pmc p3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
  pmc p4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
  pmc p5 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
  cd $R
  python scripts/pmc_summary.py $OUT/$wl > $OUT/${wl}_pmc_summary.txt
  cp $OUT/$wl/pmc_main_kernel.json $OUT/${wl}_pmc_main_kernel.json 2>/dev/null
  python scripts/trace_timeline.py $OUT/${wl}_kernel_trace.csv > $OUT/${wl}_timeline.txt
  echo "== $wl"; cut -c1-300 $OUT/${wl}_bench.json; head -10 $OUT/${wl}_kernel_stats.csv | cut -c1-150; cat $OUT/${wl}_timeline.txt; tail -8 $OUT/${wl}_pmc_summary.txt
  cd /tmp
done
cd $R
rm -f $OUT/*_agent_info.csv $OUT/*.log $OUT/*_domain_stats.csv
rm -rf $OUT/c2 $OUT/c4 $OUT/c5mini $OUT/c3shard
