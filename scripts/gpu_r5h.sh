#!/bin/bash
# round 5: counters of the final kernel (why does the candidate-row gather take ~10 us whether the corpus is in HBM or in the
# Infinity Cache?) - instruction cache, waits, instruction mix, L2
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd /tmp; mkdir -p $R/gpurun_out/r5h; O=$R/gpurun_out/r5h
pmc() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "final_kernel" --output-format csv -d $O/p -o $name -- python $R/bench.py --workload c2 --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 --pipelined-steps 0 --latency 0 --check-queries 0 > $O/$name.log 2>&1; }
pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
pmc p2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
pmc p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmc p4 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $R
python - <<'PY'
import csv,glob,collections
allc={}
for f in sorted(glob.glob('gpurun_out/r5h/p/p*_counter_collection.csv')):
    by=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
    if by:
        for k,v in by[max(by)].items(): allc[k]=v
for k in sorted(allc): print(f"{k:36s} {allc[k]:18.1f}")
PY
ls gpurun_out/r5h/p | head; tail -3 gpurun_out/r5h/p4.log | cut -c1-200
rm -f gpurun_out/r5h/p/*agent_info* gpurun_out/r5h/p/*kernel_trace*
