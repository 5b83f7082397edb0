#!/bin/bash
# PMC counter passes for the coarse kernel (separate passes; --kernel-trace only).
set -u
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "coarse_kernel" --output-format csv -d $R/$OUT -o $name -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $R/$OUT/$name.log 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run p2 FETCH_SIZE TCC_HIT_sum
run p3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run p4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
run p5 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_CYCLES
run p6 TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr
cd $R; python scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
