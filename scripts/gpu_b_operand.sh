#!/bin/bash
# round 6: what the query (B) operand costs the bf16 coarse kernel - counters of the dominant launch under the timing-only
# ablation masks 1024 (no B DMA, no B fragment reads), 2048 (no B DMA), 4096 (no B fragment reads); measurement flavour.
#   WL=c2 bash scripts/gpu_b_operand.sh      -> gpurun_out/b_operand_<wl>.txt
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${WL:-c2}
OUT=$R/gpurun_out/bop_$WL; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for m in ${MASKS:-0 1024 2048 4096}; do
  if [ $m = 0 ]; then unset CGV_ABLATE; else export CGV_ABLATE=$m; fi
  for pass in a b; do
    if [ $pass = a ]; then C="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; else C="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex coarse --output-format csv -d $OUT/m${m}$pass -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 --pipelined-steps 0 --latency 0 --coalesced-threads 0 --check-queries 0 > $OUT/m${m}$pass.log 2>&1
  done
  python - <<PY >> $R/gpurun_out/b_operand_$WL.txt
import csv,glob,collections
row={}
for ps in 'ab':
    f=glob.glob('$OUT/m${m}'+ps+'/**/p_counter_collection.csv',recursive=True)
    t=glob.glob('$OUT/m${m}'+ps+'/**/p_kernel_trace.csv',recursive=True)
    dur={}
    for r in csv.DictReader(open(t[0])):
        dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    acc=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        acc[r['Dispatch_Id']][r['Counter_Name']]=acc[r['Dispatch_Id']].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
    best=max((d for d in acc if d in dur), key=lambda d: dur[d])
    row.update(acc[best]); row['ns_'+ps]=dur[best]
g=row['GRBM_GUI_ACTIVE']; ns=row['ns_a']
cyc=g/8 if g/ns>4 else g
print('$WL mask %5s  launch %.3f ms  clock %.3f GHz  MFMA busy %.3f  WAIT_INST_ANY/WAVE %.3f  WAIT_ANY/WAVE %.3f | INSTS per MFMA: LDS %.3f VMEM %.3f VALU %.3f SALU %.3f  LDS_IDX_ACTIVE/cycle %.3f'%(
  '$m', ns/1e6, cyc/ns, row['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc, row['SQ_WAIT_INST_ANY']/row['SQ_WAVE_CYCLES'], row['SQ_WAIT_ANY']/row['SQ_WAVE_CYCLES'],
  row['SQ_INSTS_LDS']/row['SQ_INSTS_MFMA'], row['SQ_INSTS_VMEM']/row['SQ_INSTS_MFMA'], row['SQ_INSTS_VALU']/row['SQ_INSTS_MFMA'], row['SQ_INSTS_SALU']/row['SQ_INSTS_MFMA'],
  row['SQ_LDS_IDX_ACTIVE']/256/cyc))
PY
done
cat $R/gpurun_out/b_operand_$WL.txt
