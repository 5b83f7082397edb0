#!/bin/bash
# Cycles / clock / MFMA busy of the C2 main coarse launch for the round-4 kernel variants (measurement flavour, CGV_EPI):
#   1 = the production 8-wave kernel, 65 = + static priority for waves 4-7, 129 = one wave per SIMD (128 x 128 per wave) with the
#   folded epilogue, 385 = the same with its epilogue as one block at the tile boundary.  HISTORY.md §9.2
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clockv; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for e in ${EPIS:-1 65 129 385}; do
  export CGV_EPI=$e
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-include-regex coarse --output-format csv -d $OUT/e$e -o p -- python $R/bench.py --workload c2 --steps 6 --warmup 2 --cpu-seconds 0 --pipelined-steps 0 > $OUT/e$e.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('$OUT/e$e/**/p_counter_collection.csv',recursive=True)
t=glob.glob('$OUT/e$e/**/p_kernel_trace.csv',recursive=True)
dur={}
for r in csv.DictReader(open(t[0])):
    dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'])
acc=collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    acc[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
# the main launches = the coarse dispatches with the longest durations, one per step: take the MEDIAN of the last steps (the
# first launch of a process carries the code-object load)
cand=sorted([d for d in acc if d in dur], key=lambda d: int(d))
longest=max(dur[d][0] for d in cand)
mains=[d for d in cand if dur[d][0] > 0.6*longest][-5:]
mains.sort(key=lambda d: dur[d][0])
best=mains[len(mains)//2]
c=acc[best]; ns=dur[best][0]
cyc=c['GRBM_GUI_ACTIVE']/8
print('epi $e  %-44s  median of %d main launches: %.4f ms  cycles/XCD %.0f  clock %.3f GHz  MFMA busy %.3f'%(dur[best][1][:44], len(mains), ns/1e6, cyc, cyc/ns, c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc))
PY
done
