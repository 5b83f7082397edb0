#!/bin/bash
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so   # the measurement flavour (make ABLATE=1): ablation masks, knobs, traces
# clock of the coarse kernel under ablations: GRBM_GUI_ACTIVE cycles / kernel duration (is the kernel power-limited?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clock; rm -rf $OUT; mkdir -p $OUT
cd /tmp
WL=${WL:-c5mini}
for m in ${MASKS:-0 1 3}; do
  if [ $m = 0 ]; then unset CGV_ABLATE_W4 CGV_ABLATE; else export CGV_ABLATE_W4=$m CGV_ABLATE=$m; fi
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-include-regex coarse --output-format csv -d $OUT/m$m -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-seconds 0 --pipelined-steps 0 > $OUT/m$m.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('$OUT/m$m/**/p_counter_collection.csv',recursive=True)
t=glob.glob('$OUT/m$m/**/p_kernel_trace.csv',recursive=True)
dur={}
for r in csv.DictReader(open(t[0])):
    dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'])
acc=collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    acc[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
best=None
for d,c in acc.items():
    if d in dur and (best is None or dur[d][0]>dur[best][0]): best=d
c=acc[best]; ns=dur[best][0]
cyc=c['GRBM_GUI_ACTIVE']/8 if c['GRBM_GUI_ACTIVE']>1e9 else c['GRBM_GUI_ACTIVE']
print('mask $m longest coarse launch %.3f ms  GRBM %.0f  clock(GRBM/8/t) %.3f GHz clock(GRBM/t) %.3f  MFMA busy %.0f  SQ_BUSY %.0f'%(ns/1e6,c['GRBM_GUI_ACTIVE'],c['GRBM_GUI_ACTIVE']/8/ns,c['GRBM_GUI_ACTIVE']/ns,c['SQ_VALU_MFMA_BUSY_CYCLES'],c['SQ_BUSY_CYCLES']))
PY
done
