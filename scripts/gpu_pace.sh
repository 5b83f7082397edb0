#!/bin/bash
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so   # the measurement flavour (make ABLATE=1): ablation masks, knobs, traces
# Workgroup pacing A/B (CGV_NO_PACE=1 = off): HBM fetch traffic of the longest coarse launch (one PMC pass:
# FETCH_SIZE + TCC_HIT_sum; a third TCC counter exceeds what one pass can collect and rocprofv3 aborts)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for wl in ${WLS:-c5mini}; do
for v in pace nopace; do
  if [ $v = pace ]; then unset CGV_NO_PACE; else export CGV_NO_PACE=1; fi
  timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --kernel-include-regex coarse --output-format csv -d $OUT/$wl$v -o p -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --cpu-seconds 0 --pipelined-steps 0 > $OUT/$wl$v.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('$OUT/$wl$v/**/*counter_collection.csv',recursive=True)
t=glob.glob('$OUT/$wl$v/**/*kernel_trace.csv',recursive=True)
if not f or not t:
    print('$wl $v: no output', glob.glob('$OUT/$wl$v/**',recursive=True)[:8])
else:
    dur={}
    for r in csv.DictReader(open(t[0])):
        dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    acc=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        acc[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
    best=max((d for d in acc if d in dur), key=lambda d: dur[d])
    c=acc[best]
    print('$wl $v longest coarse launch %.3f ms FETCH_SIZE x2 = %.2f GB  TCC_HIT %.3e'%(dur[best]/1e6, c['FETCH_SIZE']*2*1024/1e9, c['TCC_HIT_sum']))
PY
done
done
