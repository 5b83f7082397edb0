#!/bin/bash
export CGV_LIB_PATH=${GRAFT_REPO_ROOT:-.}/codegraph-rust_amd/lib/libcgvec_hip_ablate.so   # the measurement flavour (make ABLATE=1): ablation masks, knobs, traces
# One GPU call's worth of round evidence: the -m gpu suite, a bench line per workload, kernel timelines of C2 and its
# 8-GPU shard, the final kernel's phase stamps, and (optionally) interleaved A/B runs of knob variants.
#   gpu_round.sh <tag> ["<ab variants for c2>"]       outputs under gpurun_out/<tag>/
export TMPDIR=/tmp
TAG=${1:-round}; VARIANTS=${2:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -X faulthandler -m pytest tests -m gpu -v --tb=short -p no:cacheprovider --timeout 900 > $O/pytest_gpu_full.txt 2>&1; tail -15 $O/pytest_gpu_full.txt | cut -c1-200; grep -c PASSED $O/pytest_gpu_full.txt
for wl in c2 c4 c3shard c5mini c2f32 c2shard8; do
  timeout 400 python bench.py --workload $wl --cpu-seconds 0 2>$O/bench_$wl.err | tail -1 > $O/${wl}_bench.json
  python - <<PY
import json
r=json.loads(open('$O/${wl}_bench.json').read())
print('$wl', 'ms/step', r['ms_per_step'], 'qps', r['value'], 'pipelined', r['pipelined_qps'], 'launch_ms', r['roofline']['avg_launch_ms'], 'frac', r['roofline']['frac'], r['roofline']['bound'], 'fb', r['pipeline']['fallback_queries'], 'ingest GB/s', r['ingest']['gb_per_s'])
PY
done
for g in 2 4; do
  timeout 300 python bench.py --workload c2 --sharded-handle $g 2>$O/bench_sh$g.err | tail -1 > $O/c2_sharded${g}_bench.json; cut -c1-200 $O/c2_sharded${g}_bench.json
done
if [ -n "$VARIANTS" ]; then
  timeout 400 python scripts/ab.py --workload c2 --variants "$VARIANTS" --rounds 4 --steps 12 > $O/ab_c2.txt 2>$O/ab_c2.err; cat $O/ab_c2.txt
fi
CGV_TRACE=1 timeout 200 python scripts/ab.py --workload c2 --variants "default:" --rounds 1 --steps 5 --trace > $O/c2_final_kernel_phases.txt 2>$O/trace_c2.err; tail -12 $O/c2_final_kernel_phases.txt
for wl in c2 c2shard8; do
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$wl -o t -- python $R/bench.py --workload $wl --steps 6 --warmup 3 --settle-ms 0 --cpu-seconds 0 --pipelined-steps 0 > $O/tl_$wl.log 2>&1
  cd $R; python scripts/trace_timeline.py $O/tl_$wl/t_kernel_trace.csv > $O/${wl}_timeline.txt 2>&1; cat $O/${wl}_timeline.txt
done
