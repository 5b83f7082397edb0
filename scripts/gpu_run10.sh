#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x 2>&1 | tail -6
echo "== bench c2"; timeout 900 python bench.py --steps 20 --warmup 3 --cpu-seconds 3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('qps', r['value'], 'step_ms', r['ms_per_step'], 'coarse_ms', r['roofline']['avg_launch_ms'], 'TF', r['roofline']['achieved'], 'recall', r.get('recall_at_10'), r.get('ordered_match_rate'), r.get('score_bit_exact_rate'), 'fb', r['pipeline']['fallback_queries'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -o c2 -- python $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
cd $R; python scripts/trace_timeline.py gpurun_out/prof_cur/c2_kernel_trace.csv
