#!/bin/bash
# round-3 experiment e: sharded begin/end + new tests, bench lines of every workload on the round-3 pipeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_c_client.py tests/test_gpu_configs.py::test_c2_full_size_anchored_and_strong_scaling_shards tests/test_gpu_store.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -25 > $O/pytest_new.txt; cat $O/pytest_new.txt
for wl in c2 c4 c3shard c5mini c2f32 c2shard8; do
  timeout 400 python bench.py --workload $wl --cpu-seconds 0 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json
  python - <<PY
import json
r=json.loads(open('$O/bench_$wl.json').read())
print('$wl', 'ms/step', r['ms_per_step'], 'median', r['median_ms_per_step'], 'qps', r['value'], 'pipelined', r['pipelined_qps'], 'launch_ms', r['roofline']['avg_launch_ms'], 'frac', r['roofline']['frac'], r['roofline']['bound'], 'fb', r['pipeline']['fallback_queries'], 'ingest', r['ingest']['gb_per_s'])
PY
done
timeout 300 python bench.py --workload c2 --cpu-seconds 0 --settle-ms 0 2>/dev/null | tail -1 > $O/bench_c2_nosettle.json
python -c "
import json; r=json.loads(open('$O/bench_c2_nosettle.json').read()); print('c2 no settle: ms/step', r['ms_per_step'], 'launch', r['roofline']['avg_launch_ms'])"
for g in 2 4; do
  timeout 300 python bench.py --workload c2 --sharded-handle $g 2>$O/bench_sh$g.err | tail -1 > $O/bench_c2_sharded$g.json; cut -c1-900 $O/bench_c2_sharded$g.json
done
