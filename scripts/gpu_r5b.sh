#!/bin/bash
# round 5, second call: the -m gpu suite (new: small-batch tests), then the latency side fields of c1 / c2 / c2shard8
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5b; O=gpurun_out/r5b
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt | cut -c1-300
b() { name=$1; shift; timeout 600 python bench.py "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "== $name rc=$?"; tail -2 $O/$name.err | cut -c1-300; }
b c1 --workload c1
b c2 --workload c2 --cpu-seconds 0
b c2shard8 --workload c2shard8 --cpu-seconds 0
b c4 --workload c4 --cpu-seconds 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'unreadable',e); continue
    r=d.get('roofline') or {}
    ph=d.get('pipelined_host') or {}
    print(f.split('/')[-1], 'ms/step', d.get('ms_per_step'), 'med', d.get('median_ms_per_step'), d.get('step_ms_percentiles'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'),
          'piped_host', ph.get('ms_per_batch'), ph.get('same_results_as_serial_step'), 'piped_dev', (d.get('pipelined') or {}).get('ms_per_batch'),
          'exact', (d.get('exact_check') or {}).get('ordered_match_rate'), 'err', d.get('error'))
    if d.get('latency'): print('   latency', json.dumps(d['latency'])[:1100])
PY
