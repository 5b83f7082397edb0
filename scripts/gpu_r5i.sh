#!/bin/bash
# round 5: the default bench line (and C4 / c3shard / c5mini) with the stamped PMC passes of THIS source in profiles/ - `roofline.traffic`,
# `limits.clock_ghz`, the bound decided at the measured clock
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5i; O=gpurun_out/r5i
python bench.py 2>/dev/null | tail -1 > $O/default_bench.json; cut -c1-300 $O/default_bench.json
for wl in c4 c3shard c5mini; do python bench.py --workload $wl --cpu-seconds 0 2>/dev/null | tail -1 > $O/${wl}_bench.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5i/*.json')):
    d=json.loads(open(f).read()); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['bound'], r['frac'], r.get('traffic'), json.dumps(r.get('limits')))
    if 'cpu_baseline' in d and d['cpu_baseline']: print('  cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['thread_sweep_ms_per_query'], d['cpu_baseline']['score_ms'], d['cpu_baseline']['sort_ms'])
PY
