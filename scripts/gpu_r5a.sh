#!/bin/bash
# round 5, first call: the -m gpu suite (new: pipelined join-free step, host-in/host-out batches in flight, the world > 1 bench
# program on one GPU over gloo), then bench lines: C2 / c2shard8 / c1 plain (value, pipelined_host, latency, exact_check), the
# one-rank RCCL dry run at depth 2 and 3, the 8-rank gloo dry run of C2
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5a; O=gpurun_out/r5a
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt | cut -c1-300
b() { name=$1; shift; timeout 600 python bench.py "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "== $name rc=$?"; cut -c1-400 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
b c2 --workload c2
b c2shard8 --workload c2shard8 --cpu-seconds 0
b c1 --workload c1
b c2_d2 --workload c2 --cpu-seconds 0 --depth 2 --latency 0 --steps 100
b c2shard8_d2 --workload c2shard8 --cpu-seconds 0 --depth 2 --latency 0
d() { name=$1; n=$2; shift; shift; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $n "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; echo "== $name"; cut -c1-400 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
d fd_c2 1 --force-dist --workload c2
d fd_c2shard8 1 --force-dist --workload c2shard8
d fd_c2shard8_d2 1 --force-dist --workload c2shard8 --depth 2
d gloo8_c2 8 --dist-backend gloo --workload c2 --steps 20 --warmup 3 --pipelined-steps 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'unreadable',e); continue
    r=d.get('roofline') or {}
    ph=d.get('pipelined_host') or {}
    print(f.split('/')[-1], 'ms/step', d.get('ms_per_step'), 'med', d.get('median_ms_per_step'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'),
          'piped_host', ph.get('ms_per_batch'), ph.get('batches_in_flight'), ph.get('same_results_as_serial_step'),
          'piped_dev', (d.get('pipelined') or {}).get('ms_per_batch'), 'resident', (d.get('hbm_resident_serial') or {}).get('ms_per_step'),
          'recall', d.get('recall_at_10'), 'exact', (d.get('exact_check') or {}).get('ordered_match_rate'), (d.get('exact_check') or {}).get('queries'),
          'err', d.get('error'))
    if d.get('latency'): print('   latency', json.dumps(d['latency'])[:900])
PY
