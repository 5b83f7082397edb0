#!/bin/bash
# round-end collection: the whole GPU suite (driver form), profiles (kernel stats + timeline + 5 PMC passes, source-stamped) for
# four workloads, plain bench lines for the rest, the one-handle sharded path (C2 over 2 / 4 shards, C3 WHOLE over 8), the
# one-rank dry run of the multi-GPU bench
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.txt | cut -c1-200
grep -c "abort_bt: native" gpurun_out/pytest_gpu_$TAG.txt
bash scripts/collect_profiles.sh $TAG c2 c4 c3shard c5mini > gpurun_out/collect_$TAG.log 2>&1
tail -70 gpurun_out/collect_$TAG.log | cut -c1-220
O=gpurun_out/profiles_$TAG
for wl in c2shard8 c2f32; do
  timeout 900 python bench.py --workload $wl --cpu-seconds 0 2>/dev/null | tail -1 > $O/${wl}_bench.json
  cut -c1-200 $O/${wl}_bench.json
done
for g in 2 4; do
  timeout 300 python bench.py --workload c2 --sharded-handle $g --steps 50 2>/dev/null | tail -1 > $O/c2_sharded${g}_bench.json
done
timeout 600 python bench.py --workload c3 --sharded-handle 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/c3_sharded8_bench.json; cut -c1-300 $O/c3_sharded8_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist 2>/dev/null | grep '^{' | tail -1 > $O/c2_force_dist_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --force-dist --workload c2shard8 2>/dev/null | grep '^{' | tail -1 > $O/c2shard8_force_dist_bench.json
# the world > 1 rank program on this ONE GPU (records and control over gloo): 8 and 2 ranks of C2 - control flow and parity, not speed
for w in 8 2; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29550 + w)) bench.py --gpus $w --dist-backend gloo --steps 30 --warmup 3 --pipelined-steps 30 2>/dev/null | grep '^{' | tail -1 > $O/c2_gloo${w}_dryrun_bench.json
done
timeout 300 python bench.py --workload c1 2>/dev/null | tail -1 > $O/c1_bench.json; cut -c1-200 $O/c1_bench.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/profiles_'"$TAG"'/*_bench.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step', d.get('ms_per_step'), 'median', d.get('median_ms_per_step'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'),
          'pipelined_host', (d.get('pipelined_host') or {}).get('ms_per_batch'), 'pipelined', (d.get('pipelined') or {}).get('ms_per_batch'), 'exact_check', (d.get('exact_check') or {}).get('ordered_match_rate'), 'resident', (d.get('hbm_resident_serial') or {}).get('ms_per_step'),
          'xch', (d.get('multi_gpu') or {}).get('exchange_ms'), d.get('last_exchange_ms'), 'recall', d.get('recall_at_10'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
# the small-batch path: kernel timeline of single-query calls + counters of its one coarse launch (COARSE_TOP2) on C2
cd /tmp
for wl in c2shard8 c1 c2; do
  it=300; [ $wl = c2 ] && it=60
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O -o lat_$wl -- python $R/scripts/lat_loop.py --workload $wl --nq 1 --iters $it > $R/$O/lat_$wl.log 2>&1
  tail -1 $R/$O/lat_$wl.log; rm -f $R/$O/lat_${wl}_kernel_trace.csv
done
t2() { name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "coarse_kernel" --output-format csv -d $R/$O/top2 -o $name -- python $R/scripts/lat_loop.py --workload c2 --nq 1 --iters 5 > $R/$O/top2_$name.log 2>&1; }
t2 p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
t2 p2 FETCH_SIZE TCC_HIT_sum
t2 p3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
cd $R; python scripts/pmc_summary.py $O/top2 > $O/c2_top2_pmc_summary.txt 2>&1; tail -8 $O/c2_top2_pmc_summary.txt
rm -rf $O/top2 $O/*_agent_info.csv $O/*_domain_stats.csv $O/top2_*.log
