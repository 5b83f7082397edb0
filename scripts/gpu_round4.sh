#!/bin/bash
# round 4, call B: the whole GPU suite (the pinned case back in-process, abort_bt hooked in), then the first measurements of
# the round's changes: fused sample + emit launch on/off (in-process A/B, measurement flavour), the join-free exchange
# (one-rank dry run), C2 / shard lines, the half-batch proxy for intra-call sub-batching.
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 900 --durations=12 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest_gpu.txt | cut -c1-240
grep -c "abort_bt: native" $O/pytest_gpu.txt
for wl in c2 c2shard8; do
  timeout 300 python scripts/ab.py --workload $wl --variants "classic:fuse_sample=0;fused:fuse_sample=1" --rounds 3 --steps 30 > $O/ab_fuse_$wl.txt 2>$O/ab_fuse_$wl.err; tail -3 $O/ab_fuse_$wl.txt | cut -c1-400
done
timeout 300 python bench.py --workload c2 --cpu-seconds 12 > $O/c2_bench.json 2>$O/c2_bench.err; python -c "
import json; d=json.loads(open('$O/c2_bench.json').read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['pipelined_qps'], d.get('cpu_baseline'), d.get('recall_at_10'))"
timeout 300 python bench.py --workload c2shard8 --cpu-seconds 0 > $O/c2shard8_bench.json 2>$O/c2shard8_bench.err; python -c "
import json; d=json.loads(open('$O/c2shard8_bench.json').read().strip().splitlines()[-1]); print('c2shard8', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['avg_launch_ms'], d['hbm_resident_serial']['ms_per_step'], d['pipelined']['ms_per_batch'])"
timeout 300 python bench.py --workload c2half --cpu-seconds 0 > $O/c2half_bench.json 2>$O/c2half_bench.err; python -c "
import json; d=json.loads(open('$O/c2half_bench.json').read().strip().splitlines()[-1]); print('c2half (512 queries)', d['ms_per_step'], d['hbm_resident_serial']['ms_per_step'], 'two in flight per batch', d['pipelined']['ms_per_batch'])"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist --cpu-seconds 0 > $O/c2_force_dist_bench.json 2>$O/fd.err; python -c "
import json; d=json.loads(open('$O/c2_force_dist_bench.json').read().strip().splitlines()[-1]); print('force-dist', d['ms_per_step'], d['multi_gpu']['exchange_ms'], d.get('recall_at_10'), d.get('ordered_match_rate'), d.get('score_bit_exact_rate'), d.get('check'))"
