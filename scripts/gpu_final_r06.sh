#!/bin/bash
# round 6's round-end collection: gpu_final.sh (suite in the driver's form, stamped profiles of c2 / c4 / c3shard / c5mini, shard and
# sharded-handle lines, one-rank RCCL dry runs) + c1, the driver-form default line, the 8-rank gloo dry run of the rank program,
# the one-rank RCCL line with the sharded query exchange forced.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; cd $R
bash scripts/gpu_final.sh $TAG 2>&1 | tail -120
O=gpurun_out/profiles_$TAG
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2_driverform_bench.json; cut -c1-200 $O/c2_driverform_bench.json
timeout 600 python bench.py --workload c1 2>/dev/null | tail -1 > $O/c1_bench.json; cut -c1-200 $O/c1_bench.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 8 --dist-backend gloo --cpu-seconds 0 --steps 10 --warmup 2 --pipelined-steps 9 2>/dev/null | grep '^{' | tail -1 > $O/c2_gloo8_dryrun_bench.json; cut -c1-200 $O/c2_gloo8_dryrun_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 1 --force-dist --workload c2shard8 --cpu-seconds 0 --query-exchange sharded 2>/dev/null | grep '^{' | tail -1 > $O/c2shard8_force_dist_sharded_queries_bench.json
python - <<PY
import json
for f in ("c2shard8_force_dist_bench","c2shard8_force_dist_sharded_queries_bench","c2_force_dist_bench"):
    try:
        d=[json.loads(l) for l in open("$O/"+f+".json") if l.startswith("{")][-1]
    except Exception as e:
        print(f,"unreadable",e); continue
    mg=d["multi_gpu"]; print(f, d["value"], d["ms_per_step"], (d.get("pipelined_host") or {}).get("ms_per_batch"), mg["query_exchange"]["replicated_ms"], mg["query_exchange"]["sharded_ms"], mg["query_exchange"]["timed_steps_use"], {k:v["max"] for k,v in mg["per_rank_phases_us"].items()}, mg["all_gather_latency_us"]["packed_records"], d.get("recall_at_10"))
PY
