#!/bin/bash
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4g/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4g/pytest_gpu.txt | cut -c1-200; grep -c "abort_bt: native" gpurun_out/r4g/pytest_gpu.txt
bash scripts/collect_profiles.sh r04x c2 > gpurun_out/r4g/collect.log 2>&1; tail -8 gpurun_out/r4g/collect.log | cut -c1-200
timeout 300 python bench.py --workload c2 --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/r4g/c2_bench_after_pmc.json; python -c "
import json; d=json.loads(open('gpurun_out/r4g/c2_bench_after_pmc.json').read()); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_note'], d['roofline']['frac'])"
timeout 900 python bench.py --workload c5shard --cpu-seconds 0 --pipelined-steps 4 2>gpurun_out/r4g/c5shard.err | tail -1 > gpurun_out/r4g/c5shard_bench.json; cut -c1-400 gpurun_out/r4g/c5shard_bench.json
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4g/pytest_gpu_2.txt 2>&1; echo "pytest#2 rc=$?"; tail -2 gpurun_out/r4g/pytest_gpu_2.txt | cut -c1-200; grep -c "abort_bt: native" gpurun_out/r4g/pytest_gpu_2.txt
