#!/bin/bash
# final call of the round: PMC pass of the C2 main launch on the final sources (the stamp bench.py checks), the whole GPU suite, smoke
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4k
bash scripts/collect_profiles.sh r04y c2 > gpurun_out/r4k/collect.log 2>&1; tail -6 gpurun_out/r4k/collect.log | cut -c1-200
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4k/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4k/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r4k/default_bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r4k/default_bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['recall_at_10'])"
