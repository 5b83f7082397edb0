#!/bin/bash
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4k
bash scripts/collect_profiles.sh r04y c2 > gpurun_out/r4k/collect.log 2>&1; tail -6 gpurun_out/r4k/collect.log | cut -c1-200
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4k/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4k/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
