#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -8
timeout 600 python bench.py --workload c5mini --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_c5mini.json
timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_c2.json
