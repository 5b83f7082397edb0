#!/bin/bash
# round-2 call A: the whole -m gpu suite (new sharded / guarantee / config tests included), then the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=25 -s ) > gpurun_out/r2a_pytest.log 2>&1
tail -40 gpurun_out/r2a_pytest.log
timeout 300 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; tail -c 1500 gpurun_out/r2a_bench.json; tail -3 gpurun_out/r2a_bench.err
