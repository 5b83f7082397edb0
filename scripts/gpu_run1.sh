#!/bin/bash
# First GPU session: diagnostics, parity tests, smoke, bench, rocprof. Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2
echo "== diag"; timeout 300 python scripts/diag_coarse.py 2>&1 | tee gpurun_out/diag.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench small"; timeout 300 python bench.py --workload small --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -3 | tee gpurun_out/bench_small.log
echo "== bench c2"; timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_c2.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -3)
ls -R gpurun_out | head -40
