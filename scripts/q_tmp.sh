#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LIBC_FATAL_STDERR_=1
for i in 1 2 3; do
  echo "== plain run $i"
  timeout 300 python -X faulthandler -m pytest tests/test_gpu_robustness.py tests/test_gpu_sharded.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 200 -k "pinned_host or id_map or sharded_handle_matches" 2>&1 | tail -12
done
export MALLOC_CHECK_=3
for i in 1 2; do
  echo "== MALLOC_CHECK_ run $i"
  timeout 300 python -X faulthandler -m pytest tests/test_gpu_robustness.py tests/test_gpu_sharded.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 200 -k "pinned_host or id_map or sharded_handle_matches" 2>&1 | tail -12
done
