#!/bin/bash
# Hunt for the intermittent suite abort: the in-process stress under several allocator / logging modes,
# then the two test files in suite order a few times. Everything un-captured; abort_bt names the native stack.
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/abort; cd $R
O=gpurun_out/abort
S=${STRESS_S:-50}
echo "== stress plain (both forms)";      timeout 300 python scripts/stress_suite_abort.py $S both > $O/stress_plain.txt 2>&1; echo "rc=$?"; tail -3 $O/stress_plain.txt
echo "== stress MALLOC_CHECK_/PERTURB_";  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout 300 python scripts/stress_suite_abort.py $S both > $O/stress_mcheck.txt 2>&1; echo "rc=$?"; tail -3 $O/stress_mcheck.txt
echo "== stress AMD_LOG_LEVEL=1, heap";   AMD_LOG_LEVEL=1 timeout 300 python scripts/stress_suite_abort.py $S heap > $O/stress_log1.txt 2>&1; echo "rc=$?"; tail -3 $O/stress_log1.txt
for i in 1 2 3; do
  echo "== pytest robustness+sharded (in-process pinned case) run $i"
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_robustness.py tests/test_gpu_sharded.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 > $O/pytest_pair_$i.txt 2>&1
  echo "rc=$?"; tail -4 $O/pytest_pair_$i.txt | cut -c1-200
done
grep -l "abort_bt: native" $O/*.txt
