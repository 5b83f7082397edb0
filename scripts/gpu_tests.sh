#!/bin/bash
# the -m gpu suite, verbose, un-captured (-s: a message the HIP runtime / glibc prints before an abort is in the log), with the
# full log kept (gpurun_out/pytest_gpu_full.txt): a crash names its test
export TMPDIR=/tmp
export LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 1700 python -X faulthandler -m pytest tests -m gpu -v -s --tb=short -p no:cacheprovider --timeout 900 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu_full.txt 2>&1
echo "rc=$?"; tail -25 gpurun_out/pytest_gpu_full.txt | cut -c1-220; echo "passed: $(grep -c PASSED gpurun_out/pytest_gpu_full.txt)"
