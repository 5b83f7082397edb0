#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x 2>&1 | tail -15
