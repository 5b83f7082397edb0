#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py tests/test_gpu_guarantee.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | cut -c1-400
