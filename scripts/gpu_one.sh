#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 200 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py -m gpu -q -x -k "fp8 or nonfinite" -p no:cacheprovider 2>&1 | tail -4
