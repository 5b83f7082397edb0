#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixed_row_scales or one_wave" -p no:cacheprovider 2>&1 | tail -6
