#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
