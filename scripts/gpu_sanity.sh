#!/bin/bash
# final sanity of a tree: smoke, the sharded / C-client tests, the default bench line
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_sharded.py tests/test_c_client.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-1500
