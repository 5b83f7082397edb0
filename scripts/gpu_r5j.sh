#!/bin/bash
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5j; O=gpurun_out/r5j
timeout 900 python -m pytest tests/test_c_client.py tests/test_gpu_configs.py::test_c2_full_size_anchored_and_strong_scaling_shards tests/test_c_abi.py -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 python bench.py --workload c5shard --cpu-seconds 0 2>/dev/null | tail -1 > $O/c5shard_bench.json; python -c "
import json; d=json.loads(open('$O/c5shard_bench.json').read()); r=d['roofline']; print('c5shard', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['rows_per_launch'], d['pipeline']['fallback_queries'], d['exact_check'])"
