#!/bin/bash
# round 6: emitting sample, final form - suite subset, in-process A/B, kernel timelines of the new form
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_guarantee.py tests/test_gpu_configs.py tests/test_gpu_small_batch.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -30 > $O/r6p_tests.txt
rm -f $O/r6p_ab.txt $O/r6p_timeline.txt
for w in c2 c2shard8 c4; do python scripts/ab.py --workload $w --variants "old:sample_emit=0;new:" --rounds 4 --steps 30 >> $O/r6p_ab.txt 2>> $O/r6p_ab.err; done
cd /tmp && export TMPDIR=/tmp
for w in c2 c2shard8; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o p -- python $R/scripts/ab.py --workload $w --variants "new:" --rounds 1 --steps 40 > /dev/null 2>> $O/r6p_ab.err
  echo "== $w new" >> $O/r6p_timeline.txt
  python $R/scripts/trace_timeline.py $(find /tmp/prof_$w -name "*kernel_trace.csv" | head -1) >> $O/r6p_timeline.txt 2>&1
done
cat $O/r6p_tests.txt $O/r6p_ab.txt $O/r6p_timeline.txt
