#!/bin/bash
# round 6: (1) per-kernel timeline of the emitting-sample A/B (rocprofv3 kernel trace of scripts/ab.py, one variant per run);
# (2) the "query operand for free" timing probes (CGV_ABLATE=1024 / 2048 / 4096) against the production form on C2 and the C3 shard.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -f $O/r6n_*.txt
for w in c2 c2shard8; do
  for v in "old:sample_emit=0" "new:"; do
    tag=${v%%:*}
    rm -rf /tmp/prof_${w}_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${w}_$tag -o p -- python $R/scripts/ab.py --workload $w --variants "$v" --rounds 1 --steps 40 > /dev/null 2> $O/r6n_prof_${w}_$tag.err
    f=$(find /tmp/prof_${w}_$tag -name "*kernel_stats.csv" | head -1)
    t=$(find /tmp/prof_${w}_$tag -name "*kernel_trace.csv" | head -1)
    echo "== $w $tag" >> $O/r6n_kernel_stats.txt
    head -12 "$f" | cut -c1-220 >> $O/r6n_kernel_stats.txt
    python $R/scripts/trace_timeline.py "$t" >> $O/r6n_kernel_stats.txt 2>&1
  done
done
export CGV_LIB_PATH=$R/codegraph-rust_amd/lib/libcgvec_hip_ablate.so
cd $R
for wl in c2 c3shard; do
for abl in 0 1024 2048 4096 0 1024 2048 4096; do
CGV_ABLATE=$abl timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --cpu-seconds 0 --latency 0 --coalesced-threads 0 --pipelined-steps 0 --check-queries 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$wl mask $abl','coarse_ms',r['roofline']['avg_launch_ms'])" >> $O/r6n_bfree.txt
done
done
cat $O/r6n_kernel_stats.txt $O/r6n_bfree.txt
