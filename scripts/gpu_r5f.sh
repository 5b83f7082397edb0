#!/bin/bash
# round 5: the final kernel's phase stamps on a corpus that lives in HBM (C2, 1.5 GB) and on one that fits the 256 MB Infinity
# Cache (c2shard8, 192 MB) - is the candidate-row gather bound by random 64-byte HBM reads?
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1 CGV_TRACE=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5f; O=gpurun_out/r5f
for wl in c2 c2shard8 small; do
  timeout 300 python scripts/ab.py --workload $wl --variants "new:" --rounds 1 --steps 8 --trace > $O/trace_$wl.txt 2>/dev/null; echo "== $wl"; tail -12 $O/trace_$wl.txt | cut -c1-250
done
