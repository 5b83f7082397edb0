#!/bin/bash
# round 5, third call: threshold ladder A/B (staged plan vs ladder inside the planned launches vs ladder + ONE emitting launch),
# kernel timelines of single-query calls
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5c; O=gpurun_out/r5c
V="staged:;lad1:ladder=1;lad2:ladder=2;staged2:"
for wl in c2 c2shard8 c3shard c4; do
  st=15; [ $wl = c3shard ] && st=6
  timeout 600 python scripts/ab.py --workload $wl --variants "$V" --rounds 3 --steps $st > $O/ab_$wl.txt 2> $O/ab_$wl.err; echo "== ab $wl rc=$?"; cat $O/ab_$wl.txt | cut -c1-330; tail -2 $O/ab_$wl.err | cut -c1-200
done
cd /tmp
for wl in c2shard8 c1 c2; do
  it=300; [ $wl = c2 ] && it=60
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O -o lat_$wl -- python $R/scripts/lat_loop.py --workload $wl --nq 1 --iters $it > $R/$O/lat_$wl.log 2>&1
  echo "== lat $wl"; tail -1 $R/$O/lat_$wl.log; head -12 $R/$O/lat_${wl}_kernel_stats.csv | cut -d, -f1-8 | cut -c1-200
done
cd $R; rm -f $O/*_agent_info.csv $O/*_domain_stats.csv; ls -la $O | head -30
