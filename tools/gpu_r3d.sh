#!/bin/bash
# round 3, call D: same-box A/B of the fused pool; step-count sweep of the bench (is the 20-step figure a transient?)
set -u
TAG=${1:-r03d}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-conv-events"
for rep in 1 2; do
  for k in 20 200; do
    timeout 200 $B --steps $k --warmup 5 > $O/fused_${k}_$rep.log 2>&1
    timeout 200 $B --steps $k --warmup 5 --no-fuse-pool > $O/unfused_${k}_$rep.log 2>&1
  done
done
for k in 20 50 100 500; do timeout 200 $B --steps $k --warmup 5 > $O/sweep_$k.log 2>&1; done
timeout 200 $B --steps 20 --warmup 60 > $O/sweep_20_w60.log 2>&1
timeout 200 $B --steps 20 --warmup 5 --eager > $O/eager_20.log 2>&1
timeout 200 $B --steps 200 --warmup 5 --eager > $O/eager_200.log 2>&1
for f in $O/*.log; do echo -n "$(basename $f) "; tail -1 $f | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log | cut -c1-200
