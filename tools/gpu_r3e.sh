#!/bin/bash
# round 3, call E: launch-mode A/B on one box (graph replay vs eager, head stream on/off), interleaved repetitions
set -u
TAG=${1:-r03e}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-conv-events --warmup 5"
for rep in 1 2 3; do
  timeout 200 $B --steps 20 > $O/graph_20_$rep.log 2>&1
  timeout 200 $B --steps 20 --eager > $O/eager_20_$rep.log 2>&1
  timeout 200 $B --steps 20 --eager --no-tail-stream > $O/eager_notail_20_$rep.log 2>&1
  timeout 200 $B --steps 20 --no-tail-stream > $O/graph_notail_20_$rep.log 2>&1
  timeout 200 $B --steps 20 --auto-launch > $O/auto_20_$rep.log 2>&1
done
timeout 200 $B --steps 300 > $O/graph_300.log 2>&1
timeout 200 $B --steps 300 --eager > $O/eager_300.log 2>&1
for f in $O/*.log; do echo -n "$(basename $f) "; tail -1 $f | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config'].get('launch'), j['config'].get('launch_mode_calibration'))"; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log | cut -c1-200
