#!/bin/bash
# round 2, call J: single-launch batch norm of the small maps: tests, bench A/B (+ matching on the tail stream)
set -u
TAG=${1:-r02j}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "batchnorm" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -3 $O/kern.log
( timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py tests/test_gpu_yolov3.py -q -x ) > $O/model.log 2>&1; echo "model exit $?" >> $O/model.log; tail -3 $O/model.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_new_$i.log 2>&1; tail -1 $O/bench_new_$i.log | cut -c1-140
  timeout 300 python bench.py --debug-set 4:0 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-140
  timeout 300 python bench.py --match-stream --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_match_$i.log 2>&1; tail -1 $O/bench_match_$i.log | cut -c1-140
done
