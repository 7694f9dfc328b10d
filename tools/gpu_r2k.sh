#!/bin/bash
# round 2, call K: deterministic split-reduce of the filter gradients: tests, per-layer and bench A/B against the atomics
set -u
TAG=${1:-r02k}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -k "conv or wgrad" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
( timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py tests/test_gpu_fcos_model.py -q -x ) > $O/model.log 2>&1; echo "model exit $?" >> $O/model.log; tail -3 $O/model.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_new_$i.log 2>&1; tail -1 $O/bench_new_$i.log | cut -c1-140
  timeout 300 python bench.py --debug-set 5:1 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-140
done
