#!/bin/bash
# round 2, call H: four-wave filter-gradient kernel (128 x 128 wave tiles): tests, same-box A/B per layer, bench A/B
set -u
TAG=${1:-r02h}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -k "conv" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
timeout 300 python tools/conv_bench.py conv3_2,conv4_1,conv4_2,conv5_2,conv6 wgrad 20 0:0,0:536870912 > $O/convbench.log 2>&1; tail -14 $O/convbench.log
if [ "${2:-}" = "bench" ]; then
( timeout 600 python -m pytest tests/test_gpu_ssd300_b32.py -q -k "in_situ or gradient" ) > $O/b32.log 2>&1; echo "b32 exit $?" >> $O/b32.log; tail -3 $O/b32.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_new_$i.log 2>&1; tail -1 $O/bench_new_$i.log | cut -c1-140
  timeout 300 python bench.py --kernel-dbg 536870912 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-140
done
fi
