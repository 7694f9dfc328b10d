#!/bin/bash
# round 2, call G: 512- and 192-pixel tile variants of the raster-run halo kernel: tests, same-box A/B per layer, bench A/B
set -u
TAG=${1:-r02g}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -k "conv" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
( timeout 600 python -m pytest tests/test_gpu_ssd300_b32.py -q -k "in_situ or gradient" ) > $O/b32.log 2>&1; echo "b32 exit $?" >> $O/b32.log; tail -3 $O/b32.log
timeout 300 python tools/conv_bench.py conv2_1,conv2_2,conv3_1,conv3_2,conv5_2 fwd,dgrad 20 0:0,0:402653184 > $O/convbench.log 2>&1; tail -12 $O/convbench.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_new_$i.log 2>&1; tail -1 $O/bench_new_$i.log | cut -c1-140
  timeout 300 python bench.py --kernel-dbg 402653184 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-140
done
