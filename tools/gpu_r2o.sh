#!/bin/bash
# round 2, call O: 64 x 512 halo tiles for Cout <= 64 (conv2_1 dgrad): tests, per-layer and bench A/B
set -u
TAG=${1:-r02o}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "64_channel_tiles" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
( timeout 600 python -m pytest tests/test_gpu_ssd300_b32.py -q -k "in_situ or gradient" ) > $O/b32.log 2>&1; echo "b32 exit $?" >> $O/b32.log; tail -3 $O/b32.log
timeout 300 python tools/conv_bench.py conv2_1,conv2_2 dgrad 20 0:0,0:134217728 > $O/convbench.log 2>&1; tail -4 $O/convbench.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_new_$i.log 2>&1; tail -1 $O/bench_new_$i.log | cut -c1-140
  timeout 300 python bench.py --kernel-dbg 134217728 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-140
done
