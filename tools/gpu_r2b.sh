#!/bin/bash
# round 2, call B: conv kernel tests (new wide halo variant), b32 parity tests, conv micro-bench of conv2_x, bench
set -u
TAG=${1:-r02b}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
( timeout 900 python -m pytest tests/test_gpu_ssd300_b32.py -q -s ) > $O/b32.log 2>&1; echo "b32 exit $?" >> $O/b32.log
grep -E "passed|failed|exit|bf16 scores|bf16 detections|per level" $O/b32.log | tail -8
timeout 300 python tools/conv_bench.py conv2_1,conv2_2,conv3_1 fwd,dgrad 20 0:0,0:67108864 > $O/convbench.log 2>&1; tail -12 $O/convbench.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-table $O/conv_table.txt > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
