#!/bin/bash
# round 3, call C: fused conv1_2 + pool1, recorded arg-max pool5 -- kernel tests, in-situ SSD300, bench, clock trace
set -u
TAG=${1:-r03c}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "pool2x2 or argmax or maxpool" ) > $O/kern.log 2>&1
tail -4 $O/kern.log
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -s -k "ssd300" ) > $O/insitu.log 2>&1
grep -E "passed|failed|in-situ|out of bound" $O/insitu.log | cut -c1-600 | tail
( time timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -x -q ) > $O/ssd.log 2>&1
tail -4 $O/ssd.log
timeout 300 python bench.py --conv-table $O/conv_table.txt > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
timeout 300 python tools/clock_trace.py $O/clock_trace.md 2000 > $O/clock_trace.log 2>&1
head -16 $O/clock_trace.md
