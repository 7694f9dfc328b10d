#!/bin/bash
# round 3, call F: first-layer wgrad kernel; whole kernel-test file; SSD300 suites; in-situ; bench with conv table
set -u
TAG=${1:-r03f}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "first_layer" ) > $O/kern_c8.log 2>&1
tail -15 $O/kern_c8.log | cut -c1-300
( time timeout 1200 python -m pytest tests/test_gpu_kernels.py -q ) > $O/kern.log 2>&1
tail -5 $O/kern.log | cut -c1-300
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -s -k "ssd300" ) > $O/insitu.log 2>&1
grep -E "passed|failed|in-situ|out of bound" $O/insitu.log | cut -c1-600 | tail -5
( time timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -x -q ) > $O/ssd.log 2>&1
tail -4 $O/ssd.log
timeout 300 python bench.py --conv-table $O/conv_table.txt --no-cpu-baseline > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
grep -E "C8 K64|H300" $O/conv_table.txt
