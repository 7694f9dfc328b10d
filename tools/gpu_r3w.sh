#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "nms or mining or loss or softmax" ) > $O/k.log 2>&1
grep -E "passed|failed|rror" $O/k.log | head -5 | cut -c1-300
for i in 1 2; do timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-conv-events 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('100 steps', d['value'], d['ms_per_step'])"; done
