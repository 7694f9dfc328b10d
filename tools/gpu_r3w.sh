#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -q -x -k "loss or mining or ssd300" ) > $O/k.log 2>&1
grep -E "passed|failed|rror" $O/k.log | head -5 | cut -c1-300
BCMD="python bench.py --steps 12 --warmup 14 --no-cpu-baseline --no-conv-events --eager"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/timeline.py $O/trace > $O/timeline.md 2>&1; head -2 $O/timeline.md
grep -E "softmax|nms_|ssd_loss|zero_fill" $O/timeline.md | cut -c1-110
rm -rf $O/trace
