#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "nms or mining or loss or softmax" ) > $O/k.log 2>&1
grep -E "passed|failed|rror" $O/k.log | head -5 | cut -c1-300
timeout 900 python tools/ab_bench.py base= same=cfg:verbose=0 twojoins=cfg:front_join_via_tail=0 --rounds 10 --block 25 > $O/ab.md 2>&1; cat $O/ab.md
BCMD="python bench.py --steps 12 --warmup 14 --no-cpu-baseline --no-conv-events --eager"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/timeline.py $O/trace > $O/timeline.md 2>&1; head -2 $O/timeline.md
grep -E "softmax|nms_|ssd_loss" $O/timeline.md | cut -c1-110
rm -rf $O/trace
