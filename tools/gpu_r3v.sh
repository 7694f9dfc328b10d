#!/bin/bash
set -u
TAG=${1:-r03v}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_yolov3.py tests/test_gpu_insitu_configs.py -q -x -k "yolov3" ) > $O/y.log 2>&1
grep -E "passed|failed|rror" $O/y.log | head -5 | cut -c1-300
for i in 1 2 3; do for v in wgrad_stream=0 wgrad_stream=1; do
  echo -n "$v: "; timeout 300 python bench.py --config yolov3 --steps 30 --warmup 5 --no-cpu-baseline --model-cfg $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
