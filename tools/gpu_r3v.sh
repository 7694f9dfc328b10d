#!/bin/bash
set -u
TAG=${1:-r03v}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do for v in 4:1024 4:1408 4:5632 4:22000; do
  echo -n "$v: "; timeout 300 python bench.py --config yolov3 --steps 30 --warmup 5 --no-cpu-baseline --debug-set $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
