#!/bin/bash
set -u
TAG=${1:-r6b}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1800 python tools/gate_table.py ssd300,yolov3,fcos,centernet,yolov2,retinanet 300,600,1000 2 ) > $O/gate.log 2>&1
grep "^GATE" $O/gate.log | cut -c1-700
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
grep '^{' $O/bench.log > $O/bench_line.json; tail -3 $O/bench.log | cut -c1-1500
