#!/bin/bash
set -u
TAG=${1:-r6l}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python tools/gate_table.py yolov3,retinanet,ssd300,fcos,centernet 300,600 1 4 f32x1sim ) > $O/gate_x1sim.log 2>&1
grep "^GATE " $O/gate_x1sim.log | cut -c1-330
