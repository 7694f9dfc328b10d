#!/bin/bash
set -u
TAG=${1:-r6f}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python tools/gate_table.py ssd300,yolov3,fcos,centernet,yolov2,retinanet 300,600 1 4 ) > $O/gate4.log 2>&1
grep "^GATE " $O/gate4.log | cut -c1-330
( time timeout 900 python tools/gate_table.py ssd300,yolov3,fcos,centernet,yolov2 300 1 1 ) > $O/gate1.log 2>&1
grep "^GATE " $O/gate1.log | cut -c1-330
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bf16_gate.py --durations=5 ) > $O/gpu.log 2>&1
tail -12 $O/gpu.log | cut -c1-300
