#!/bin/bash
set -u
TAG=${1:-r6a}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bf16_gate.py --durations=8 ) > $O/gpu.log 2>&1
tail -15 $O/gpu.log | cut -c1-300
( time timeout 1500 python tools/gate_table.py ssd300,yolov3,fcos,centernet,yolov2,retinanet 300,600,1000 2 ) > $O/gate.log 2>&1
grep "^GATE" $O/gate.log | cut -c1-600
