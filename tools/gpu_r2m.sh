#!/bin/bash
# round 2, call M: YOLOv2: box-side kernels, model parity, throughput; regression of the engine's other users
set -u
TAG=${1:-r02m}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_yolov2.py -q -x -s ) > $O/y2.log 2>&1; echo "y2 exit $?" >> $O/y2.log; tail -15 $O/y2.log
( timeout 900 python -m pytest tests/test_gpu_refinedet_model.py tests/test_gpu_pfpnet_model.py -q -x ) > $O/rd.log 2>&1; echo "rd exit $?" >> $O/rd.log; tail -3 $O/rd.log
timeout 600 python tools/yolov2_bench.py > $O/bench.log 2>&1; tail -3 $O/bench.log
