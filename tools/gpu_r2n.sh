#!/bin/bash
# round 2, call N: HIP-graph replay in the shared engine (RefineDet / PFPNetR / YOLOv2): tests + throughput graph vs eager
set -u
TAG=${1:-r02n}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_yolov2.py tests/test_gpu_refinedet_model.py tests/test_gpu_pfpnet_model.py -q -x ) > $O/t.log 2>&1; echo "exit $?" >> $O/t.log; tail -4 $O/t.log
for b in refinedet pfpnet yolov2; do
  timeout 600 python tools/${b}_bench.py bf16 32 10 > $O/${b}.log 2>&1; tail -1 $O/${b}.log
done
