#!/bin/bash
# round 3, call N: bf16 gate measurement for RefineDet320 / PFPNetR / YOLOv2; in-situ rerun of the failing PFPNetR cases
set -u
TAG=${1:-r03n}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for m in refinedet pfpnet yolov2; do
  timeout 600 python tools/bf16_after_training.py $m 300 4 > $O/bf16_$m.log 2>&1
  tail -5 $O/bf16_$m.log | cut -c1-420
done
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -k "pfpnet" ) > $O/insitu.log 2>&1
tail -3 $O/insitu.log | cut -c1-300
