#!/bin/bash
# round 3, call M: in-situ for the 8f.4 classes; tightened inference tolerances; bench lines of those classes
set -u
TAG=${1:-r03m}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_insitu_configs.py -q -s -k "ssd512 or refinedet or pfpnet or yolov2" ) > $O/insitu.log 2>&1
grep -E "passed|failed|in-situ|out of bound" $O/insitu.log | cut -c1-500 | tail -12
( time timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_retinanet_model.py -q -k "inference" ) > $O/infer.log 2>&1
tail -5 $O/infer.log | cut -c1-400
for t in ssd512 refinedet pfpnet yolov2; do timeout 300 python tools/${t}_bench.py bf16 32 10 > $O/bench_$t.log 2>&1; tail -1 $O/bench_$t.log | cut -c1-200; done
