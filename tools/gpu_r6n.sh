#!/bin/bash
set -u
TAG=${1:-r6o}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_insitu_configs.py tests/test_gpu_retinanet_model.py tests/test_gpu_yolov3.py tests/test_gpu_refinedet_model.py tests/test_gpu_lhrcnn.py -q -x -k "x3 or retinanet or X3 or yolov3 or refinedet or lhrcnn" ) > $O/x3_tests.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/x3_tests.log | cut -c1-300 | tail -5
for i in 1 2; do
for b in 0 8192; do
  timeout 300 python bench.py --config yolov3 --steps 10 --warmup 3 --no-cpu-baseline --no-conv-events --debug-set 6:$b 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('yolov3 dbg2=$b', d['dtype'], d['value'], d['ms_per_step'])"
done; done
for b in 0 8192; do
timeout 300 python bench.py --config retinanet --steps 10 --warmup 3 --no-cpu-baseline --no-conv-events --debug-set 6:$b 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('retinanet dbg2=$b', d['dtype'], d['value'], d['ms_per_step'])"
done
timeout 300 python tools/refinedet_bench.py f32x3 32 5 2>&1 | grep "images/s"
for i in 1 2 3; do timeout 300 python tools/yolov2_bench.py bf16 32 10 2>&1 | grep "images/s"; done
