#!/bin/bash
set -u
TAG=${1:-r6c}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad or determin or l2norm" ) > $O/kernels.log 2>&1; tail -3 $O/kernels.log | cut -c1-300
( time timeout 1200 python tools/step_determinism.py ssd300,yolov3,fcos,centernet,yolov2,retinanet f32,bf16 3 ) > $O/det.log 2>&1
timeout 300 python tools/step_determinism.py retinanet f32x3 3 >> $O/det.log 2>&1
grep -E "^DET|^  " $O/det.log | cut -c1-420
L=conv1_1,conv1_2,conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,conv7,conv8_1,conv8_2,pred1,pred2,pred3,conv9_1,conv9_2
ODTK_WG=0 timeout 300 python tools/conv_bench.py $L wgrad 30 0 > $O/wgrad_atomics.txt 2>&1
ODTK_WG=1 timeout 300 python tools/conv_bench.py $L wgrad 30 0 > $O/wgrad_det.txt 2>&1
paste <(grep -E "wgrad" $O/wgrad_atomics.txt | cut -c1-90) <(grep -E "wgrad" $O/wgrad_det.txt | cut -c1-90) | head -40
for i in 1 2; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 5:1 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('determ ', d['value'], d['ms_per_step'])"
done
