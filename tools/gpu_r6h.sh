#!/bin/bash
set -u
TAG=${1:-r6h}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_insitu_configs.py tests/test_gpu_retinanet_model.py -q -x -k "x3 or retinanet or X3" ) > $O/x3_tests.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/x3_tests.log | cut -c1-300 | tail -5
for c in retinanet yolov3; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-conv-events 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['dtype'], d['value'], d['ms_per_step'])"
done
L=conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,conv7,conv8_2
timeout 300 python tools/conv_bench.py $L wgrad 30 0:0,0:1073741824 > $O/wgrad_v8_short.txt 2>&1; grep wgrad $O/wgrad_v8_short.txt | cut -c1-200
YCMD="python bench.py --config yolov3 --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ytrace -- $YCMD > $O/ytrace.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/ytrace 7 > $O/yolov3_f32x3_trace.md; rm -rf $O/ytrace; head -32 $O/yolov3_f32x3_trace.md | cut -c1-170
