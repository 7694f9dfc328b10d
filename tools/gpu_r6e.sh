#!/bin/bash
set -u
TAG=${1:-r6e}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad or determin" ) > $O/kernels.log 2>&1; grep -E "passed|failed" $O/kernels.log | cut -c1-300
L=conv1_1,conv1_2,conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,conv7,conv8_1,conv8_2,pred1,pred2,pred3,conv9_1,conv9_2
ODTK_WG=0 timeout 300 python tools/conv_bench.py $L wgrad 30 0 > $O/wgrad_atomics.txt 2>&1
ODTK_WG=1 timeout 300 python tools/conv_bench.py $L wgrad 30 0 > $O/wgrad_det.txt 2>&1
paste <(grep -E "wgrad" $O/wgrad_atomics.txt | cut -c1-64) <(grep -E "wgrad" $O/wgrad_det.txt | cut -c18-95) | head -40
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 5:0 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('atomics', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 5:1 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('determ ', d['value'], d['ms_per_step'])"
done
cd /tmp; ODTK_WG=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/dettrace -- python $R/tools/conv_bench.py conv3_2,conv4_2,conv1_2,conv2_2 wgrad 20 0 > $O/dettrace.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/dettrace 1 > $O/det_wgrad_trace.md 2>&1; rm -rf $O/dettrace; head -10 $O/det_wgrad_trace.md | cut -c1-200
for c in yolov3 fcos; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-conv-events --debug-set 5:0 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c atomics', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-conv-events --debug-set 5:1 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c determ ', d['value'], d['ms_per_step'])"
done
