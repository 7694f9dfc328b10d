#!/bin/bash
set -u
TAG=${1:-r6d}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/wreg_equal.py > $O/wreg_equal.txt 2>&1; cat $O/wreg_equal.txt | tail -6 | cut -c1-250
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "filter_from_registers or wgrad_split_reduce" ) > $O/kernels.log 2>&1; tail -3 $O/kernels.log | cut -c1-300
L=conv4_1,conv4_2,conv5_2,conv6,pred1,pred2
ODTK_DBG2=0 timeout 300 python tools/conv_bench.py $L fwd,dgrad 30 0 > $O/v6_base.txt 2>&1
ODTK_DBG2=65536 timeout 300 python tools/conv_bench.py $L fwd,dgrad 30 0 > $O/v6_wreg.txt 2>&1
paste <(grep -E "fwd|dgrad" $O/v6_base.txt | cut -c1-95) <(grep -E "fwd|dgrad" $O/v6_wreg.txt | cut -c18-95) | head -30
for i in 1 2; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 6:65536 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wreg   ', d['value'], d['ms_per_step'])"
done
cd /tmp; ODTK_WG=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/dettrace -- python $R/tools/conv_bench.py conv3_2,conv4_2,conv1_2 wgrad 20 0 > $O/dettrace.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/dettrace 1 > $O/det_wgrad_trace.md 2>&1; rm -rf $O/dettrace; head -12 $O/det_wgrad_trace.md | cut -c1-200
timeout 300 python tools/step_determinism.py retinanet f32,f32x3 3 > $O/det_retina.log 2>&1; grep -E "^DET|^  " $O/det_retina.log | cut -c1-300
( time timeout 1200 python tools/gate_table.py ssd300,yolov3,fcos,centernet,yolov2,retinanet 300 2 ) > $O/gate.log 2>&1
grep "^GATE " $O/gate.log | cut -c1-330
