#!/bin/bash
# round 3, call K: remaining GPU test files; YOLOv3 launch mode + BN variants; YOLOv3 kernel trace
set -u
TAG=${1:-r03k}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ssd512.py tests/test_gpu_tf_known_answers.py tests/test_gpu_yolov2.py tests/test_gpu_yolov3.py tests/test_gpu_bf16_gate.py -q ) > $O/pytest_rest.log 2>&1
tail -6 $O/pytest_rest.log | cut -c1-400
B="python bench.py --config yolov3 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events"
for rep in 1 2; do
  timeout 200 $B > $O/y_eager_$rep.log 2>&1
  timeout 200 $B --graph > $O/y_graph_$rep.log 2>&1
  timeout 200 $B --debug-set 4:-2 > $O/y_eager_two_$rep.log 2>&1
  timeout 200 $B --debug-set 4:1400 > $O/y_eager_rows1400_$rep.log 2>&1
  timeout 200 $B --debug-set 4:6000 > $O/y_eager_rows6000_$rep.log 2>&1
done
for f in $O/y_*.log; do echo -n "$(basename $f) "; grep '^{' $f | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --config yolov3 --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/yolov3_trace.md 2>&1
rm -rf $O/trace
head -40 $O/yolov3_trace.md
