#!/bin/bash
set -u
TAG=${1:-r6i}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
YCMD="python $R/bench.py --config yolov3 --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ytrace -- $YCMD > $O/ytrace.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/ytrace 7 > $O/yolov3_f32x3_trace.md; rm -rf $O/ytrace; head -40 $O/yolov3_f32x3_trace.md | cut -c1-170
