#!/bin/bash
# round 2, call I: kernel timeline of one step (eager launches, two streams)
set -u
TAG=${1:-r02i}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
BCMD="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events --eager"
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/timeline.py $O/trace 1 > $O/timeline.md; head -12 $O/timeline.md
python tools/summarize_trace_csv.py $O/trace 7 > $O/trace.md; head -3 $O/trace.md
rm -rf $O/trace
