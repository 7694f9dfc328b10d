#!/bin/bash
set -u
TAG=${1:-r03q2}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
BCMD="python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-conv-events --eager"
for lib in libodtk_head.so libodtk.so; do
  ODTK_LIB=$R/object-detection-tensorflow_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace_$lib -- $BCMD > $O/trace_$lib.log 2>&1
  tail -1 $O/trace_$lib.log | cut -c1-200
  python tools/summarize_trace_csv.py $O/trace_$lib 15 > $O/trace_$lib.md
  python tools/timeline.py $O/trace_$lib > $O/timeline_$lib.md 2>&1
  head -8 $O/timeline_$lib.md; grep -E "c64k64|c8k64" $O/timeline_$lib.md
  grep -E "c64k64|total kernel" $O/trace_$lib.md
  rm -rf $O/trace_$lib
done
