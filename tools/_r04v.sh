#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04v; mkdir -p $O; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --config retinanet --dtype f32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-conv-events > $O/x3.log 2>&1
python tools/summarize_trace_csv.py $O/trace 6 > $O/trace.md
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/trace
head -45 $O/trace.md
