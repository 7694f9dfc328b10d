O=gpurun_out/r04j; mkdir -p $O; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --config yolov3 --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/yolov3_trace.md; rm -rf $O/trace; head -45 $O/yolov3_trace.md | cut -c1-150
