#!/bin/bash
mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_gpu_retinanet_model.py -q -m gpu -k "f32_model_matches" -s 2>&1 | grep -v "^E   \|where" | tail -12 > gpurun_out/r04w/model.log
cat gpurun_out/r04w/model.log
R=$(pwd); O=$R/gpurun_out/r04w; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --config retinanet --dtype f32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-conv-events > $O/x3.log 2>&1
python tools/summarize_trace_csv.py $O/trace 6 > $O/trace.md
rm -rf $O/trace
head -28 $O/trace.md
