#!/bin/bash
# First GPU call of the next round: what the second session of round 3 left unverified on hardware (its GPU minutes were spent), in order of value.
#   bash tools/gpu_round4_first.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=${1:-r04a}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
# 1. LHRCNN's opt-in bf16 engine: every launch of a step in situ at 700 x 1100 batch 32, then its step time next to the f32 engine's
ODTK_RUN_UNVERIFIED=1 timeout 300 python -m pytest tests/test_gpu_lhrcnn.py tests/test_gpu_tf_known_answers.py -q -s -k "(in_situ and bf16) or crop_and_resize_tables or pooling_same or conv2d_orientation or momentum_optimizer" > $O/lhrcnn_bf16_insitu.log 2>&1; tail -3 $O/lhrcnn_bf16_insitu.log | cut -c1-300
timeout 120 python tools/lhrcnn_bench.py 32 5 700 1100 f32 2>&1 | tail -1 | tee $O/lhrcnn_bench_f32.log
timeout 120 python tools/lhrcnn_bench.py 32 5 700 1100 bf16 2>&1 | tail -1 | tee $O/lhrcnn_bench_bf16.log
# 2. kernel trace of the f32 step with the vectorised depthwise kernels (profiles/r03zzzz_lhrcnn_700x1100_b32_kernel_trace.md has the first version)
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python tools/lhrcnn_bench.py 32 3 > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 5 > $O/lhrcnn_trace.md; rm -rf $O/trace; head -14 $O/lhrcnn_trace.md | cut -c1-160
