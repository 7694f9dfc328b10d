#!/bin/bash
# First GPU call of the next round: what the second session of round 3 left unverified on hardware (its GPU minutes were spent), in order of value.
#   bash tools/gpu_round4_first.sh <tag>   -> gpurun_out/<tag>/...
# About 18 minutes of box time as written (18 bench processes of 25-40 s, three test selections, one trace): run it under `gpurun --timeout 1500`, or
# comment out sections -- 1 (LHRCNN bf16) and 2 (batch-norm ticket on YOLOv3) carry the most.
set -u
TAG=${1:-r04a}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
# 1. LHRCNN's opt-in bf16 engine: every launch of a step in situ at 700 x 1100 batch 32, then its step time next to the f32 engine's
ODTK_RUN_UNVERIFIED=1 timeout 300 python -m pytest tests/test_gpu_lhrcnn.py tests/test_gpu_tf_known_answers.py -q -s -k "(in_situ and bf16) or second_pinned or crop_and_resize_tables or pooling_same or conv2d_orientation or momentum_optimizer" > $O/lhrcnn_bf16_insitu.log 2>&1; tail -3 $O/lhrcnn_bf16_insitu.log | cut -c1-300
timeout 120 python tools/lhrcnn_bench.py 32 5 700 1100 f32 2>&1 | tail -1 | tee $O/lhrcnn_bench_f32.log
timeout 120 python tools/lhrcnn_bench.py 32 5 700 1100 bf16 2>&1 | tail -1 | tee $O/lhrcnn_bench_bf16.log
# 2. batch norm with the finalize launch folded into the statistics launch by ticket (odtk_debug_set(4, -7), default off): first its parity on hardware (the
#    fences are what the CPU emulation cannot see), then YOLOv3 at config 4's per-GPU share with and without it (profiles/r03k: 150 finalize launches = 1.45 of 11.1 ms)
#    Round 2 measured a ticket scheme with a device-scope fence per workgroup SLOWER: -7 is that scheme again (expect the same), -9 the fence-free one
#    (write-through partials + s_waitcnt, no L2 write-back / invalidate) -- the candidate.
ODTK_RUN_UNVERIFIED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "batchnorm and ticket" > $O/bn_ticket_tests.log 2>&1; tail -2 $O/bn_ticket_tests.log | cut -c1-200
for sw in "" "--debug-set 4:-7" "--debug-set 4:-9" "" "--debug-set 4:-9"; do timeout 300 python bench.py --config yolov3 --steps 30 --warmup 5 --no-cpu-baseline $sw 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('yolov3 [$sw]', d['value'], d['ms_per_step'])" | tee -a $O/yolov3_bn_ticket_ab.log; done
for sw in "" "--debug-set 4:-9" "" "--debug-set 4:-9"; do timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-conv-events $sw 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ssd300 [$sw]', d['value'], d['ms_per_step'])" | tee -a $O/yolov3_bn_ticket_ab.log; done     # (7 layers of the SSD300 step take the three-launch batch norm; tools/ab_bench.py resolves 0.3 % if this looks promising)
for sw in "" "--debug-set 4:-9"; do timeout 300 python bench.py --config retinanet --steps 6 --warmup 2 --no-cpu-baseline $sw 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet [$sw]', d['value'], d['ms_per_step'])" | tee -a $O/yolov3_bn_ticket_ab.log; done
# 2b. the same ticket scheme for the bf16 group norms of FCOS (odtk_debug_set(7, -7), default off): parity, then config 5 with and without it (about 60 finalize
#     launches of a 17 ms step); CenterNet (config 3) takes the batch-norm one
ODTK_RUN_UNVERIFIED=1 timeout 300 python -m pytest tests/test_gpu_retinanet_model.py -q -k "ticket_finalize" > $O/gn_ticket_tests.log 2>&1; tail -2 $O/gn_ticket_tests.log | cut -c1-200
for sw in "" "--debug-set 7:-7" "--debug-set 7:-9" "" "--debug-set 7:-9"; do timeout 300 python bench.py --config fcos --steps 20 --warmup 4 --no-cpu-baseline $sw 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fcos [$sw]', d['value'], d['ms_per_step'])" | tee -a $O/gn_ticket_ab.log; done
for sw in "" "--debug-set 4:-9"; do timeout 300 python bench.py --config centernet --steps 20 --warmup 4 --no-cpu-baseline $sw 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centernet [$sw]', d['value'], d['ms_per_step'])" | tee -a $O/yolov3_bn_ticket_ab.log; done
# 3. kernel trace of the f32 step with the vectorised depthwise kernels (profiles/r03zzzz_lhrcnn_700x1100_b32_kernel_trace.md has the first version)
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python tools/lhrcnn_bench.py 32 3 > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 5 > $O/lhrcnn_trace.md; rm -rf $O/trace; head -14 $O/lhrcnn_trace.md | cut -c1-160
