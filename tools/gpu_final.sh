#!/bin/bash
# End-of-round check on one box: smoke(), the profile set of tools/gpu_profile.sh (bench line, kernel trace, three PMC passes), the one-step timeline, the phase
# times without a profiler, and the bench lines of the other BASELINE configurations.  (The full GPU suite is its own call: python -m pytest tests -m gpu -q.)
#   bash tools/gpu_final.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=${1:-final}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
bash tools/gpu_profile.sh $TAG > $O/profile.log 2>&1; head -3 $O/profile.log | cut -c1-250
timeout 120 python tools/phase_times.py 30 2>/dev/null > $O/phase_times.md; head -5 $O/phase_times.md
for c in retinanet yolov3 fcos centernet; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 2>$O/err_$c.log | grep '^{' > $O/bench_line_$c.json
  python -c "import json;d=json.load(open('$O/bench_line_$c.json'));print('$c', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])"
done
# (round 6: YOLOv3's class default is f32x3; the bf16 engine the gate does not admit, next to it)
timeout 600 python bench.py --config yolov3 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err_yolov3_bf16.log | grep '^{' > $O/bench_line_yolov3_bf16.json
python -c "import json;d=json.load(open('$O/bench_line_yolov3_bf16.json'));print('yolov3 bf16', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])"
# BASELINE configurations 3-5 (per-GPU shares): kernel trace + the PMC passes of the same command (round-4 review: "none has a PMC set")
for c in yolov3 fcos centernet retinanet; do
  YCMD="python bench.py --config $c --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events"
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ytrace -- $YCMD > $O/ytrace.log 2>&1
  python tools/summarize_trace_csv.py $O/ytrace 7 > $O/${c}_trace.md; rm -rf $O/ytrace
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/ypmcA -- $YCMD > $O/ypmcA.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $O/ypmcB -- $YCMD > $O/ypmcB.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $O/ypmcC -- $YCMD > $O/ypmcC.log 2>&1
  python tools/pmc_summary.py $O/${c}_pmc.json $O/ypmcA $O/ypmcB $O/ypmcC > $O/${c}_pmc.md 2>&1; rm -rf $O/ypmcA $O/ypmcB $O/ypmcC
  head -4 $O/${c}_trace.md
done
# the classes of SURVEY.md 8f.4 on their default engines (round 4: f32x3 for RefineDet320 / PFPNetR / LH_RCNN) and the engines next to them
: > $O/classes.log
for e in f32 f32x3 bf16; do
  timeout 300 python tools/refinedet_bench.py $e 32 5 2>&1 | grep "images/s" >> $O/classes.log
  timeout 300 python tools/pfpnet_bench.py $e 32 5 2>&1 | grep "images/s" >> $O/classes.log
  timeout 300 python tools/yolov2_bench.py $e 32 5 2>&1 | grep "images/s" >> $O/classes.log
done
for e in f32 f32x3; do
  timeout 300 python tools/fcos_bench.py $e 16 5 2>&1 | grep "images/s" >> $O/classes.log
  timeout 300 python tools/centernet_bench.py $e 16 5 2>&1 | grep "images/s" >> $O/classes.log
  timeout 300 python tools/lhrcnn_bench.py 32 3 700 1100 $e 2>&1 | grep -i "images/s" | cut -c1-90 >> $O/classes.log
done
timeout 300 python tools/ssd512_bench.py 2>&1 | grep "images/s" >> $O/classes.log
cat $O/classes.log

