#!/bin/bash
# round 2, call D: head-stream equivalence test, launch-mode auto, inference test, bench (auto / graph / eager) twice
set -u
TAG=${1:-r02d}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -q -s -k "head_stream or test_one_image or loss_curve or epoch" ) > $O/ssd.log 2>&1; echo "ssd exit $?" >> $O/ssd.log
grep -E "passed|failed|exit|bf16 scores|bf16 detections|per level|^FAILED|^ERROR|launch mode" $O/ssd.log | tail -12
for i in 1 2; do
  for mode in "" "--graph" "--eager"; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events $mode > $O/bench_${i}_${mode#--}.log 2>&1
    python - <<P
import json
d=json.loads([l for l in open("$O/bench_${i}_${mode#--}.log") if l.startswith('{')][-1])
print("$mode", d['value'], d['ms_per_step'], d['config'].get('launch_mode_calibration'), d['config']['launch'][:40])
P
  done
done
