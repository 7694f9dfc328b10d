#!/bin/bash
set -u
TAG=${1:-r6g}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_bf16_gate.py tests/test_gpu_ssd300.py tests/test_gpu_yolov3.py tests/test_gpu_dist.py -q -x -s --durations=8 ) > $O/gate_tests.log 2>&1
grep -E "^GATE|passed|failed|^FAILED|Error" $O/gate_tests.log | cut -c1-330 | tail -20
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
grep '^{' $O/bench.log > $O/bench_line.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6g/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('epoch'), d.get('extras_s'))
for k,v in d['configs'].items(): print(k, v.get('images_per_sec'), v.get('dtype'), v.get('mfma_busy_pct'), v.get('hbm_bytes_per_step'), (v.get('engine_admission') or '')[:60], v.get('error') or v.get('skipped') or v.get('pmc_note') or '')
PY
