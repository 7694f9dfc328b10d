#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05zzz; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_insitu_configs.py -q -k "bf16" > $O/t_insitu.log 2>&1; tail -2 $O/t_insitu.log
timeout 120 python -m pytest tests/test_gpu_dist.py -q -k "sync_bn" > $O/t_syncbn.log 2>&1; tail -2 $O/t_syncbn.log | grep -E "passed|failed"
timeout 150 python -m pytest tests/test_gpu_yolov3.py tests/test_gpu_centernet.py tests/test_gpu_yolov2.py -q > $O/t_models.log 2>&1; tail -2 $O/t_models.log
for c in yolov3 centernet; do
  timeout 200 python bench.py --config $c --steps 10 --warmup 3 2>$O/err_$c.log | grep '^{' > $O/bench_line_$c.json
  python -c "import json;d=json.load(open('$O/bench_line_$c.json'));print('$c', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])"
  YCMD="python bench.py --config $c --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events"
  timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ytrace -- $YCMD > $O/ytrace.log 2>&1
  python tools/summarize_trace_csv.py $O/ytrace 7 > $O/${c}_trace.md; rm -rf $O/ytrace
  head -1 $O/${c}_trace.md
done
