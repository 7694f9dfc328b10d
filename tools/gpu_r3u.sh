#!/bin/bash
set -u
TAG=${1:-r03u2}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for c in retinanet yolov3 fcos centernet; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 2>$O/err_$c.log | tail -1 > $O/bench_line_$c.json
  python -c "import json;d=json.load(open('$O/bench_line_$c.json'));print('$c', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'], d['cpu_baseline']['value'])"
done
