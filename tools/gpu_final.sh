#!/bin/bash
# End-of-round check on one box: full GPU suite, smoke(), the profile set of tools/gpu_profile.sh and the bench lines of the other BASELINE configurations.
#   bash tools/gpu_final.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=${1:-final}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu.log 2>&1
tail -4 $O/gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
bash tools/gpu_profile.sh $TAG > $O/profile.log 2>&1; head -3 $O/profile.log | cut -c1-250
for c in retinanet yolov3 fcos centernet; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 2>$O/err_$c.log | tail -1 > $O/bench_line_$c.json
  python -c "import json;d=json.load(open('$O/bench_line_$c.json'));print('$c', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])"
done
