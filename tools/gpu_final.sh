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
