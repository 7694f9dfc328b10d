#!/bin/bash
set -u
TAG=${1:-r6m}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/y2.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['R']); sys.path.insert(0, os.environ['R'] + '/tools')
from odtk import ops
ops.debug_set(5, int(os.environ.get('WG', '1')))
sys.argv = ['yolov2_bench.py', 'bf16', '32', '5']
exec(open(os.environ['R'] + '/tools/yolov2_bench.py').read())
PY
for wg in 1 0 1 0; do R=$R WG=$wg timeout 300 python /tmp/y2.py 2>&1 | grep "images/s" | sed "s/^/WG=$wg /"; done
cd /tmp; R=$R WG=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/tr1 -- python /tmp/y2.py > $O/tr1.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/tr1 7 > $O/yolov2_det_trace.md; rm -rf $O/tr1; head -16 $O/yolov2_det_trace.md | cut -c1-170
cd /tmp; R=$R WG=0 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/tr0 -- python /tmp/y2.py > $O/tr0.log 2>&1; cd $R
python tools/summarize_trace_csv.py $O/tr0 7 > $O/yolov2_atomics_trace.md; rm -rf $O/tr0; head -12 $O/yolov2_atomics_trace.md | cut -c1-170
