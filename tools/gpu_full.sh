#!/bin/bash
# The whole GPU tier on one box, as the driver runs it (-x: stop at the first failure), then smoke().   bash tools/gpu_full.sh <tag> -> gpurun_out/<tag>/{gpu.log, smoke.log}
set -u
TAG=${1:-full}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( git -C $R rev-parse HEAD 2>/dev/null; time timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 ) > $O/gpu.log 2>&1
tail -14 $O/gpu.log | cut -c1-300
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
