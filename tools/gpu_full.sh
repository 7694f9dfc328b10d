#!/bin/bash
set -u
TAG=${1:-full}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu.log 2>&1
tail -5 $O/gpu.log | cut -c1-300
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
