#!/bin/bash
# full GPU check: all gpu tests + bench (+ optional trace)   usage: bash tools/gpu_full.sh <tag> [trace]
set -u
TAG=${1:-full}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log
if [ "${2:-}" = "trace" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-conv-events > $O/trace.log 2>&1
  python tools/summarize_trace_csv.py $O/trace 6 > $O/trace.md; head -30 $O/trace.md
fi
