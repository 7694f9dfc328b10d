#!/bin/bash
# round 2, call A: the new batch-32 parity tests (verbose), the whole GPU suite, smoke, bench
set -u
TAG=${1:-r02a}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ssd300_b32.py -q -s ) > $O/b32.log 2>&1
echo "b32 exit $?" >> $O/b32.log
grep -E "passed|failed|exit|Error|assert" $O/b32.log | tail -15
( time timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ssd300_b32.py ) > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-600
