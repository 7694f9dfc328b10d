#!/bin/bash
# round 2, call F: SSD512 + the earlier fixes (Adam order, dist tolerance), then the whole GPU suite, fcos bench
set -u
TAG=${1:-r02f}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ssd512.py tests/test_gpu_centernet_model.py -q -s ) > $O/new.log 2>&1; echo "new exit $?" >> $O/new.log
grep -E "passed|failed|exit|^FAILED|^ERROR" $O/new.log | tail -12
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ssd512.py --deselect tests/test_gpu_centernet_model.py ) > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log; grep -E "passed|failed|^FAILED|^ERROR|exit" $O/pytest.log | tail -8
timeout 300 python tools/fcos_bench.py f32 16 3 512 > $O/fcos_bench.log 2>&1; tail -1 $O/fcos_bench.log
timeout 300 python tools/fcos_bench.py bf16 16 5 512 >> $O/fcos_bench.log 2>&1; tail -1 $O/fcos_bench.log
