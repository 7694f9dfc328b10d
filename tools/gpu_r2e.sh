#!/bin/bash
# round 2, call E: CenterNet model, FCOS shared heads, TF known answers on the GPU; then the whole GPU suite
set -u
TAG=${1:-r02e}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_centernet_model.py tests/test_gpu_fcos_model.py tests/test_gpu_tf_known_answers.py -q -s ) > $O/new.log 2>&1; echo "new exit $?" >> $O/new.log
grep -E "passed|failed|exit|^FAILED|^ERROR|relative gradient" $O/new.log | tail -12
timeout 300 python tools/centernet_bench.py f32 16 3 512 > $O/cn_bench.log 2>&1; tail -1 $O/cn_bench.log
timeout 300 python tools/centernet_bench.py bf16 16 5 512 >> $O/cn_bench.log 2>&1; tail -1 $O/cn_bench.log
timeout 300 python tools/fcos_bench.py f32 16 3 512 > $O/fcos_bench.log 2>&1; tail -1 $O/fcos_bench.log
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_centernet_model.py --deselect tests/test_gpu_fcos_model.py --deselect tests/test_gpu_tf_known_answers.py ) > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
