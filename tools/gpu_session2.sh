#!/bin/bash
# GPU session 2: v3 engine correctness + per-layer A/B
set -u
R=$(pwd); O=$R/gpurun_out/s2; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv" ) > $O/pytest_conv.log 2>&1
echo "pytest exit $?" >> $O/pytest_conv.log
timeout 600 python tools/conv_bench.py all fwd,dgrad 10 1,2 > $O/convbench.log 2>&1
tail -5 $O/pytest_conv.log; cat $O/convbench.log
