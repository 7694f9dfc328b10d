#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/s5; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv" ) > $O/pytest_conv.log 2>&1
echo "pytest exit $?" >> $O/pytest_conv.log
tail -15 $O/pytest_conv.log
timeout 600 python tools/conv_bench.py all fwd,dgrad 10 1,2,3 > $O/convbench.log 2>&1
cat $O/convbench.log
timeout 600 python tools/conv_bench.py conv2_2,conv3_2,conv4_2 fwd 10 3,203,204,208 > $O/convbench_dbg.log 2>&1
cat $O/convbench_dbg.log
