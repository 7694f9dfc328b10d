#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/s3; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/conv_bench.py conv1_2,conv2_2,conv3_2,conv4_2,conv5_2 fwd 10 2,101,102,103,104 > $O/convbench_dbg.log 2>&1
cat $O/convbench_dbg.log
