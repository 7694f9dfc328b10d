#!/bin/bash
# round 2, call L: PFPNetR: new glue kernels, model parity, RefineDet regression, throughput
set -u
TAG=${1:-r02l}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "resize_bilinear_align or copy_channels" ) > $O/kern.log 2>&1; echo "kern exit $?" >> $O/kern.log; tail -4 $O/kern.log
( timeout 1200 python -m pytest tests/test_gpu_pfpnet_model.py -q -x -s ) > $O/pfp.log 2>&1; echo "pfp exit $?" >> $O/pfp.log; tail -12 $O/pfp.log
( timeout 900 python -m pytest tests/test_gpu_refinedet_model.py tests/test_gpu_refinedet.py -q -x ) > $O/rd.log 2>&1; echo "rd exit $?" >> $O/rd.log; tail -3 $O/rd.log
timeout 600 python tools/pfpnet_bench.py > $O/bench.log 2>&1; tail -4 $O/bench.log
