#!/bin/bash
set -u
TAG=${1:-r03p}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pool2x2 or c64 or conv_v3_engine or halo_kernel_64" ) > $O/kern.log 2>&1
grep -E "passed|failed|rror" $O/kern.log | head -5 | cut -c1-300
timeout 300 python tools/c64_ablate.py > $O/ablate.md 2>&1; cat $O/ablate.md
echo HEAD build; ODTK_LIB=$R/object-detection-tensorflow_amd/libodtk_head.so timeout 300 python tools/c64_ablate.py 2>&1 | cut -d'|' -f1-3 | tee $O/ablate_head.md
