#!/bin/bash
set -u
TAG=${1:-r03s}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py tests/test_gpu_ssd512.py tests/test_gpu_dist.py -q -x -k "ssd or rccl or dist" ) > $O/ssd.log 2>&1
grep -E "passed|failed|rror" $O/ssd.log | head -5 | cut -c1-300
timeout 900 python tools/ab_bench.py base= same=cfg:verbose=0 noside=cfg:side_front=0 --rounds 10 --block 25 > $O/ab.md 2>&1; cat $O/ab.md
