#!/bin/bash
set -u
TAG=${1:-r03s}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py tests/test_gpu_ssd300.py tests/test_gpu_ssd512.py -q -x -k "ssd" ) > $O/ssd.log 2>&1
grep -E "passed|failed|rror" $O/ssd.log | head -5 | cut -c1-300
timeout 600 python tools/ab_bench.py base= b1=cfg:twg_batch=1 b2=cfg:twg_batch=2 b8=cfg:twg_batch=8 b16=cfg:twg_batch=16 --rounds 8 --block 25 > $O/ab.md 2>&1; cat $O/ab.md
