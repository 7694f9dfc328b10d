#!/bin/bash
# round 3, call L: tail filter gradients on a third stream -- SSD300 tests + same-process A/B
set -u
TAG=${1:-r03l}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py tests/test_gpu_ssd512.py -x -q ) > $O/ssd.log 2>&1
tail -4 $O/ssd.log | cut -c1-300
( time timeout 600 python -m pytest tests/test_gpu_insitu_configs.py -q -k "ssd300" ) > $O/insitu.log 2>&1
tail -3 $O/insitu.log | cut -c1-300
timeout 600 python tools/ab_bench.py base= notwg=cfg:tail_wgrad_stream=0 list=cfg:use_graph=list --rounds 8 --block 25 > $O/ab.md 2>&1
tail -5 $O/ab.md
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
