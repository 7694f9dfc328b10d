#!/bin/bash
# round 3, call J: launch list tests + A/B; bf16 gate tests; RetinaNet longer training; FULL gpu suite
set -u
TAG=${1:-r03j}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ssd300.py -q -x -k "launch_list" ) > $O/list.log 2>&1
tail -4 $O/list.log | cut -c1-300
timeout 600 python tools/ab_bench.py base= list=cfg:use_graph=list graph=cfg:use_graph=1 --rounds 8 --block 25 > $O/ab_list.md 2>&1
tail -6 $O/ab_list.md
timeout 600 python tools/bf16_after_training.py retinanet 1500 2 > $O/bf16_retinanet_1500.log 2>&1
tail -4 $O/bf16_retinanet_1500.log | cut -c1-400
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_full.log 2>&1
tail -6 $O/pytest_full.log | cut -c1-400
