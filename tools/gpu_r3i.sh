#!/bin/bash
# round 3, call I: in-process A/B of the SSD300 step; bf16 after f32 training for the norm-heavy classes
set -u
TAG=${1:-r03i}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/ab_bench.py base= wide=4:-3 rows4096=4:4096 three=4:0 nofuse=cfg:fuse_pool=0 notail=cfg:tail_stream=0 graph=cfg:use_graph=1 --rounds 8 --block 25 > $O/ab.md 2>&1
tail -10 $O/ab.md
for m in retinanet fcos centernet yolov3; do
  timeout 600 python tools/bf16_after_training.py $m 300 4 > $O/bf16_$m.log 2>&1
  tail -6 $O/bf16_$m.log | cut -c1-400
done
