#!/bin/bash
# round 3, call H: bench.py --config lines (configs 3-5) + in-process A/B of the SSD300 step + power cap
set -u
TAG=${1:-r03h}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for c in retinanet yolov3 fcos centernet; do
  ( time timeout 600 python bench.py --config $c --steps 10 --warmup 3 ) > $O/bench_$c.log 2>&1
  grep '^{' $O/bench_$c.log | tail -1 | cut -c1-420
done
for c in retinanet fcos centernet; do
  timeout 300 python bench.py --config $c --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_${c}_bf16.log 2>&1
  grep '^{' $O/bench_${c}_bf16.log | tail -1 | cut -c1-200
done
timeout 600 python tools/ab_bench.py base= wide=4:-3 rows4096=4:4096 three=4:0 nofuse=cfg:fuse_pool=0 notail=cfg:tail_stream=0 oldc8=2:2048 --rounds 8 --block 25 > $O/ab.md 2>&1
cat $O/ab.md | tail -12
cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap 2>/dev/null | head -8 > $O/power_cap.txt; cat $O/power_cap.txt | tr '\n' ' '
