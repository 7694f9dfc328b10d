#!/bin/bash
# round 3, call O: batch-norm launch policy -- kernel tests, YOLOv3 / SSD300 effect, in-situ of yolov3 + ssd300
set -u
TAG=${1:-r03o}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "batchnorm" ) > $O/kern_bn.log 2>&1
tail -3 $O/kern_bn.log | cut -c1-300
( time timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -k "yolov3 or ssd300 or centernet-bf16" ) > $O/insitu.log 2>&1
tail -3 $O/insitu.log | cut -c1-300
B="python bench.py --config yolov3 --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events"
for rep in 1 2; do
  timeout 200 $B > $O/y_new_$rep.log 2>&1
  timeout 200 $B --debug-set 4:-5 > $O/y_noauto_$rep.log 2>&1
  timeout 200 $B --debug-set 4:-5,4:1024 > $O/y_old_$rep.log 2>&1
done
for f in $O/y_*.log; do echo -n "$(basename $f) "; grep '^{' $f | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
timeout 600 python tools/ab_bench.py base= noauto=4:-5 --rounds 6 --block 25 > $O/ab.md 2>&1
tail -4 $O/ab.md
