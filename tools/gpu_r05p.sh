#!/bin/bash
# round 5, late: single-launch batch norm of mid-size maps (spin barrier per column group); A/B: odtk_debug_set(4, -9)
set -u
R=$(pwd); O=$R/gpurun_out/r05p; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -x -k "bn" > $O/t_bn.log 2>&1; tail -3 $O/t_bn.log
for i in 1 2; do
ODTK_BN_COOP=0 timeout 100 python tools/bn_bench.py big 2>&1 | grep -v amdgpu | sed 's/^/OFF /' | cut -c1-150 >> $O/bn.txt
timeout 100 python tools/bn_bench.py big 2>&1 | grep -v amdgpu | sed 's/^/ON  /' | cut -c1-150 >> $O/bn.txt
done
cat $O/bn.txt
timeout 300 python -m pytest tests/test_gpu_insitu_configs.py -q -x -k "yolov3 or centernet" > $O/t_insitu.log 2>&1; tail -3 $O/t_insitu.log
for i in 1 2; do
timeout 200 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 ON  /"
timeout 200 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events --debug-set 4:-9 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 OFF /"
done
