#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05r; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
ODTK_DBG2=32768 timeout 300 python tools/conv_bench.py y26_3,y13_3,pred2 fwd,dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/OFF /' >> $O/ab.txt
timeout 300 python tools/conv_bench.py y26_3,y13_3,pred2 fwd,dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/ON  /' >> $O/ab.txt
done
cut -c1-110 $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "halo_kernel_on_128 or is_taken or fused" > $O/t_kern.log 2>&1; tail -3 $O/t_kern.log
for i in 1 2 3; do
timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 ON  /"
timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events --debug-set 6:32768 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 OFF /"
done
