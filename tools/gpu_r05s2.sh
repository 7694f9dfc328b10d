#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05s; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
ODTK_DBG2=16384 timeout 300 python tools/conv_bench.py y52_3,y26_3 dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/OFF /' >> $O/ab2.txt
timeout 300 python tools/conv_bench.py y52_3,y26_3 dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/ON  /' >> $O/ab2.txt
done
cut -c1-110 $O/ab2.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "halo_kernel_on_128 or is_taken" > $O/t_kern2.log 2>&1; tail -2 $O/t_kern2.log
for i in 1 2 3; do
timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 ON  /"
timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events --debug-set 6:16384 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 OFF /"
done
