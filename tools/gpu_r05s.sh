#!/bin/bash
# round 5, late: 128 x 128 tiles of the halo kernel instead of chunk-range split-K + finish (A/B: key 6 bit 14)
set -u
R=$(pwd); O=$R/gpurun_out/r05s; mkdir -p $O; export TMPDIR=/tmp
L=y26_3,y13_3,y52_3,pred2,f32_3,f16_3,conv5_2
for i in 1 2; do
ODTK_DBG2=16384 timeout 300 python tools/conv_bench.py $L fwd,dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/OFF /' >> $O/ab.txt
timeout 300 python tools/conv_bench.py $L fwd,dgrad 30 0 2>&1 | grep -v amdgpu.ids | sed 's/^/ON  /' >> $O/ab.txt
done
cut -c1-110 $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "halo_kernel_on_128 or is_taken or conv_v9_engine or fused" > $O/t_kern.log 2>&1; tail -3 $O/t_kern.log
for c in yolov3 fcos centernet; do for i in 1 2; do
timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$c ON  /"
timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events --debug-set 6:16384 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$c OFF /"
done; done
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/ssd300 ON  /'
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 6:16384 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/ssd300 OFF /'
done
