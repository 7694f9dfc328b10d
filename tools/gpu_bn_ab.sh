#!/bin/bash
# A/B of two builds of libodtk.so (libodtk_base.so = HEAD, libodtk.so = working tree): batch-norm shapes and the BN-heavy configurations
set -u
R=$(pwd); O=$R/gpurun_out/r05o; mkdir -p $O; export TMPDIR=/tmp
L=object-detection-tensorflow_amd
cp $L/libodtk.so /tmp/new.so
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -k "batchnorm" 2>&1 | tail -2
cp $L/libodtk_base.so $L/libodtk.so; timeout 100 python tools/bn_bench.py narrow 2>&1 | grep -v amdgpu | sed 's/^/BASE /' | cut -c1-150 > $O/bn.txt
cp /tmp/new.so $L/libodtk.so; timeout 100 python tools/bn_bench.py narrow 2>&1 | grep -v amdgpu | sed 's/^/NEW  /' | cut -c1-150 >> $O/bn.txt
cat $O/bn.txt
for c in centernet yolov3; do for i in 1 2; do
cp $L/libodtk_base.so $L/libodtk.so; timeout 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$c BASE /"
cp /tmp/new.so $L/libodtk.so; timeout 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$c NEW  /"
done; done
