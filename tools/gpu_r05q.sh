#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05q; mkdir -p $O; export TMPDIR=/tmp
L=object-detection-tensorflow_amd
cp $L/libodtk.so /tmp/new.so
for i in 1 2; do
cp $L/libodtk_base.so $L/libodtk.so; timeout 200 python tools/bn_bench.py big 2>&1 | grep -v amdgpu | sed 's/^/BASE /' | cut -c1-150 >> $O/bn.txt
cp /tmp/new.so $L/libodtk.so; timeout 200 python tools/bn_bench.py big 2>&1 | grep -v amdgpu | sed 's/^/NEW  /' | cut -c1-150 >> $O/bn.txt
done
cat $O/bn.txt
for i in 1 2; do
cp $L/libodtk_base.so $L/libodtk.so; timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 BASE /"
cp /tmp/new.so $L/libodtk.so; timeout 300 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline --no-conv-events 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/yolov3 NEW  /"
done
