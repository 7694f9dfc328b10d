#!/bin/bash
mkdir -p gpurun_out/r04x
timeout 1500 python -m pytest tests/test_gpu_insitu_configs.py -q -m gpu -k "retinanet-f32x3" -x > gpurun_out/r04x/insitu_x3.log 2>&1
tail -3 gpurun_out/r04x/insitu_x3.log
grep -n "in-situ retinanet" -A12 gpurun_out/r04x/insitu_x3.log | cut -c1-150
