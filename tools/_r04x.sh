#!/bin/bash
mkdir -p gpurun_out/r04x
timeout 2600 python -m pytest tests/test_gpu_lhrcnn.py tests/test_gpu_refinedet.py tests/test_gpu_refinedet_model.py tests/test_gpu_pfpnet_model.py tests/test_gpu_bf16_gate.py tests/test_gpu_yolov2.py tests/test_gpu_fcos_model.py tests/test_gpu_centernet_model.py tests/test_gpu_dist.py -q -m gpu 2>&1 > gpurun_out/r04x/affected.log
tail -12 gpurun_out/r04x/affected.log | cut -c1-300
