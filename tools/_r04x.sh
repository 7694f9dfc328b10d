#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_retinanet_model.py tests/test_gpu_fcos_model.py tests/test_gpu_centernet_model.py tests/test_gpu_yolov3.py tests/test_gpu_yolov2.py tests/test_gpu_lhrcnn.py tests/test_gpu_refinedet_model.py tests/test_gpu_pfpnet_model.py tests/test_gpu_ssd300.py -q -m gpu -k "not in_situ" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_insitu_configs.py tests/test_gpu_lhrcnn.py -q -m gpu -k "f32 and (in_situ or insitu)" 2>&1 | tail -3
