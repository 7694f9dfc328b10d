#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retinanet_model.py -q -m gpu -k "x3 or f32_model_matches" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -m gpu -k "retinanet-f32x3 or fcos-f32x3" 2>&1 | tail -2
timeout 600 python bench.py --config retinanet --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | cut -c60-160
