#!/bin/bash
mkdir -p gpurun_out/r04x
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "x3 or test_conv_v3_engine or legacy_engine_extra" -x 2>&1 | tail -3
timeout 300 python tools/refinedet_bench.py f32x3 32 5 2>&1 | grep "images/s"
timeout 300 python tools/refinedet_bench.py bf16 32 5 2>&1 | grep "images/s"
timeout 300 python tools/pfpnet_bench.py f32x3 32 5 2>&1 | grep "images/s"
