#!/bin/bash
mkdir -p gpurun_out/r04zz
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04zz/full_gpu_tests.log 2>&1
tail -6 gpurun_out/r04zz/full_gpu_tests.log | cut -c1-300
