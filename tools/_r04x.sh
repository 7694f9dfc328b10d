#!/bin/bash
mkdir -p gpurun_out/r04zzz
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04zzz/full_gpu_tests.log 2>&1
tail -4 gpurun_out/r04zzz/full_gpu_tests.log | cut -c1-300
