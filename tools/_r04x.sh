#!/bin/bash
mkdir -p gpurun_out/r04x
timeout 600 python bench.py --config retinanet --steps 3 --warmup 2 --no-cpu-baseline --conv-table gpurun_out/r04x/table_policy.md 2>&1 | grep '^{' | cut -c60-160
timeout 600 python bench.py --config retinanet --steps 3 --warmup 2 --no-cpu-baseline --debug-set 6:8 --conv-table gpurun_out/r04x/table_all.md 2>&1 | grep '^{' | cut -c60-160
