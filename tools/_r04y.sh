#!/bin/bash
mkdir -p gpurun_out/r04y
for e in f32 f32x3; do
timeout 600 python bench.py --config retinanet --dtype $e --steps 3 --warmup 2 --no-cpu-baseline --conv-table gpurun_out/r04y/table_$e.md 2>&1 | grep '^{' > gpurun_out/r04y/bench_$e.json
done
ls -la gpurun_out/r04y
