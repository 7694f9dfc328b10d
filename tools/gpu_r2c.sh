#!/bin/bash
# round 2, call C: head stream beside the extras chain -- correctness (SSD300 / dist tests), same-box A/B bench
set -u
TAG=${1:-r02c}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py tests/test_gpu_dist.py -q -s ) > $O/ssd.log 2>&1; echo "ssd exit $?" >> $O/ssd.log
grep -E "passed|failed|exit|bf16 scores|bf16 detections|per level|^FAILED|^ERROR" $O/ssd.log | tail -12
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events > $O/bench_tail_$i.log 2>&1; tail -1 $O/bench_tail_$i.log | cut -c1-160
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events --no-tail-stream > $O/bench_notail_$i.log 2>&1; tail -1 $O/bench_notail_$i.log | cut -c1-160
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events --eager > $O/bench_tail_eager.log 2>&1; tail -1 $O/bench_tail_eager.log | cut -c1-160
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-conv-events --eager --no-tail-stream > $O/bench_notail_eager.log 2>&1; tail -1 $O/bench_notail_eager.log | cut -c1-160
