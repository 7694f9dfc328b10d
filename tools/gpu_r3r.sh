#!/bin/bash
set -u
TAG=${1:-r03r}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" ) > $O/kern.log 2>&1
grep -E "passed|failed|rror" $O/kern.log | head -5 | cut -c1-300
for i in 1 2; do
  for dbg in -2147483648 0; do
    timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --kernel-dbg=$dbg --conv-table $O/conv_${dbg}_$i.md 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dbg', d['value'], d['ms_per_step'], d['roofline'])"
  done
done 2>&1 | tee $O/ab.log
