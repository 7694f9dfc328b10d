#!/bin/bash
set -u
TAG=${1:-r03t}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  for lib in libodtk_head.so libodtk_early.so; do
    ODTK_LIB=$R/object-detection-tensorflow_amd/$lib timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --conv-table $O/conv_${lib}_$i.md 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done 2>&1 | tee $O/ab.log
paste <(grep v6 $O/conv_libodtk_head.so_2.md | sort | awk '{print $1,$4,$5,$6,$7,$11,$12}') <(grep v6 $O/conv_libodtk_early.so_2.md | sort | awk '{print $12}')
