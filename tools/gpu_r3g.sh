#!/bin/bash
# round 3, call G: batch-norm small-kernel shapes A/B; BN kernel tests; one-step timeline of the current step
set -u
TAG=${1:-r03g}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "batchnorm" ) > $O/kern_bn.log 2>&1
tail -3 $O/kern_bn.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-conv-events --warmup 5 --steps 30"
for rep in 1 2; do
  timeout 200 $B > $O/new_$rep.log 2>&1
  timeout 200 $B --debug-set 4:-3 > $O/wide_$rep.log 2>&1
  timeout 200 $B --debug-set 4:4096 > $O/rows4096_$rep.log 2>&1
  timeout 200 $B --debug-set 4:0 > $O/three_$rep.log 2>&1
  timeout 200 $B --debug-set 4:0,4:-2 > $O/two_$rep.log 2>&1
done
for f in $O/*_[12].log; do echo -n "$(basename $f) "; tail -1 $f | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/trace -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events > $O/trace.log 2>&1
python tools/timeline.py $O/trace 1 > $O/timeline.md 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/trace.md 2>&1
rm -rf $O/trace
head -12 $O/timeline.md
