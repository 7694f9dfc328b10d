#!/bin/bash
# round 3, call A: in-situ shadow of the BASELINE configs, RCCL world-1 paths, baseline bench line, clock / power trace
set -u
TAG=${1:-r03a}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_insitu_configs.py -q -s --durations=0 ) > $O/insitu.log 2>&1
echo "insitu exit $?" >> $O/insitu.log
grep -E "passed|failed|error|in-situ|exit|out of bound" $O/insitu.log | cut -c1-700 | tail -40
( time timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -k "rccl" ) > $O/rccl.log 2>&1
tail -5 $O/rccl.log
timeout 300 python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
timeout 300 python tools/clock_trace.py $O/clock_trace.md 2000 > $O/clock_trace.log 2>&1
head -20 $O/clock_trace.md
ls /sys/class/drm/*/device/hwmon/*/ 2>/dev/null | head -40 > $O/hwmon_ls.txt
