#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench.json; python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-conv-events 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('200 steps', d['value'], d['ms_per_step'])"
