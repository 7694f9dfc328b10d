#!/bin/bash
set -u
TAG=${1:-r03s}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/ab_bench.py base= same=cfg:verbose=0 notwg=cfg:tail_wgrad_stream=0 notail=cfg:tail_stream=0 graph=cfg:use_graph=1 list=cfg:use_graph=list match=cfg:match_stream=1 nofuse=cfg:fuse_pool=0 --rounds 8 --block 25 > $O/ab.md 2>&1; cat $O/ab.md
timeout 900 python tools/ab_bench.py base= same2=cfg:verbose=0 same3=cfg:seed=0 same4=cfg:keep_prob=0.5 same5=cfg:mode=train --own-streams --rounds 6 --block 25 > $O/ab_own.md 2>&1; cat $O/ab_own.md
