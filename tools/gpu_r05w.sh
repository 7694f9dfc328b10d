#!/bin/bash
# round 5, late: when the C-ABI communicator is created (before / after the model's streams) -- the data-parallel step in a world of one rank
set -u
R=$(pwd); O=$R/gpurun_out/r05w; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-extras --no-cpu-baseline --no-conv-events"
for i in 1 2; do
timeout 300 $B > $O/plain_$i.log 2>&1; echo plain $(grep -o '"ms_per_step": [0-9.]*' $O/plain_$i.log | head -1)
timeout 300 $B --dp-world1 > $O/torch_$i.log 2>&1; echo torch $(grep -o '"ms_per_step": [0-9.]*' $O/torch_$i.log | head -1) $(grep -o '"ms_per_step_without_comm": [0-9.]*' $O/torch_$i.log)
timeout 300 $B --dp-world1 --collective odtk > $O/odtk_early_$i.log 2>&1; echo odtk_early $(grep -o '"ms_per_step": [0-9.]*' $O/odtk_early_$i.log | head -1) $(grep -o '"ms_per_step_without_comm": [0-9.]*' $O/odtk_early_$i.log)
ODTK_BENCH_LATE_COMM=1 timeout 300 $B --dp-world1 --collective odtk > $O/odtk_late_$i.log 2>&1; echo odtk_late $(grep -o '"ms_per_step": [0-9.]*' $O/odtk_late_$i.log | head -1) $(grep -o '"ms_per_step_without_comm": [0-9.]*' $O/odtk_late_$i.log)
done
