#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05t; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/dp_determinism.py warm 3 2>&1 | grep "losses" | cut -c1-150 > $O/dp_determinism.txt; cat $O/dp_determinism.txt | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "wgrad or l2norm" > $O/t_kern.log 2>&1; tail -2 $O/t_kern.log
timeout 900 python -m pytest tests/test_gpu_ssd300_b32.py -q -k "deterministic" > $O/t_b32.log 2>&1; tail -2 $O/t_b32.log
timeout 900 python -m pytest tests/test_gpu_dist.py -q -k "odtk_comm or c_abi_collective" > $O/t_comm.log 2>&1; grep -E "passed|failed" $O/t_comm.log
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events > $O/bench_$i.log 2>&1; echo default $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log | head -1)
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 5:1 > $O/bench_det_$i.log 2>&1; echo deterministic $(grep -o '"ms_per_step": [0-9.]*' $O/bench_det_$i.log | head -1)
done
