#!/bin/bash
# round 5, late: odtk_comm, scratch arenas, NMS scan with sub-block flags, per-kind bf16 bounds -- tests + a kernel trace of the step
set -u
R=$(pwd); O=$R/gpurun_out/r05y; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "odtk_comm or c_abi_collective" > $O/t_comm.log 2>&1; tail -3 $O/t_comm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "nms or mining or x3 or loss" > $O/t_kern.log 2>&1; tail -3 $O/t_kern.log
timeout 900 python -m pytest tests/test_gpu_engine_bf16.py -q -s > $O/t_bf16.log 2>&1; grep -E "loss f32|passed|failed" $O/t_bf16.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -q > $O/t_ssd.log 2>&1; tail -3 $O/t_ssd.log
timeout 900 python -m pytest tests/test_gpu_insitu_configs.py -q -k "f32x3" > $O/t_x3.log 2>&1; tail -3 $O/t_x3.log
timeout 300 python bench.py --no-extras --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
BCMD="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events --eager --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/trace.md; rm -rf $O/trace
grep -E "nms_|ssd_loss|softmax_ce|total kernel" $O/trace.md
timeout 300 python bench.py --no-extras --no-cpu-baseline --dp-world1 --collective odtk > $O/bench_dp_odtk.log 2>&1; tail -1 $O/bench_dp_odtk.log | cut -c1-200
