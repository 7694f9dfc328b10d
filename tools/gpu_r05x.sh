#!/bin/bash
# round 5, late: the filter-gradient flush with a rotated tile order per split (A/B, key 6 bit 14), DP step through the C-ABI collective on the side stream
set -u
R=$(pwd); O=$R/gpurun_out/r05x; mkdir -p $O; export TMPDIR=/tmp
L=conv1_2,conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,conv7,pred1,pred2
for i in 1 2; do
ODTK_DBG2=16384 timeout 300 python tools/conv_bench.py $L wgrad 30 0 2>&1 | sed 's/^/SAME /' >> $O/wgrad_ab.txt
timeout 300 python tools/conv_bench.py $L wgrad 30 0 2>&1 | sed 's/^/ROT  /' >> $O/wgrad_ab.txt
done
cat $O/wgrad_ab.txt | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "odtk_comm or c_abi_collective" > $O/t_comm.log 2>&1; tail -3 $O/t_comm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "wgrad" > $O/t_wgrad.log 2>&1; tail -3 $O/t_wgrad.log
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events > $O/bench_$i.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log | head -1
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --debug-set 6:16384 > $O/bench_same_$i.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/bench_same_$i.log | head -1
done
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --dp-world1 --collective odtk > $O/bench_dp_odtk.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/bench_dp_odtk.log | head -1
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-conv-events --dp-world1 > $O/bench_dp_torch.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/bench_dp_torch.log | head -1
BCMD="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events --eager --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/trace.md; cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/trace
grep -E "nms_|ssd_loss|softmax_ce" $O/kernel_stats.csv | cut -c1-160
