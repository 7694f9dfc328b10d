O=gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "v7_persistent" > $O/tests_v7.log 2>&1; tail -6 $O/tests_v7.log | cut -c1-300
L=conv3_1,conv3_2,conv4_1,conv4_2,conv6,pred1
for i in 1 2; do
echo V7; python tools/conv_bench.py $L fwd 30 0 2>&1 | grep "conv\|pred\|sum"
echo V6; ODTK_DBG2=1 python tools/conv_bench.py $L fwd 30 0 2>&1 | grep "conv\|pred\|sum"
done
