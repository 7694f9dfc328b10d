O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
python tools/conv_bench.py conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,conv7,pred1,pred2,conv8_2 fwd,dgrad 20 0 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
