O=gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_v3_engine or geometries_bf16 or preprocess" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
L=conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6,pred1
python tools/conv_bench.py $L fwd,dgrad 20 0 2>&1 | grep -v amdgpu.ids | tee $O/nst4.txt
ODTK_DBG2=1 python tools/conv_bench.py $L fwd,dgrad 20 0 2>&1 | grep -v amdgpu.ids | tee $O/nst3.txt
