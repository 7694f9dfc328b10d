set -u
O=gpurun_out/r04e; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_v3_engine or geometries_bf16" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
python tools/conv_bench.py conv2_1,conv2_2,conv1_2 wgrad 20 0,2:131072 2>&1 | tee $O/conv_bench_wgrad.txt
python tools/conv_bench.py conv4_1,conv6,pred1 fwd,dgrad 20 0,2:268435456 2>&1 | tee $O/conv_bench_192.txt
