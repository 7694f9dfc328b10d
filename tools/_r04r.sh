O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "sign_bits or pool2x2" > $O/tests_k.log 2>&1; tail -4 $O/tests_k.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -q -x > $O/tests_ssd.log 2>&1; tail -4 $O/tests_ssd.log | cut -c1-300
python tools/ab_bench.py base= nobits=cfg:relu_bits=0 --rounds 6 --block 20 2>&1 | tail -4 | tee $O/ab.txt
python tools/conv_bench.py conv1_2 dgrad 20 0 2>&1 | grep conv1_2
