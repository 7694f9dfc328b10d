timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "two_workgroups or geometries_bf16" 2>&1 | tail -4 | cut -c1-300
python tools/ab_bench.py base= no2wg=6:1 --rounds 6 --block 20 2>&1 | tail -3
