O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pool2x2" > $O/tests_pool.log 2>&1; tail -5 $O/tests_pool.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ssd300.py tests/test_gpu_ssd300_b32.py -q -x > $O/tests_ssd.log 2>&1; tail -5 $O/tests_ssd.log | cut -c1-300
for i in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-conv-events 2>/dev/null | grep '^{' > $O/new_$i.json; python -c "import json;d=json.load(open('$O/new_$i.json'));print('fused pools',d['value'],d['ms_per_step'])"
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-conv-events --kernel-dbg 65536 2>/dev/null | grep '^{' > $O/old_$i.json; python -c "import json;d=json.load(open('$O/old_$i.json'));print('x',d['value'],d['ms_per_step'])" 
done
