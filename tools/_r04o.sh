O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
for i in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/new_$i.json; python -c "import json;d=json.load(open('$O/new_$i.json'));r=d['roofline'];print('new',d['value'],d['ms_per_step'],r['frac'],r['family']['achieved'],r['by_pass'])"
ODTK_LIB=tools/probes/bin/libodtk_old.so timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/old_$i.json; python -c "import json;d=json.load(open('$O/old_$i.json'));r=d['roofline'];print('old-epilogue',d['value'],d['ms_per_step'],r['frac'],r['family']['achieved'])"
done
