#!/bin/bash
mkdir -p gpurun_out/r04x
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "batchnorm or colsum" 2>&1 | tail -3 > gpurun_out/r04x/kernels.log
cat gpurun_out/r04x/kernels.log
timeout 900 python -m pytest tests/test_gpu_retinanet_model.py -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --config yolov3 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r04x/bench_yolov3.json
python - <<'PY'
import json
for n in ('yolov3',):
    j=json.load(open(f'gpurun_out/r04x/bench_{n}.json'))
    print(n, j['value'], j['ms_per_step'], j['roofline']['family'])
PY
