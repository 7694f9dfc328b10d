#!/bin/bash
# x3 engine: kernel parity, model parity, gate, RetinaNet bench
mkdir -p gpurun_out/r04u
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "x3 or colsum" 2>&1 | tail -5 > gpurun_out/r04u/kernels.log
timeout 900 python -m pytest tests/test_gpu_retinanet_model.py -q -m gpu -k "f32_model_matches" -s 2>&1 | grep -v "^E   \|where" | tail -30 > gpurun_out/r04u/model.log
timeout 900 python tools/bf16_after_training.py retinanet x3 > gpurun_out/r04u/gate.log 2>&1
timeout 600 python bench.py --config retinanet --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r04u/bench_x3.json
cat gpurun_out/r04u/kernels.log; cat gpurun_out/r04u/model.log; tail -4 gpurun_out/r04u/gate.log
python - <<'PY'
import json
for n in ('x3',):
    try:
        j=json.load(open(f'gpurun_out/r04u/bench_{n}.json'))
        print(n, j['value'], j['ms_per_step'], j.get('roofline',{}).get('by_pass'))
    except Exception as e:
        print(n, 'failed', e)
PY
