#!/usr/bin/env python
"""Writes tests/golden/ssd300_bf16_mock_b32.json: what bf16 STORAGE alone costs on SSD300 at batch 32 -- the comparison of
tests/test_gpu_ssd300_b32.py (bf16 engine vs f32 engine, same weights, same batch) run on the CPU with tests/mock_ops.py
(torch f32 math; in bf16 mode every stored activation / operand is rounded to bf16).  No GPU, ~4 min on 8 cores."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                    # noqa: E402
import mock_ops                                 # noqa: E402
import test_gpu_ssd300_b32 as T                 # noqa: E402
from oracle import ssd300_ref as R              # noqa: E402

torch.set_num_threads(len(os.sched_getaffinity(0)))
T.CONFIG['device'] = 'cpu'
with mock_ops.installed():
    p = R.init_params(5)
    imgs, gt = R.synthetic_batch(T.B, 77)
    ms = {}
    for dt in ('f32', 'bf16'):
        m = T._model(dt, use_graph=False)
        m.load_oracle_params(p)
        m.set_batch(imgs, gt)
        m._step_front()
        m._backward()
        ms[dt] = m
    out = {'_what': 'SSD300 batch 32, oracle seed 5, synthetic_batch(32, 77): bf16-STORAGE arithmetic on the CPU (tests/mock_ops.py) against '
                    'the same mock in f32.  activation = relative Frobenius error per layer; gradient = [cosine, norm ratio] per parameter.',
           'activation': {k: round(v, 4) for k, v in T.activation_errors(ms['f32'], ms['bf16']).items()},
           'gradient': {k: [round(c, 4), round(r, 3)] for k, (c, r) in T.gradient_report(ms['f32'], ms['bf16']).items()}}
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'ssd300_bf16_mock_b32.json'), 'w'), indent=0)
print(out)
