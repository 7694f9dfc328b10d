#!/usr/bin/env python
"""Writes tests/golden/ssd300_bf16_mock_small.json: what bf16 STORAGE alone costs on SSD300 at the SMALL batches of smoke() (__graft_entry__.py) and of
tests/test_gpu_ssd300.py::test_train_step_parity -- the bf16-engine-vs-f32-engine comparison (same weights, same batch, one forward + loss + backward) run on the CPU with
tests/mock_ops.py (torch f32 math; in bf16 mode every stored activation / operand is rounded to bf16), per configuration:
gradient = [cosine, norm ratio] per parameter, loss = [f32, bf16].  No GPU; tools/calib_bf16_mock_b32.py is the batch-32 counterpart.
    python tools/calib_bf16_mock_small.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                    # noqa: E402
import mock_ops                                 # noqa: E402
import test_gpu_ssd300_b32 as T                 # noqa: E402
from oracle import ssd300_ref as R              # noqa: E402

CONFIGS = {'smoke': (8, 1, 124), 'test_train_step_parity': (8, 5, 40)}       # name -> (batch, R.init_params seed, R.synthetic_batch seed)


def run(batch, pseed, dseed):
    p = R.init_params(pseed)
    imgs, gt = R.synthetic_batch(batch, dseed)
    prov = {'data_shape': [300, 300, 3], 'num_train': batch, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    ms = {}
    import odtk
    for dt in ('f32', 'bf16'):
        m = odtk.SSD300(dict(T.CONFIG, compute_dtype=dt, batch_size=batch, use_graph=False, device='cpu'), prov)
        m.load_oracle_params(p)
        m.set_batch(imgs, gt)
        m._step_front()
        m._backward()
        ms[dt] = m
    lf, lb = (float(ms[k].loss_parts[:, 3].sum()) / batch for k in ('f32', 'bf16'))
    return {'batch': batch, 'param_seed': pseed, 'data_seed': dseed, 'loss': [round(lf, 5), round(lb, 5)],
            'gradient': {k: [round(c, 4), round(r, 3)] for k, (c, r) in T.gradient_report(ms['f32'], ms['bf16']).items()}}


if __name__ == '__main__':
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    out = {'_what': 'SSD300 at small batches: bf16-STORAGE arithmetic on the CPU (tests/mock_ops.py) against the same mock in f32; '
                    'gradient = [cosine, norm ratio] per parameter (tools/calib_bf16_mock_small.py)'}
    with mock_ops.installed():
        for name, (b, ps, ds) in CONFIGS.items():
            out[name] = run(b, ps, ds)
            print(name, out[name]['loss'], {k: v[0] for k, v in out[name]['gradient'].items()}, flush=True)
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'ssd300_bf16_mock_small.json'), 'w'), indent=0)
