#!/usr/bin/env python
"""Generates tests/golden/centernet_train.npz and centernet_variables.json by constructing the REFERENCE's own CenterNet class (train mode,
the configuration of testcenternet.py:20-32 at 128 x 128 / batch 2) on the eager TF-1.x shim and running two training steps through its
session (`sess.run([train_op, loss])`, CenterNet.py:298-309): losses, a subsample of every parameter kind after the FIRST step (Adam), the
moving statistics; plus name / shape / trainable of every variable of the graph.  The shim traces BOTH branches of tf.cond for variable
creation (TRACE_DEAD_COND_BRANCHES), as TensorFlow's graph builder does: the 1x1 shortcut convs of _basic_block exist even where unused.
The parameters of oracle/centernet_net_ref.init_params(51) are pushed into the shim's variables in creation order first.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_centernet_net.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import centernet_net_ref as NR    # noqa: E402
from oracle import centernet_ref as CR        # noqa: E402
from oracle import tf_shim                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['c0.w', 'c0.gamma', 'c2.w', 'c3.w', 'c5.w', 'c8.w', 'c8.gamma', 'c9.beta', 'c25.w', 'c30.w', 'c47.w', 'c49.w', 'c50.w', 'c51.w', 'c53.gamma', 'c55.w',
        'c57.w', 'c60.w', 'c62.w', 'c63.w', 'c63.beta', 'c64.w', 'c65.w', 'c65.gamma', 'c0.mmean', 'c30.mvar', 'c51.mmean', 'c8.mmean', 'c8.mvar', 'c65.mvar']
CONFIG = {'mode': 'train', 'input_size': 128, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'batch_size': 2, 'score_threshold': 0.1, 'top_k_results_output': 100}


def batches():
    out = []
    for s in (700, 701):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 128, 128, 3, generator=g) * 255).round(), CR.synthetic_gt(2, 128, s + 10, pad=8, max_obj=4)))
    return out


def main():
    tf_shim.install()
    tf_shim.TRACE_DEAD_COND_BRANCHES = True
    sys.modules['tensorflow'].cond = tf_shim.cond
    ref = tf_shim.load_reference_module('/root/reference/CenterNet.py', 'reference_CenterNet')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'num_train': 4, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    m = ref.CenterNet(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'centernet_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    kernels = [k for k in V if k.endswith('/kernel')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    specs = NR.layer_specs()
    assert len(kernels) == len(bns) == len(specs) == 66, (len(kernels), len(bns))
    p = NR.init_params(51)
    with torch.no_grad():
        for i, (kn, bn) in enumerate(zip(kernels, bns)):
            assert ('transpose' in kn) == (specs[i][1] == 'dconv'), (i, kn)
            V[kn].copy_(p[f'c{i}.w'].permute(1, 2, 3, 0))          # conv: [co][r][s][ci] -> HWIO; transposed conv: [ci][r][s][co] -> [kh][kw][co][ci]
            V[kn[:-len('kernel')] + 'bias'].copy_(p[f'c{i}.b'])
            V[bn + '/gamma'].copy_(p[f'c{i}.gamma']); V[bn + '/beta'].copy_(p[f'c{i}.beta'])
    losses = []
    out = dict(names=np.asarray(kernels), bn_names=np.asarray(bns))
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.001, m.is_training: True})
        losses.append(float(loss))
        if step:
            continue
        for key in KEEP:
            i, kind = int(key[1:].split('.')[0]), key.split('.')[1]
            name = {'w': kernels[i], 'b': kernels[i][:-len('kernel')] + 'bias', 'gamma': bns[i] + '/gamma', 'beta': bns[i] + '/beta',
                    'mmean': bns[i] + '/moving_mean', 'mvar': bns[i] + '/moving_variance'}[kind]
            v = V[name].detach()
            v = v.permute(3, 0, 1, 2) if kind == 'w' else v
            flat = v.contiguous().reshape(-1)
            out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    out['losses'] = np.asarray(losses, np.float64)
    np.savez_compressed(os.path.join(OUT, 'centernet_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses)
    print(kernels[:4], kernels[50:54], bns[50:54], kernels[-3:])
    tf_shim.TRACE_DEAD_COND_BRANCHES = False
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
