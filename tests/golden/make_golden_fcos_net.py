#!/usr/bin/env python
"""Generates tests/golden/fcos_train.npz and fcos_variables.json by constructing the REFERENCE's own FCOS class (train mode, the
configuration of testfcos.py:20-32 at 128 x 160 / batch 2) on the eager TF-1.x shim and running two training steps through its session
(`sess.run([train_op, loss])`, FCOS.py:401-412): losses, a subsample of every parameter kind after the FIRST step; plus name / shape /
trainable of every variable of the graph.  The parameters of oracle/fcos_net_ref.init_params(41) are pushed into the shim's variables
in creation order first (k-th conv kernel <-> k-th group norm).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_fcos_net.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fcos_net_ref as NR         # noqa: E402
from oracle import fcos_ref as FR             # noqa: E402
from oracle import tf_shim                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['l0.w', 'l0.gamma', 'l1.b', 'l1.gamma', 'l3.w', 'l4.w', 'l4.beta', 'l30.w', 'l64.w', 'l65.w', 'l67.b', 'l68.w', 'l70.w', 'l74.w', 'l75.w', 'l79.w',
        'l79.b', 'l80.w', 'l80.b', 'l81.gamma', 'l84.beta', 'l85.w', 'l85.gamma', 'l85.b']
CONFIG = {'mode': 'train', 'data_shape': [128, 160, 3], 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45}


def batches():
    out = []
    for s in (600, 601):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 128, 160, 3, generator=g) * 255).round(), FR.synthetic_gt(2, 128, s + 10)))
    return out


def main():
    tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/FCOS.py', 'reference_FCOS')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'num_train': 4, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    m = ref.FCOS(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'fcos_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    kernels = [k for k in V if k.endswith('/kernel')]
    gns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    assert len(kernels) == len(gns) == 86, (len(kernels), len(gns))        # 75 + ONE set of 11 head layers shared by the levels
    p = NR.init_params(41)
    with torch.no_grad():
        for i, (kn, gn) in enumerate(zip(kernels, gns)):
            V[kn].copy_(p[f'l{i}.w'].permute(1, 2, 3, 0))
            V[kn[:-len('kernel')] + 'bias'].copy_(p[f'l{i}.b'])
            V[gn + '/gamma'].copy_(p[f'l{i}.gamma']); V[gn + '/beta'].copy_(p[f'l{i}.beta'])
    losses = []
    out = dict(names=np.asarray(kernels), gn_names=np.asarray(gns))
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.001, m.is_training: True})
        losses.append(float(loss))
        if step:
            continue
        for key in KEEP:
            i, kind = int(key[1:].split('.')[0]), key.split('.')[1]
            name = {'w': kernels[i], 'b': kernels[i][:-len('kernel')] + 'bias', 'gamma': gns[i] + '/gamma', 'beta': gns[i] + '/beta'}[kind]
            v = V[name].detach()
            v = v.permute(3, 0, 1, 2) if kind == 'w' else v
            flat = v.contiguous().reshape(-1)
            out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    out['losses'] = np.asarray(losses, np.float64)
    np.savez_compressed(os.path.join(OUT, 'fcos_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses)
    print(kernels[:3], gns[:6], gns[65:70], gns[75:78])
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
