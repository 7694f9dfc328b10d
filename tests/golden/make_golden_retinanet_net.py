#!/usr/bin/env python
"""Generates tests/golden/retinanet_train.npz and retinanet_variables.json by constructing the REFERENCE's own RetinaNet class
(detection graph, train mode, the configuration of testretinanet.py:22-41 at 128 x 128 / batch 2) on the eager TF-1.x shim and
running two training steps through its session (`sess.run([train_op, loss])`, RetinaNet.py:488-499): losses, a subsample of every
parameter kind after the steps, moving statistics, and the predictions of the first forward pass; plus name / shape / trainable of
every variable of the graph.  The parameters of oracle/retinanet_net_ref.init_params(31) are pushed into the shim's variables in
creation order first.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_retinanet_net.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import retinanet_net_ref as NR    # noqa: E402
from oracle import retinanet_ref as RR        # noqa: E402
from oracle import tf_shim                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['l0.w', 'l0.gamma', 'l1.b', 'l1.gamma', 'l3.w', 'l4.w', 'l4.beta', 'l30.w', 'l63.w', 'l64.b', 'l65.w', 'l66.w', 'l67.gamma', 'l69.w', 'l71.w',
        'l76.w', 'l76.b', 'l81.w', 'l116.w', 'l121.b', 'l0.mmean', 'l1.mvar', 'l65.mmean', 'l121.mvar']
CONFIG = {'is_bottleneck': True, 'residual_block_list': [3, 4, 6, 3], 'init_conv_filters': 16, 'mode': 'train', 'is_pretraining': False,
          'data_shape': [128, 128, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'data_format': 'channels_last', 'batch_size': 2,
          'gamma': 2.0, 'alpha': 0.25, 'nms_score_threshold': 0.8, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45}


def batches():
    out = []
    for s in (500, 501):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 128, 128, 3, generator=g) * 255).round(), RR.synthetic_gt(2, 128, s + 10)))
    return out


def main():
    tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/RetinaNet.py', 'reference_RetinaNet')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'num_train': 4, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    m = ref.RetinaNet(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'retinanet_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    kernels = [k for k in V if k.endswith('/kernel')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    assert len(kernels) == len(bns) == 122, (len(kernels), len(bns))
    p = NR.init_params(31)
    with torch.no_grad():
        for i, (kn, bn) in enumerate(zip(kernels, bns)):
            V[kn].copy_(p[f'l{i}.w'].permute(1, 2, 3, 0))
            V[kn[:-len('kernel')] + 'bias'].copy_(p[f'l{i}.b'])
            V[bn + '/gamma'].copy_(p[f'l{i}.gamma']); V[bn + '/beta'].copy_(p[f'l{i}.beta'])
    losses = []
    out = dict(names=np.asarray(kernels))
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.01, m.is_training: True})
        losses.append(float(loss))
        if step:
            continue
        # parameters after the FIRST step only: with batch norms over 2 samples (p7 is 1 x 1 at this input size) the second
        # update is dominated by float chaos (8 % on the stem kernel between two correct implementations)
        for key in KEEP:
            i, kind = int(key[1:].split('.')[0]), key.split('.')[1]
            name = {'w': kernels[i], 'b': kernels[i][:-len('kernel')] + 'bias', 'gamma': bns[i] + '/gamma', 'beta': bns[i] + '/beta',
                    'mmean': bns[i] + '/moving_mean', 'mvar': bns[i] + '/moving_variance'}[kind]
            v = V[name].detach()
            v = v.permute(3, 0, 1, 2) if kind == 'w' else v
            flat = v.contiguous().reshape(-1)
            out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    out['losses'] = np.asarray(losses, np.float64)
    np.savez_compressed(os.path.join(OUT, 'retinanet_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses)
    print(kernels[:6], kernels[64:68], kernels[72:74])
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
