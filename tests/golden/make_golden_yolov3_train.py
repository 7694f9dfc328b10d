#!/usr/bin/env python
"""Generates tests/golden/yolov3_train.npz and yolov3_variables.json by constructing the REFERENCE's own YOLOv3 class in train
mode on the eager TF-1.x shim (64 x 64 input, batch 2) and running two training steps through its session
(`sess.run([train_op, loss])`, YOLOv3.py:437-450): losses, a subsample of every parameter kind after the steps, the moving
statistics; plus name / shape / dtype / trainable of every variable of the graph (the names a tf.train.Saver checkpoint holds).
The parameters of oracle/yolov3_net_ref.init_params(21) are pushed into the shim's variables in creation order first.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_yolov3_train.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tf_shim                    # noqa: E402
from oracle import yolov3_net_ref as NR       # noqa: E402
from oracle import yolov3_ref as YR           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['c0.w', 'c0.gamma', 'c1.b', 'c5.w', 'c26.beta', 'c43.w', 'c51.w', 'c52.w', 'c58.w', 'c58.gamma', 'c59.w', 'c59.beta', 'c66.b', 'c67.w',
        'c74.w', 'c74.beta', 'c0.mmean', 'c26.mvar', 'c58.mmean', 'c74.mvar']
CONFIG = {'mode': 'train', 'data_shape': [64, 64, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
          'batch_size': 2, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3, 'nms_score_threshold': 0.5,
          'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'priors': YR.PRIORS_PX}


def batches():
    out = []
    for s in (300, 301):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 64, 64, 3, generator=g) * 255).round(), YR.synthetic_gt(2, 64, s + 10, max_obj=3)))
    return out


def main():
    tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/YOLOv3.py', 'reference_YOLOv3')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'num_train': 4, 'train_generator': (lambda: None, It()), 'val_generator': None, 'num_val': 0}
    m = ref.YOLOv3(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'yolov3_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    kernels = [k for k in V if k.endswith('/kernel')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    p = NR.init_params(21)
    with torch.no_grad():
        for i, (kn, bn) in enumerate(zip(kernels, bns)):
            V[kn].copy_(p[f'c{i}.w'].permute(1, 2, 3, 0))
            V[kn[:-len('kernel')] + 'bias'].copy_(p[f'c{i}.b'])
            V[bn + '/gamma'].copy_(p[f'c{i}.gamma']); V[bn + '/beta'].copy_(p[f'c{i}.beta'])
    losses = []
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.01, m.is_training: True})
        losses.append(float(loss))
    out = dict(losses=np.asarray(losses, np.float64))
    for key in KEEP:
        i, kind = int(key[1:].split('.')[0]), key.split('.')[1]
        name = {'w': kernels[i], 'b': kernels[i][:-len('kernel')] + 'bias', 'gamma': bns[i] + '/gamma', 'beta': bns[i] + '/beta',
                'mmean': bns[i] + '/moving_mean', 'mvar': bns[i] + '/moving_variance'}[kind]
        v = V[name].detach()
        v = v.permute(3, 0, 1, 2) if kind == 'w' else v
        flat = v.contiguous().reshape(-1)
        out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'yolov3_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
