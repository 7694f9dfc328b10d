#!/usr/bin/env python
"""Generates tests/golden/yolov3_net.npz by running the REFERENCE's own network code -- YOLOv3._feature_extractor,
_darknet_block, _yolo3_header, _conv_layer, _bn (YOLOv3.py:389-417, :484-514), called unbound on the eager TF-1.x shim --
on a 64 x 64 batch of 2 with the parameters of oracle/yolov3_net_ref.init_params pushed into the shim's variables in
creation order.  Stored: the three prediction maps in training mode (batch statistics) and in inference mode (moving
statistics), the updated moving statistics of a few layers, and the gradient of a fixed scalar with respect to a few
parameters (torch autograd THROUGH the reference's graph).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_yolov3_net.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tf_shim                    # noqa: E402
from oracle import yolov3_net_ref as NR       # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
GRAD_KEYS = ['c0.w', 'c1.gamma', 'c5.w', 'c30.beta', 'c51.w', 'c58.w', 'c59.w', 'c66.gamma', 'c67.w', 'c74.w', 'c74.beta']
STAT_KEYS = ['c0', 'c26', 'c59', 'c74']


def main():
    tf = tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/YOLOv3.py', 'reference_YOLOv3')

    class Me:
        data_format = 'channels_last'
        num_classes = 20
        num_priors = 3
        final_units = 75
        is_training = True
    me = Me()
    for name in ('_feature_extractor', '_darknet_block', '_yolo3_header', '_conv_layer', '_bn'):
        setattr(me, name, (lambda n: (lambda *a, **k: getattr(ref.YOLOv3, n)(me, *a, **k)))(name))

    def network(images):                        # YOLOv3.py:82-88
        tf_shim._SCOPE_COUNT.clear()
        tf_shim.S.pending = []
        with tf.variable_scope('backone'):
            pyd1, pyd2, pyd3 = me._feature_extractor(images)
        with tf.variable_scope('head'):
            pred1, top_down = me._yolo3_header(pyd1, 1024, 'pyd1', )
            pred2, top_down = me._yolo3_header(pyd2, 256, 'pyd2', top_down)
            pred3, _ = me._yolo3_header(pyd3, 128, 'pyd3', top_down)
        return [pred1, pred2, pred3]

    g = torch.Generator().manual_seed(3)
    images = torch.rand(2, 64, 64, 3, generator=g) * 255 - torch.tensor(NR.MEAN_RGB)
    network(images)                             # creates the variables
    V = tf_shim.S.variables
    kernels = [k for k in V if k.endswith('/kernel')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    assert len(kernels) == len(bns) == 75, (len(kernels), len(bns))
    p = NR.init_params(11)
    for k in p:                                  # non-trivial moving statistics for the inference-mode run
        if k.endswith('.mmean'):
            p[k] = 0.05 * torch.randn(p[k].shape, generator=g)
        if k.endswith('.mvar'):
            p[k] = 0.5 + torch.rand(p[k].shape, generator=g)
    with torch.no_grad():
        for i, (kn, bn) in enumerate(zip(kernels, bns)):
            V[kn].copy_(p[f'c{i}.w'].permute(1, 2, 3, 0))
            V[kn[:-len('kernel')] + 'bias'].copy_(p[f'c{i}.b'])
            V[bn + '/gamma'].copy_(p[f'c{i}.gamma']); V[bn + '/beta'].copy_(p[f'c{i}.beta'])
            V[bn + '/moving_mean'].copy_(p[f'c{i}.mmean']); V[bn + '/moving_variance'].copy_(p[f'c{i}.mvar'])
    out = dict(images=images.numpy(), names=np.asarray(kernels))
    me.is_training = True
    preds = network(images)
    weights = [torch.randn(q.shape, generator=g) for q in preds]
    scalar = sum((q * w).sum() for q, w in zip(preds, weights))
    names = []
    for key in GRAD_KEYS:
        i, kind = int(key[1:].split('.')[0]), key.split('.')[1]
        names.append({'w': kernels[i], 'gamma': bns[i] + '/gamma', 'beta': bns[i] + '/beta'}[kind])
    grads = torch.autograd.grad(scalar, [V[n] for n in names])
    for key, gr in zip(GRAD_KEYS, grads):
        flat = (gr.permute(3, 0, 1, 2) if key.endswith('.w') else gr).contiguous().reshape(-1)
        out['grad_' + key.replace('.', '__')] = flat[::max(1, flat.numel() // 2048)].numpy().copy()       # subsampled: small fixture
    for l, (q, w) in enumerate(zip(preds, weights)):
        out[f'train_pred{l + 1}'] = q.detach().numpy(); out[f'weight{l + 1}'] = w.numpy()
    pend = {id(t): v for kind, t, v in tf_shim.S.pending}
    for s in STAT_KEYS:
        i = int(s[1:])
        out[f'new_mmean_{s}'] = pend[id(V[bns[i] + '/moving_mean'])].detach().numpy()
        out[f'new_mvar_{s}'] = pend[id(V[bns[i] + '/moving_variance'])].detach().numpy()
    me.is_training = False
    with torch.no_grad():
        for l, q in enumerate(network(images)):
            out[f'test_pred{l + 1}'] = q.numpy()
    np.savez_compressed(os.path.join(OUT, 'yolov3_net.npz'), **out)
    print('variables', len(V), 'pred shapes', [tuple(q.shape) for q in preds], 'first kernels', kernels[:3], kernels[52:55])
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
