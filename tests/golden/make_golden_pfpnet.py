#!/usr/bin/env python
"""Generates tests/golden/pfpnet_train.npz, pfpnet_variables.json and pfpnet_names.json by constructing the REFERENCE's own PFPNetR class (train mode,
input 320, batch 2) on the eager TF-1.x shim and running two training steps through its session: losses, a subsample of every parameter kind after the
FIRST step, moving statistics; plus name / shape / trainable of every variable of the graph.  The parameters of oracle/pfpnet_net_ref.init_params(71) are
pushed into the shim's variables in creation order first (the VGG trunk through the shim's NewCheckpointReader, as the reference initialises it).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_pfpnet.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import pfpnet_net_ref as PR       # noqa: E402
from oracle import refinedet_ref as FR        # noqa: E402
from oracle import tf_shim                    # noqa: E402
import make_golden as MG                      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['conv1_1.w', 'conv3_2.b', 'conv4_3.w', 'fl1.w', 'fl1.gamma', 'fl4.w', 'fl2_1d.w', 'fl2_1c.w', 'fl3_1d.gamma', 'fl4_1c.beta', 'fl4_3d.w', 'fl1_2.w', 'fl1_4.w',
        'fl3_4.beta', 'arm1.c1.w', 'arm1.loc.w', 'arm2.conf.w', 'arm4.conf.gamma', 'tcb4.c1.w', 'tcb4.c2.w', 'tcb3.d.w', 'tcb1.c2.beta', 'tcb1.d.w', 'odm1.c1.w',
        'odm1.loc.w', 'odm3.conf.w', 'odm4.conf.beta', 'feat1_l2_norm', 'feat2_l2_norm', 'fl3.mmean', 'fl4_2d.mvar', 'odm2.conf.mmean']
CONFIG = {'mode': 'train', 'input_size': 320, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 2,
          'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': './vgg_16.ckpt'}


def batches():
    out = []
    for s in (900, 901):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 320, 320, 3, generator=g) * 255).round(), FR.synthetic_gt(2, 320, s + 10, pad=8, max_obj=4)))
    return out


def main():
    p = PR.init_params(71)
    tf_shim.install(MG.vgg_tensors(p))
    ref = tf_shim.load_reference_module('/root/reference/PFPNetR.py', 'reference_PFPNetR')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'data_shape': [320, 320, 3], 'num_train': 4, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    m = ref.PFPNetR(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'pfpnet_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    specs = PR.layer_specs()
    kernels = [k for k in V if k.endswith('/kernel') or k.split('/')[-1].startswith(('kernel_', 'kenrel_'))]
    biases = [k for k in V if k.endswith('/bias') or k.split('/')[-1].startswith('bias_')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    bn_specs = [s for s in specs if s[1] != 'vgg']
    assert len(kernels) == len(biases) == len(specs) == 91 and len(bns) == len(bn_specs) == 81, (len(kernels), len(biases), len(bns))
    tfname = {}
    with torch.no_grad():
        for s, kn, bn_ in zip(specs, kernels, biases):
            assert ('transpose' in kn) == (s[1] == 'dconv'), (s[0], kn)
            assert tuple(V[kn].shape) == tuple(p[s[0] + '.w'].permute(1, 2, 3, 0).shape), (s[0], kn, tuple(V[kn].shape))
            V[kn].copy_(p[s[0] + '.w'].permute(1, 2, 3, 0)); V[bn_].copy_(p[s[0] + '.b'])
            tfname[s[0] + '.w'], tfname[s[0] + '.b'] = kn, bn_
        for s, bn in zip(bn_specs, bns):
            V[bn + '/gamma'].copy_(p[s[0] + '.gamma']); V[bn + '/beta'].copy_(p[s[0] + '.beta'])
            for a, b in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
                tfname[f'{s[0]}.{a}'] = f'{bn}/{b}'
        for k in ('feat1_l2_norm', 'feat2_l2_norm'):
            V['feature_extractor/' + k].copy_(p[k]); tfname[k] = 'feature_extractor/' + k
    losses = []
    out = dict(names=np.asarray(kernels), bn_names=np.asarray(bns))
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.001, m.is_training: True})
        losses.append(float(loss))
        if step:
            continue
        for key in KEEP:
            v = V[tfname[key]].detach()
            v = v.permute(3, 0, 1, 2) if key.endswith('.w') else v
            flat = v.contiguous().reshape(-1)
            out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    out['losses'] = np.asarray(losses, np.float64)
    json.dump(tfname, open(os.path.join(OUT, 'pfpnet_names.json'), 'w'), indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'pfpnet_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
