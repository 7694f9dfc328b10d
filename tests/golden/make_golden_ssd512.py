#!/usr/bin/env python
"""Generates tests/golden/ssd512.npz and ssd512_variables.json from the REFERENCE's own SSD512.py executed on the eager TF-1.x shim
(the file's one-line syntax defect at :41-43 repaired in memory, exactly as for SSD300.py): the 24 912 priors of its _get_abbox with its own
scale / aspect tables (:116-125), the graph's variables, and ONE training step of the whole class at batch 1 (loss, a subsample of
parameters after the step).  Parameters: oracle/ssd512_ref.init_params(11) pushed into the shim's variables.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_ssd512.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ssd512_ref as R5           # noqa: E402
from oracle import ssd300_ref as R            # noqa: E402
from oracle import tf_shim                    # noqa: E402
import make_golden as MG                      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['conv1_1.w', 'conv4_3.b', 'conv7.gamma', 'conv11_2.w', 'conv12_1.w', 'conv12_2.w', 'conv12_2.gamma', 'pred5.w', 'pred7.w', 'pred7.beta', 'l2norm.gamma',
        'conv12_2.mmean', 'pred7.mvar']


def main():
    p = R5.init_params(11)
    tf_shim.install(MG.vgg_tensors(p))
    ref = tf_shim.load_reference_ssd300('/root/reference/SSD512.py')
    me = MG._Self()
    me.input_size = 512
    s = [0.07 * 512]
    s = s + [(0.15 + (0.9 - 0.15) / 5 * (i - 1)) * 512 for i in range(1, 8)]              # SSD512.py:116-118 (the driver of _get_abbox)
    s = [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 7)]
    fs = R5.feature_sizes()
    outs = [[], [], [], []]
    for lvl in range(7):
        r = ref.SSD512._get_abbox(me, s[lvl], R5.ASPECTS[lvl], [2, fs[lvl], fs[lvl], 1])
        for o, v in zip(outs, r):
            o.append(v)
    anchors = tuple(torch.cat(o, 0) for o in outs)
    assert anchors[0].shape[0] == 24912
    out = dict(y1x1=anchors[0].numpy(), y2x2=anchors[1].numpy(), yx=anchors[2].numpy(), hw=anchors[3].numpy())

    imgs, gt = R5.synthetic_batch(1, 300)

    class _It:
        def get_next(self):
            return tf_shim.wrap(imgs.clone()), tf_shim.wrap(gt.clone())
    prov = {'data_shape': [512, 512, 3], 'num_train': 1, 'num_val': 0, 'train_generator': (lambda: None, _It()), 'val_generator': None}
    m = ref.SSD512(dict(MG.CONFIG, mode='train', batch_size=1), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'ssd512_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    with R5.tables():                                            # push / pull walk the oracle's layer tables
        MG.EXTRA = [e[0] for e in R5.EXTRA_LAYERS]
        saved = MG.push_params.__globals__.get('_NH', None)
        MG.push_params(p, n_heads=7)
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.01, m.is_training: True})
        after = MG.pull_params(n_heads=7)
    out['loss'] = np.asarray([float(loss)], np.float64)
    for k in KEEP:
        flat = np.asarray(after[k]).reshape(-1)
        out[k.replace('.', '__')] = flat[::max(1, flat.size // 1024)].copy()
    np.savez_compressed(os.path.join(OUT, 'ssd512.npz'), **out)
    print('priors', anchors[0].shape, 'variables', len(variables), 'loss', float(loss))
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
