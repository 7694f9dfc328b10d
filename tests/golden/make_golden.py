#!/usr/bin/env python
"""Generates tests/golden/*.npz by executing the REFERENCE's own SSD300.py (read from
/root/reference at generation time, never copied) on the eager TF-1.x shim in oracle/tf_shim.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
The fixtures are small and committed; tests/test_oracle_golden.py checks oracle/ssd300_ref.py
against them on every run (no /root/reference needed at test time).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ssd300_ref as R           # noqa: E402
from oracle import tf_shim                   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)

CONFIG = {
    'mode': 'test', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
    'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5,
    'pretraining_weight': './vgg_16.ckpt',
}
EXTRA = [e[0] for e in R.EXTRA_LAYERS]


def vgg_tensors(p):
    t = {}
    for l in R.VGG_LAYERS:
        if isinstance(l, tuple) and l[0] + '.w' in p:          # (PFPNetR stops at conv4_3)
            n = l[0]
            key = f'vgg_16/{n.split("_")[0]}/{n}'
            t[key + '/weights'] = p[n + '.w'].permute(1, 2, 3, 0).contiguous().numpy()     # KRSC -> HWIO
            t[key + '/biases'] = p[n + '.b'].numpy()
    return t


def push_params(p, n_heads=6):
    """Copy the oracle-named parameters into the shim's variable store (batch-norm default names: numbered per enclosing variable scope)."""
    V = tf_shim.S.variables
    for scope, names in (('feature_extractor', EXTRA), ('regressor', [f'pred{i}' for i in range(1, n_heads + 1)])):
        for bn, n in enumerate(names):                   # default layer names are numbered per enclosing variable scope
            bns = 'batch_normalization' if bn == 0 else f'batch_normalization_{bn}'
            with torch.no_grad():
                V[f'{scope}/{n}/kernel'].copy_(p[n + '.w'].permute(1, 2, 3, 0))
                V[f'{scope}/{n}/bias'].copy_(p[n + '.b'])
                V[f'{scope}/{bns}/gamma'].copy_(p[n + '.gamma'])
                V[f'{scope}/{bns}/beta'].copy_(p[n + '.beta'])
                V[f'{scope}/{bns}/moving_mean'].copy_(p[n + '.mmean'])
                V[f'{scope}/{bns}/moving_variance'].copy_(p[n + '.mvar'])
    with torch.no_grad():
        V['feature_extractor/l2_norm_factor'].copy_(p['l2norm.gamma'])


def pull_params(n_heads=6):
    V = tf_shim.S.variables
    out = {}
    for l in R.VGG_LAYERS:
        if isinstance(l, tuple):
            n = l[0]
            kname = f'feature_extractor/kernel_{n}'
            if kname not in V:                       # reference typo: 'kenrel_conv2_1' (SSD300.py:212)
                kname = f'feature_extractor/kenrel_{n}'
            out[n + '.w'] = V[kname].detach().permute(3, 0, 1, 2).contiguous().numpy()
            bname = f'feature_extractor/bias_{n}'
            if bname not in V:                       # reference typo: 'bias_conv_3_1' (SSD300.py:232)
                bname = 'feature_extractor/bias_conv_3_1'
            out[n + '.b'] = V[bname].detach().numpy().copy()
    for scope, names in (('feature_extractor', EXTRA), ('regressor', [f'pred{i}' for i in range(1, n_heads + 1)])):
        for bn, n in enumerate(names):                   # default layer names are numbered per enclosing variable scope
            bns = 'batch_normalization' if bn == 0 else f'batch_normalization_{bn}'
            out[n + '.w'] = V[f'{scope}/{n}/kernel'].detach().permute(3, 0, 1, 2).contiguous().numpy()
            out[n + '.b'] = V[f'{scope}/{n}/bias'].detach().numpy().copy()
            out[n + '.gamma'] = V[f'{scope}/{bns}/gamma'].detach().numpy().copy()
            out[n + '.beta'] = V[f'{scope}/{bns}/beta'].detach().numpy().copy()
            out[n + '.mmean'] = V[f'{scope}/{bns}/moving_mean'].detach().numpy().copy()
            out[n + '.mvar'] = V[f'{scope}/{bns}/moving_variance'].detach().numpy().copy()
    out['l2norm.gamma'] = V['feature_extractor/l2_norm_factor'].detach().numpy().copy()
    return out


class _Self:
    """Minimal `self` for calling the reference's pure methods unbound."""
    input_size = 300
    num_classes = 21


def main():
    p = R.init_params(7)
    imgs, gt = R.synthetic_batch(2, 77)
    R.calibrate_bn(p, imgs, subtract_mean=False)     # test mode feeds raw pixels (reference quirk)

    # ---------------------------------------------------------------- 1. priors + per-image loss (pure functions)
    tf_shim.install(vgg_tensors(p))
    ref = tf_shim.load_reference_ssd300()
    me = _Self()
    me._smooth_l1_loss = lambda x: ref.SSD300._smooth_l1_loss(me, x)
    s = [(0.2 + (0.9 - 0.2) / 5 * (i - 1)) * 300 for i in range(1, 8)]
    s = [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 6)]
    fs = R.feature_sizes()
    outs = [[], [], [], []]
    for lvl in range(6):
        r = ref.SSD300._get_abbox(me, s[lvl], R.ASPECTS[lvl], [2, fs[lvl], fs[lvl], 1])
        for o, v in zip(outs, r):
            o.append(v)
    anchors = tuple(torch.cat(o, 0) for o in outs)
    np.savez_compressed(os.path.join(OUT, 'priors.npz'), y1x1=anchors[0].numpy(), y2x2=anchors[1].numpy(),
                        yx=anchors[2].numpy(), hw=anchors[3].numpy())
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(2, 8828, 25, generator=g).half().float()      # stored as f16: keep it exactly representable
    gts = gt.clone()
    gts[1, 1] = gts[1, 0]                                        # duplicate GT -> duplicate best anchors
    gts[1, 2:] = -1
    losses = []
    for i in range(2):
        l = ref.SSD300._compute_one_image_loss(me, pred[i, :, 21:23], pred[i, :, 23:], anchors[0], anchors[1],
                                               anchors[2], anchors[3], pred[i, :, :21], gts[i])
        losses.append(float(l))
    np.savez_compressed(os.path.join(OUT, 'one_image_loss.npz'), pred=pred.numpy().astype(np.float16),
                        gt=gts.numpy(), loss=np.asarray(losses, np.float64))
    print('priors', anchors[0].shape, 'one-image losses', losses)

    # ---------------------------------------------------------------- 2. whole class, test mode
    m = ref.SSD300(dict(CONFIG), None)
    push_params(p)
    det = {}
    for thr in (0.5, 0.2):
        m.nms_score_threshold = thr
        sc, bb, cid = m.test_one_image(imgs[:1].numpy())
        det[f'scores_{thr}'] = sc; det[f'bbox_{thr}'] = bb; det[f'class_{thr}'] = cid
        print('test mode thr', thr, 'detections', len(sc))
    np.savez_compressed(os.path.join(OUT, 'detect.npz'), seed_params=7, seed_batch=77, **det)

    # ---------------------------------------------------------------- 3. whole class, one training epoch of 2 steps
    tf_shim.install(vgg_tensors(p))
    ref = tf_shim.load_reference_ssd300()
    batches = [R.synthetic_batch(2, 100), R.synthetic_batch(2, 101)]
    state = {'i': 0}

    class _It:
        def get_next(self):
            im, g_ = batches[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g_.clone())

    def _init():
        state['i'] = 0
    prov = {'data_shape': [300, 300, 3], 'num_train': 4, 'num_val': 0, 'train_generator': (_init, _It()),
            'val_generator': None}
    m = ref.SSD300(dict(CONFIG, mode='train'), prov)
    push_params(p)
    step_losses = []
    for step in range(2):
        state['i'] = step
        _, loss = m.sess.run([m.train_op, m.loss], feed_dict={m.lr: 0.01, m.is_training: True})
        step_losses.append(float(loss))
    after = pull_params()
    keep = ['conv1_1.w', 'conv4_3.b', 'conv7.gamma', 'conv7.mmean', 'conv7.mvar', 'conv11_2.w', 'pred1.beta',
            'pred6.w', 'l2norm.gamma', 'pred3.mvar']
    np.savez_compressed(os.path.join(OUT, 'train2.npz'), losses=np.asarray(step_losses, np.float64),
                        **{k.replace('.', '__'): after[k].reshape(-1)[::37].copy() for k in keep})     # subsampled
    print('train losses', step_losses)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
