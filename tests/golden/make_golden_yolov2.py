#!/usr/bin/env python
"""Generates tests/golden/yolov2_train.npz, yolov2_variables.json and yolov2_names.json by constructing the REFERENCE's own YOLOv2 class (train mode,
testYOLOv2.py's scales and priors, 416 x 416, batch 2) on the eager TF-1.x shim and running two training steps through its session (losses, parameter
subsamples after the FIRST step, moving statistics, variable names / shapes), and its test graph once (detections of one picture under calibrated moving
statistics).  The parameters of oracle/yolov2_ref.init_params(81) are pushed into the shim's variables in creation order first.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_yolov2.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import yolov2_ref as YR           # noqa: E402
from oracle import tf_shim                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = ['b1.w', 'b1.gamma', 'b5.w', 'b13.beta', 'b17.w', 'b18.w', 'h1.w', 'h5.gamma', 'pred.w', 'pred.gamma', 'pred.beta', 'b3.mmean', 'h2.mvar', 'pred.mmean']
CONFIG = {'mode': 'train', 'is_pretraining': False, 'data_shape': [416, 416, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'data_format': 'channels_last', 'batch_size': 2, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1.,
          'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'rescore_confidence': False, 'priors': YR.PRIORS}


def batches():
    out = []
    for s in (1000, 1001):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, 416, 416, 3, generator=g) * 255).round(), YR.synthetic_gt(2, 416, s + 10, pad=8, max_obj=4)))
    return out


def push(V, p, specs):
    kernels = [k for k in V if k.endswith('/kernel')]
    biases = [k for k in V if k.endswith('/bias')]
    bns = [k[:-len('/gamma')] for k in V if k.endswith('/gamma')]
    assert len(kernels) == len(biases) == len(bns) == len(specs) == 24, (len(kernels), len(biases), len(bns))
    tfname = {}
    with torch.no_grad():
        for s, kn, bi, bn in zip(specs, kernels, biases, bns):
            V[kn].copy_(p[s[0] + '.w'].permute(1, 2, 3, 0)); V[bi].copy_(p[s[0] + '.b'])
            tfname[s[0] + '.w'], tfname[s[0] + '.b'] = kn, bi
            for a, b in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
                V[f'{bn}/{b}'].copy_(p[f'{s[0]}.{a}'])
                tfname[f'{s[0]}.{a}'] = f'{bn}/{b}'
    return tfname, kernels, bns


def main():
    p = YR.init_params(81)
    tf_shim.install({})
    ref = tf_shim.load_reference_module('/root/reference/YOLOv2.py', 'reference_YOLOv2')
    data = batches()
    state = {'i': 0}

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'data_shape': [416, 416, 3], 'num_train': 4, 'num_val': 0, 'train_generator': (None, It()), 'val_generator': None}
    m = ref.YOLOv2(dict(CONFIG), prov)
    V = tf_shim.S.variables
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    with open(os.path.join(OUT, 'yolov2_variables.json'), 'w') as f:
        json.dump(variables, f, indent=0, sort_keys=True)
    specs = YR.layer_specs()
    tfname, kernels, bns = push(V, p, specs)
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
    json.dump(tfname, open(os.path.join(OUT, 'yolov2_names.json'), 'w'), indent=0, sort_keys=True)
    tf_shim.uninstall()
    # ---- test graph: detections of one picture, moving statistics calibrated on it so that the scores are not all below the threshold
    q = YR.init_params(83)
    g = torch.Generator().manual_seed(1100)
    img = (torch.rand(1, 416, 416, 3, generator=g) * 255).round()
    stats = {}
    with torch.no_grad():
        YR.forward(q, img, True, stats_out=stats, subtract_mean=False)
    for name, (mean, unb) in stats.items():
        q[name + '.mmean'], q[name + '.mvar'] = mean.clone(), unb.clone()
    q['pred.beta'] = q['pred.beta'] + 1.5                      # lift the logits: some confidences above the 0.5 of the driver
    tf_shim.install({})
    ref = tf_shim.load_reference_module('/root/reference/YOLOv2.py', 'reference_YOLOv2_test')
    mt = ref.YOLOv2(dict(CONFIG, mode='test'), None)
    push(tf_shim.S.variables, q, specs)
    scores, bbox, cid = mt.test_one_image(tf_shim.wrap(img.clone()))
    out['det_scores'], out['det_bbox'], out['det_class'] = np.asarray(scores), np.asarray(bbox).reshape(-1, 4), np.asarray(cid)
    tf_shim.uninstall()
    np.savez_compressed(os.path.join(OUT, 'yolov2_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'losses', losses, 'detections', len(out['det_scores']))


if __name__ == '__main__':
    main()
