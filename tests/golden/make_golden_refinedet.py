#!/usr/bin/env python
"""Generates tests/golden/refinedet.npz by executing the REFERENCE's own RefineDet.py code on the eager TF-1.x shim: _get_abbox for the four levels
(anchors), _compute_one_image_loss (the two-stage ARM -> ODM loss with NMS-mined negatives) on synthetic head outputs, and the inference branch
(:189-230, source lines read from /root/reference at generation time, never copied) -- the fixtures that pin oracle/refinedet_ref.py.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_refinedet.py
"""
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refinedet_ref as FR       # noqa: E402
from oracle import tf_shim                   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class _Self:
    data_format = 'channels_last'
    num_classes = 21
    anchor_ratios = [0.5, 1.0, 2.0]
    num_anchors = 3
    batch_size = 1


def main():
    tf_shim.install()
    src = open('/root/reference/RefineDet.py').read()
    ref = tf_shim.load_reference_module('/root/reference/RefineDet.py', 'reference_RefineDet')
    me = _Self()
    me._smooth_l1_loss = lambda x: ref.RefineDet320._smooth_l1_loss(me, x)
    out = {}
    outs = [[], [], [], []]
    for f, stride in zip(FR.level_shapes(320), FR.STRIDES):
        r = ref.RefineDet320._get_abbox(me, stride * 4, stride, [1, f, f, 1])
        for o, v in zip(outs, r):
            o.append(v)
    anc = tuple(torch.cat(o, 0) for o in outs)
    A = anc[0].shape[0]
    for n, v in zip(('y1x1', 'y2x2', 'yx', 'hw'), anc):
        out[n] = v.numpy()
    g = torch.Generator().manual_seed(21)
    arm_loc = (torch.randn(3, A, 4, generator=g) * 0.3).half().float()
    arm_conf = (torch.randn(3, A, 2, generator=g) * 1.5).half().float()
    odm_loc = (torch.randn(3, A, 4, generator=g) * 0.3).half().float()
    odm_conf = (torch.randn(3, A, 21, generator=g) * 2).half().float()
    gt = FR.synthetic_gt(3, 320, 23)
    gt[1, 1] = gt[1, 0]; gt[1, 2:] = -1                       # duplicate box -> duplicate best anchors
    losses = []
    for i in range(3):
        l = ref.RefineDet320._compute_one_image_loss(me, arm_loc[i, :, :2], arm_loc[i, :, 2:], arm_conf[i], odm_loc[i, :, :2], odm_loc[i, :, 2:],
                                                     odm_conf[i], anc[0], anc[1], anc[2], anc[3], gt[i])
        losses.append(float(l))
    out.update(arm_loc=arm_loc.numpy().astype(np.float16), arm_conf=arm_conf.numpy().astype(np.float16), odm_loc=odm_loc.numpy().astype(np.float16),
               odm_conf=odm_conf.numpy().astype(np.float16), gt=gt.numpy(), loss=np.asarray(losses, np.float64))
    print('anchors', A, 'one-image losses', losses)
    # the inference branch (RefineDet.py:189-230)
    lines = src.split('\n')
    start = next(i for i, l in enumerate(lines) if 'armconft = tf.nn.softmax(armpconf[0, ...])' in l)
    end = next(i for i, l in enumerate(lines) if 'self.detection_pred = [scores, bbox, class_id]' in l)
    code = textwrap.dedent('\n'.join(lines[start:end + 1]))
    tf = sys.modules['tensorflow']
    me.nms_score_threshold, me.nms_max_boxes, me.nms_iou_threshold = 0.12, 10, 0.45
    ns = dict(tf=tf, self=me, armpconf=arm_conf[:1], odmpconf=odm_conf[:1], armpbbox_yx=arm_loc[:1, :, :2], armpbbox_hw=arm_loc[:1, :, 2:],
              odmpbbox_yx=odm_loc[:1, :, :2], odmpbbox_hw=odm_loc[:1, :, 2:], abbox_yx=anc[2], abbox_hw=anc[3])
    exec(code, ns)
    det = [v.numpy() for v in me.detection_pred]
    out.update(det_scores=det[0], det_bbox=det[1], det_class=det[2])
    print('refinedet detections', det[0].shape[0])
    np.savez_compressed(os.path.join(OUT, 'refinedet.npz'), **out)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
