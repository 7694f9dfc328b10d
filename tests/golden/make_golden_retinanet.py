#!/usr/bin/env python
"""Generates tests/golden/retina_*.npz by executing the REFERENCE's own RetinaNet.py functions
(_get_abbox, _compute_one_image_loss with _focal_loss / _smooth_l1_loss; read from /root/reference at
generation time, never copied) on the eager TF-1.x shim in oracle/tf_shim.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_retinanet.py
tests/test_oracle_golden.py checks oracle/retinanet_ref.py against the fixtures on every run.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import retinanet_ref as RR       # noqa: E402
from oracle import tf_shim                   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class _Self:
    """Minimal `self` for calling the reference's pure methods unbound (RetinaNet.py:25-45)."""
    data_format = 'channels_last'
    num_classes = 21
    anchors = [32, 64, 128, 256, 512]
    aspect_ratios = [1, 1 / 2, 2]
    anchor_size = [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]
    num_anchors = 9
    alpha = 0.25
    gamma = 2.0


def ref_anchors(ref, me, shapes):
    outs = [[], [], [], []]
    for size, (fh, fw) in zip(me.anchors, shapes):
        r = ref.RetinaNet._get_abbox(me, size, [2, fh, fw, 1])
        for o, v in zip(outs, r):
            o.append(v)
    return tuple(torch.cat(o, 0) for o in outs)


def main():
    tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/RetinaNet.py', 'reference_RetinaNet')
    me = _Self()
    me._smooth_l1_loss = lambda x: ref.RetinaNet._smooth_l1_loss(me, x)
    me._focal_loss = lambda a, b, c, d: ref.RetinaNet._focal_loss(me, a, b, c, d)
    # 1. anchors: 320x256 (non-square: x uses the H rate, RetinaNet.py:331) and 500x500 (testretinanet.py:20)
    out = {}
    for (ih, iw) in ((320, 256), (500, 500)):
        me.data_shape = [ih, iw, 3]
        shapes = RR.pyramid_shapes(ih, iw)
        a = ref_anchors(ref, me, shapes)
        tag = f'{ih}x{iw}'
        step = 1 if a[0].shape[0] < 20000 else 7
        out[f'shapes_{tag}'] = np.asarray(shapes, np.int32)
        for n, v in zip(('y1x1', 'y2x2', 'yx', 'hw'), a):
            out[f'{n}_{tag}'] = v.numpy()[::step].copy()
        out[f'count_{tag}'] = np.int64(a[0].shape[0])
        print('anchors', tag, a[0].shape[0], 'shapes', shapes)
    np.savez_compressed(os.path.join(OUT, 'retina_anchors.npz'), **out)
    # 2. per-image loss on the 320x256 anchor set
    me.data_shape = [320, 256, 3]
    shapes = RR.pyramid_shapes(320, 256)
    anc = ref_anchors(ref, me, shapes)
    A = anc[0].shape[0]
    g = torch.Generator().manual_seed(9)
    pconf = (torch.randn(3, A, 21, generator=g) * 2).half().float()
    pbox = (torch.randn(3, A, 4, generator=g) * 0.5).half().float()
    gt = RR.synthetic_gt(3, 256, 19)
    gt[1, 1] = gt[1, 0]; gt[1, 2:] = -1              # duplicate GT -> duplicate best anchors
    gt[2, 0] = torch.tensor([160., 128., 300., 240., 3.]); gt[2, 1:] = -1     # one image-sized box
    losses = []
    for i in range(3):
        l = ref.RetinaNet._compute_one_image_loss(me, pbox[i, :, :2], pbox[i, :, 2:], anc[0], anc[1], anc[2], anc[3],
                                                  pconf[i], gt[i])
        losses.append(float(l))
    np.savez_compressed(os.path.join(OUT, 'retina_loss.npz'), pconf=pconf.numpy().astype(np.float16),
                        pbox=pbox.numpy().astype(np.float16), gt=gt.numpy(), loss=np.asarray(losses, np.float64))
    print('one-image losses', losses)
    # 3. the inference branch (RetinaNet.py:224-256, source lines read at generation time) on image 0's head outputs
    import textwrap
    tf = sys.modules['tensorflow']
    code = textwrap.dedent('\n'.join(open('/root/reference/RetinaNet.py').read().split('\n')[223:256]))
    me.nms_score_threshold, me.nms_max_boxes, me.nms_iou_threshold = 0.35, 10, 0.5
    ns = dict(tf=tf, self=me, pbbox_yx=pbox[:1, :, :2], pbbox_hw=pbox[:1, :, 2:], pconf=pconf[:1], abbox_yx=anc[2], abbox_hw=anc[3])
    exec(code, ns)
    det = [v.numpy() for v in me.detection_pred]
    print('retina detections', det[0].shape[0])
    np.savez_compressed(os.path.join(OUT, 'retina_det.npz'), scores=det[0], bbox=det[1], class_id=det[2])
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
