#!/usr/bin/env python
"""Generates tests/golden/yolov3_loss.npz by executing the REFERENCE's own YOLOv3 source on the eager TF-1.x shim
(oracle/tf_shim): the head split + priors (YOLOv3.py:99-113), the whole per-image training loss loop (:116-311) and
the inference candidates (:320-350) are read from /root/reference at generation time, dedented and exec'd with
synthetic head outputs in scope; _get_priors / _get_normlized_gn are called unbound.  Nothing is copied into this
repository.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_yolov3.py
tests/test_oracle_golden.py checks oracle/yolov3_ref.py against the fixture.
"""
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import yolov3_ref as YR        # noqa: E402
from oracle import tf_shim                 # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_lines(path, first, last):
    src = open(path).read().split('\n')[first - 1:last]
    return textwrap.dedent('\n'.join(src))


def main():
    tf = tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/YOLOv3.py', 'reference_YOLOv3')
    C, P, N, S = 20, 3, 4, 160                    # 160 x 160 input -> 5 / 10 / 20 grids

    class Me:
        num_classes = C
        num_priors = P
        coord_sacle, noobj_scale, obj_scale, class_scale = 1., 1., 5., 1.      # testYOLOv3.py:25-28 (sic: coord_sacle)
        stride = [8., 16., 32.]
        batch_size = N
        mode = 'train'
    me = Me()
    me.priors = [tf.reshape(tf.constant(YR.PRIORS_PX[i], dtype=tf.float32) / me.stride[i], [1, 1, -1, 2]) for i in range(3)]   # :39-41
    me._get_priors = lambda *a: ref.YOLOv3._get_priors(me, *a)
    me._get_normlized_gn = lambda *a: ref.YOLOv3._get_normlized_gn(me, *a)
    g = torch.Generator().manual_seed(21)
    grids = [S // 32, S // 16, S // 8]
    preds = [(torch.randn(N, h, h, P, C + 5, generator=g) * 1.2).half().float() for h in grids]
    gt = YR.synthetic_gt(N, S, 13)
    gt[1, 1] = gt[1, 0]; gt[1, 1, 4] = 3.; gt[1, 2:] = -1            # two boxes, same cell, same best prior
    gt[2, 0] = torch.tensor([80., 80., 150., 140., 5.]); gt[2, 1:] = -1   # one image-sized box
    me.ground_truth = gt
    ns = dict(tf=tf, self=me)
    for l in range(3):
        ns[f'pred{l + 1}'] = preds[l]
        ns[f'p{l + 1}shape'] = [N, grids[l], grids[l], P * (C + 5)]
    exec(ref_lines('/root/reference/YOLOv3.py', 99, 113), ns)
    exec(ref_lines('/root/reference/YOLOv3.py', 116, 310), ns)
    losses = [float(v) for v in ns['total_loss']]
    print('yolov3 one-image losses', losses)
    # inference candidates (one image)
    ns2 = dict(tf=tf, self=me)
    for l in range(3):
        ns2[f'pred{l + 1}'] = preds[l][:1]
        ns2[f'p{l + 1}shape'] = [1, grids[l], grids[l], P * (C + 5)]
    exec(ref_lines('/root/reference/YOLOv3.py', 99, 113), ns2)
    exec(ref_lines('/root/reference/YOLOv3.py', 320, 350), ns2)
    conf, box = ns2['confidence'].numpy(), ns2['bbox_y1x1y2x2'].numpy()
    print('yolov3 candidates', conf.shape, box.shape)
    # ... and the whole branch incl. the per-class NMS loop (:351-368) at a threshold that leaves a few hundred candidates
    me.nms_score_threshold, me.nms_max_boxes, me.nms_iou_threshold = 0.45, 10, 0.5
    exec(ref_lines('/root/reference/YOLOv3.py', 351, 368), ns2)
    det = [v.numpy() for v in me.detection_pred]
    print('yolov3 detections', det[0].shape[0])
    out = dict(gt=gt.numpy(), loss=np.asarray(losses, np.float64), confidence=conf[::3].copy(), bbox=box[::3].copy(),
               det_scores=det[0], det_bbox=det[1], det_class_id=det[2])
    for l in range(3):
        out[f'pred{l + 1}'] = preds[l].numpy().astype(np.float16)
    np.savez_compressed(os.path.join(OUT, 'yolov3_loss.npz'), **out)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
