#!/usr/bin/env python
"""Generates tests/golden/centernet_*.npz and fcos_*.npz by executing the REFERENCE's own code on the eager
TF-1.x shim (oracle/tf_shim):
  * CenterNet._compute_one_image_loss / _keypoints_loss / _gaussian_radius      (CenterNet.py:187-270), called unbound;
  * the inline inference branch of CenterNet._build_graph                        (CenterNet.py:159-185) and of
    FCOS._build_graph (FCOS.py:197-246, up to the per-class NMS loop), whose SOURCE LINES are read from
    /root/reference at generation time, dedented and exec'd with synthetic head outputs in scope (never copied
    into this repository);
  * FCOS._compute_one_image_loss (FCOS.py:266-348) and the level assignment lines (FCOS.py:153-189).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_centernet_fcos.py
tests/test_oracle_golden.py checks oracle/centernet_ref.py and oracle/fcos_ref.py against the fixtures.
"""
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import centernet_ref as CR     # noqa: E402
from oracle import fcos_ref as FR          # noqa: E402
from oracle import tf_shim                 # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_lines(path, first, last):
    """source lines [first, last] (1-based, inclusive) of a reference file, dedented"""
    src = open(path).read().split('\n')[first - 1:last]
    return textwrap.dedent('\n'.join(src))


def centernet(tf):
    ref = tf_shim.load_reference_module('/root/reference/CenterNet.py', 'reference_CenterNet')

    class Me:
        num_classes = 20
        data_format = 'channels_last'
        score_threshold = 0.1
        top_k_results_output = 100
    me = Me()
    for name in ('_keypoints_loss', '_gaussian_radius', '_max_pooling'):
        setattr(me, name, (lambda n: (lambda *a, **k: getattr(ref.CenterNet, n)(me, *a, **k)))(name))
    H = W = 32                                   # 128 x 128 input, stride 4
    g = torch.Generator().manual_seed(5)
    kp = (torch.randn(3, H, W, 20, generator=g) * 1.5 - 2.0).half().float()
    off = torch.rand(3, H, W, 2, generator=g).half().float()
    size = (torch.rand(3, H, W, 2, generator=g) * 12).half().float()
    gt = CR.synthetic_gt(3, 128, 23)
    gt[1, 1] = gt[1, 0]; gt[1, 1, 2:4] *= 0.5; gt[1, 2:] = -1          # two boxes, same cell and class
    gt[2, 0] = torch.tensor([64., 64., 120., 110., 7.]); gt[2, 1:] = -1   # one image-sized box
    h = torch.arange(0., float(H)); w = torch.arange(0., float(W))
    mx, my = tf.meshgrid(w, h)
    losses = [float(ref.CenterNet._compute_one_image_loss(me, kp[i], off[i], size[i], gt[i], my, mx, 4.0, [H, W])) for i in range(3)]
    print('centernet one-image losses', losses)
    # inference branch, reference lines 158-185 (keypoints/offset/size/meshgrid/stride are its free variables)
    code = ref_lines('/root/reference/CenterNet.py', 159, 185)
    dets = []
    for i in range(2):
        ns = dict(tf=tf, self=me, keypoints=kp[i:i + 1].clone() + (3.0 if i else 0.0), offset=off[i:i + 1], size=size[i:i + 1],
                  meshgrid_y=my, meshgrid_x=mx, stride=4.0)
        exec(code, ns)
        dets.append([v.detach().numpy() for v in me.detection_pred])
        print('centernet decode', i, 'kept', dets[-1][0].shape[0])
    np.savez_compressed(os.path.join(OUT, 'centernet_loss.npz'), keypoints=kp.numpy().astype(np.float16), offset=off.numpy().astype(np.float16),
                        size=size.numpy().astype(np.float16), gt=gt.numpy(), loss=np.asarray(losses, np.float64),
                        **{f'det{i}_{n}': d for i, det in enumerate(dets) for n, d in zip(('scores', 'bbox', 'class_id'), det)})


def fcos(tf):
    ref = tf_shim.load_reference_module('/root/reference/FCOS.py', 'reference_FCOS')

    class Me:
        num_classes = 21                          # testFCOS-style: 20 classes + 1 (decode walks num_classes - 1)
        data_format = 'channels_last'
        nms_score_threshold = 0.1
    me = Me()
    me._compute_one_image_loss = lambda *a: ref.FCOS._compute_one_image_loss(me, *a)
    shapes = FR.level_shapes(256, 320)
    g = torch.Generator().manual_seed(11)
    N = 3
    conf = [(torch.randn(N, h, w, 21, generator=g) * 1.5 - 2.0).half().float() for h, w in shapes]
    reg = [torch.exp(torch.randn(N, h, w, 4, generator=g)).half().float() * 2 for h, w in shapes]
    cen = [torch.randn(N, h, w, 1, generator=g).half().float() for h, w in shapes]
    gt = FR.synthetic_gt(N, 256, 31)
    gt[1, 0] = torch.tensor([100., 120., 64., 64., 4.])              # size exactly 64: trains p3 AND p4
    gt[2, 0] = torch.tensor([128., 160., 250., 300., 9.]); gt[2, 1] = torch.tensor([128., 160., 250., 300., 2.]); gt[2, 2:] = -1   # area tie
    # per-level losses through the reference's own function + its level assignment lines 154-186
    code = ref_lines('/root/reference/FCOS.py', 153, 189)
    grids = {}
    for l, (h, w) in enumerate(shapes):
        gx, gy = tf.meshgrid(torch.arange(0., float(w)), torch.arange(0., float(h)))
        grids[f'grid_x{l + 3}'], grids[f'grid_y{l + 3}'] = gx, gy
    me.batch_size = N
    me.ground_truth = gt
    ns = dict(tf=tf, self=me, **grids)
    for l in range(5):
        ns[f'p{l + 3}conf'], ns[f'p{l + 3}reg'], ns[f'p{l + 3}center'] = conf[l], reg[l], cen[l]
        ns[f's{l + 3}'] = FR.STRIDES[l]
        ns[f'p{l + 3}shape'] = list(shapes[l])
    exec(code, ns)
    losses = [float(v) for v in ns['total_loss']]
    print('fcos one-image losses', losses)
    # inference branch lines 192-246 (candidates before the per-class NMS loop)
    code = ref_lines('/root/reference/FCOS.py', 197, 246)
    ns2 = dict(tf=tf, self=me, **grids)
    for l in range(5):
        ns2[f'p{l + 3}conf'], ns2[f'p{l + 3}reg'], ns2[f'p{l + 3}center'] = conf[l][:1], reg[l][:1], cen[l][:1]
        ns2[f's{l + 3}'] = FR.STRIDES[l]
    exec(code, ns2)
    pconf, pbbox = ns2['pconf'].numpy(), ns2['pbbox'].numpy()
    print('fcos candidates', pconf.shape, pbbox.shape)
    # the rest of the branch: per-class threshold + NMS loop (FCOS.py:248-265)
    me.nms_score_threshold, me.nms_max_boxes, me.nms_iou_threshold = 0.2, 10, 0.5
    exec(ref_lines('/root/reference/FCOS.py', 248, 265), ns2)
    det = [v.numpy() for v in me.detection_pred]
    print('fcos detections', det[0].shape[0])
    out = dict(gt=gt.numpy(), loss=np.asarray(losses, np.float64), shapes=np.asarray(shapes, np.int32),
               pconf=pconf[::3].copy(), pbbox=pbbox[::3].copy(), det_scores=det[0], det_bbox=det[1], det_class_id=det[2])
    for l in range(5):
        out[f'conf{l}'], out[f'reg{l}'], out[f'center{l}'] = (conf[l].numpy().astype(np.float16), reg[l].numpy().astype(np.float16),
                                                              cen[l].numpy().astype(np.float16))
    np.savez_compressed(os.path.join(OUT, 'fcos_loss.npz'), **out)


def main():
    tf = tf_shim.install()
    centernet(tf)
    fcos(tf)
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
