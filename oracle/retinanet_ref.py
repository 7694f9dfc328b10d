"""CPU fp32 restatement of the RetinaNet box side of the reference (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/RetinaNet.py:
  * anchors ............................. _get_abbox            RetinaNet.py:328-355, anchor table :39-42
  * matching + focal + smooth-L1 ........ _compute_one_image_loss RetinaNet.py:357-452
  * focal loss, softmax flavour ......... _focal_loss           RetinaNet.py:457-474 (alpha for pos AND neg,
                                          p clipped to [1e-8, 1], sum over pos+neg divided by #pos)
  * batch loop .......................... RetinaNet.py:194-213 (sum over images / batch_size)
  * inference decode .................... RetinaNet.py:223-256 (same arithmetic as SSD300's)
Pinned against the reference's own functions run on oracle/tf_shim: tests/golden/retina_*.npz
(tests/golden/make_golden_retinanet.py); TF kernel semantics underneath the shim are "parity unpinned"
exactly as for SSD300.  Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import numpy as np
import torch

ANCHOR_SIZES = [32, 64, 128, 256, 512]                      # RetinaNet.py:39
ASPECT_RATIOS = [1, 1 / 2, 2]                               # :40
ANCHOR_SCALES = [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]        # :41
NUM_ANCHORS = len(ASPECT_RATIOS) * len(ANCHOR_SCALES)       # :42


def level_priors(size):
    """python-double (h, w) list of one level, order r-major then s (RetinaNet.py:345-348)."""
    pr = []
    for r in ASPECT_RATIOS:
        for s in ANCHOR_SCALES:
            pr.append([s * size * (r ** 0.5), s * size / (r ** 0.5)])
    return pr


def pyramid_shapes(input_h, input_w):
    """Feature-map sizes of p3..p7: stem conv s2 + pool s2 (SAME -> ceil), then three stride-2 stages,
    then two stride-2 convs (RetinaNet.py:258-285, :139-144)."""
    c = lambda v: -(-v // 2)
    h, w = c(c(input_h)), c(c(input_w))          # stride 4
    shapes = []
    for _ in range(5):
        h, w = c(h), c(w)
        shapes.append((h, w))                    # strides 8, 16, 32, 64, 128
    return shapes


def anchors(data_shape, shapes):
    """y1x1, y2x2, yx, hw  [A, 2] float32 in the exact float32 op order of _get_abbox.
    Reference quirk (reproduced): for channels_last `input_h = self.data_shape[1]` (RetinaNet.py:330), i.e. the
    WIDTH of [H, W, C]; rate = data_shape[1] / fh is used for BOTH axes; centre = (i + 0.5) * rate."""
    input_h = data_shape[1]
    f32 = np.float32
    outs = [[], [], [], []]
    for size, (fh, fw) in zip(ANCHOR_SIZES, shapes):
        rate = f32(input_h) / f32(fh)
        ty = (np.arange(0., fh, dtype=f32).reshape(-1, 1, 1, 1) + f32(0.5))
        tx = (np.arange(0., fw, dtype=f32).reshape(1, -1, 1, 1) + f32(0.5))
        ty = np.tile(ty, [1, fw, 1, 1]) * rate
        tx = np.tile(tx, [fh, 1, 1, 1]) * rate
        tyx = np.tile(np.concatenate([ty, tx], -1), [1, 1, NUM_ANCHORS, 1])
        pr = np.asarray(level_priors(size), dtype=np.float64).astype(f32).reshape(1, 1, -1, 2)
        y1x1 = (tyx - pr / f32(2.)).reshape(-1, 2)
        y2x2 = (tyx + pr / f32(2.)).reshape(-1, 2)
        yx = y1x1 / f32(2.) + y2x2 / f32(2.)
        hw = y2x2 - y1x1
        for o, v in zip(outs, (y1x1, y2x2, yx, hw)):
            o.append(v.astype(f32))
    return tuple(torch.from_numpy(np.concatenate(o, 0)) for o in outs)


def match(anc, gt):
    """RetinaNet.py:359-417.  Integer outputs are the bit-exact contract of the GPU kernels:
    best[G] (first arg-max per GT), per remaining anchor max IoU / first arg-max GT,
    pos = IoU > 0.5, neg = IoU < 0.4, the band in between is ignored."""
    a_y1x1, a_y2x2, a_yx, a_hw = anc
    G = int(torch.argmin(gt[:, 0]).item())
    g = gt[:G]
    g_yx, g_hw = g[:, 0:2], g[:, 2:4]
    g_y1x1 = g_yx - g_hw / 2.
    g_y2x2 = g_yx + g_hw / 2.
    label = g[:, 4].to(torch.int32)
    i1 = torch.maximum(a_y1x1[None], g_y1x1[:, None])
    i2 = torch.minimum(a_y2x2[None], g_y2x2[:, None])
    inter = torch.clamp(i2 - i1, min=0).prod(dim=-1)
    aarea = a_hw.prod(dim=-1)[None].expand(G, -1)
    garea = g_hw.prod(dim=-1)[:, None]
    iou = inter / (aarea + garea - inter)
    best = torch.argmax(iou, dim=1)
    A = a_yx.shape[0]
    othermask = torch.ones(A, dtype=torch.bool)
    othermask[best] = False
    other_iou = iou.t()[othermask]
    m = other_iou.max(dim=1).values
    r = torch.argmax(other_iou, dim=1)
    return dict(G=G, g_yx=g_yx, g_hw=g_hw, label=label, best=best, othermask=othermask, max_iou=m, rgindex=r,
                pos=m > 0.5, neg=m < 0.4)


def smooth_l1(x):
    return torch.where(x.abs() < 1., 0.5 * x * x, x.abs() - 0.5)


def focal(pos_label, pos_logits, neg_label, neg_logits, alpha, gamma):
    """RetinaNet.py:457-474."""
    pp = torch.softmax(pos_logits, dim=-1).gather(1, pos_label.view(-1, 1).long()).squeeze(1).clamp(1e-8, 1.)
    pn = torch.softmax(neg_logits, dim=-1).gather(1, neg_label.view(-1, 1).long()).squeeze(1).clamp(1e-8, 1.)
    lp = -alpha * torch.pow(1. - pp, gamma) * torch.log(pp)
    ln = -alpha * torch.pow(1. - pn, gamma) * torch.log(pn)
    return (lp.sum() + ln.sum()) / float(pp.shape[0])


def one_image_loss(p_yx, p_hw, pconf, anc, gt, alpha=0.25, gamma=2.0, num_classes=21, detail=False):
    a_y1x1, a_y2x2, a_yx, a_hw = anc
    mt = match(anc, gt)
    best, om, pos, neg = mt["best"], mt["othermask"], mt["pos"], mt["neg"]
    o_pyx, o_phw, o_conf, o_ayx, o_ahw = p_yx[om], p_hw[om], pconf[om], a_yx[om], a_hw[om]
    pos_r = mt["rgindex"][pos]
    t_pyx = torch.cat([p_yx[best], o_pyx[pos]], 0)
    t_phw = torch.cat([p_hw[best], o_phw[pos]], 0)
    t_conf = torch.cat([pconf[best], o_conf[pos]], 0)
    t_label = torch.cat([mt["label"], mt["label"][pos_r]], 0)
    t_gyx = torch.cat([mt["g_yx"], mt["g_yx"][pos_r]], 0)
    t_ghw = torch.cat([mt["g_hw"], mt["g_hw"][pos_r]], 0)
    t_ayx = torch.cat([a_yx[best], o_ayx[pos]], 0)
    t_ahw = torch.cat([a_hw[best], o_ahw[pos]], 0)
    neg_conf = o_conf[neg]
    neg_label = torch.full((neg_conf.shape[0],), num_classes - 1, dtype=torch.int64)
    conf_loss = focal(t_label, t_conf, neg_label, neg_conf, alpha, gamma)
    tgt_yx = (t_gyx - t_ayx) / t_ahw
    tgt_hw = torch.log(t_ghw / t_ahw)
    coord = (smooth_l1(t_pyx - tgt_yx).sum(-1) + smooth_l1(t_phw - tgt_hw).sum(-1)).mean()
    total = conf_loss + coord
    if not detail:
        return total
    return dict(total=total, conf_loss=conf_loss, coord=coord, match=mt, num_pos=int(t_label.shape[0]),
                num_neg=int(neg_conf.shape[0]))


def batch_loss(p_yx, p_hw, pconf, anc, ground_truth, alpha=0.25, gamma=2.0):
    """RetinaNet.py:194-213: sequential sum over the images / batch_size."""
    n = pconf.shape[0]
    loss = torch.zeros(())
    for i in range(n):
        loss = loss + one_image_loss(p_yx[i], p_hw[i], pconf[i], anc, ground_truth[i], alpha, gamma)
    return loss / n


def decode_candidates(p_yx, p_hw, pconf, anc, thr, num_classes=21):
    """RetinaNet.py:224-238 for one image, dense form: conf [A, C-1] softmax scores, boxes [A, 4] y1x1y2x2,
    keep [A] = arg-max class is not the background, cand [A, C-1] = keep & conf >= thr."""
    a_yx, a_hw = anc[2], anc[3]
    conf = torch.softmax(pconf, dim=-1)
    keep = torch.argmax(conf, dim=-1) < num_classes - 1
    yx = p_yx * a_hw + a_yx
    hw = a_hw * torch.exp(p_hw)
    boxes = torch.cat([yx - hw / 2., yx + hw / 2.], -1)
    c = conf[:, :num_classes - 1]
    return c, boxes, keep, keep.unsqueeze(1) & (c >= thr)


def synthetic_gt(batch, input_size, seed, pad=60):
    """VOC-shaped ground truth [B, pad, 5] = [yc, xc, h, w, cls] px, pad rows -1 (image_augmentor.py:24-27)."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, 7, (1,), generator=g))
        h = torch.rand(n, generator=g) * (input_size * 0.8) + input_size * 0.1
        w = torch.rand(n, generator=g) * (input_size * 0.8) + input_size * 0.1
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return gt
