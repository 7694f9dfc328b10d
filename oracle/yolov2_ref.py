"""CPU fp32 restatement of the reference's whole YOLOv2 model (TEST INFRASTRUCTURE ONLY): network, loss, training step, inference.

Follows /root/reference/YOLOv2.py (driver testYOLOv2.py: 480 x 480, five priors in cell units, scales coord 1 / noobj 1 / obj 5 / class 1):
  * input ................................ images - mean (:56-72; test mode feeds the tensor after the subtraction)
  * backbone 'backone' (Darknet-19) ...... 18 x [tf.layers.conv2d(bias) + batch norm + leaky_relu(0.1)], five 2x2 / s2 SAME max pools (:261-312);
                                           returns lrelu18 (1024 ch), the "passthrough" lrelu17 (512 ch, SAME resolution -- no reorg), stride 32
  * head ................................. 5 x [conv + BN + leaky] (1024 3x3, 512 1x1, 1024 3x3, 512 1x1, 1024 3x3), concat(passthrough, lrelu5) over the
                                           channels, 1x1 conv + BN to (classes + 5) * priors outputs, NO activation (:75-92)
  * prediction layout .................... [N, H, W, prior, (classes | y x | h w | objectness)] (:94-99)
  * loss (per image, :102-166) ........... boxes / priors in CELL units (ground truth / 32).  For every box: its cell floor(yx); IoU of the box with the
                                           five priors centred in that cell -- intersection = prod(min(y2x2) - max(y1x1)) WITHOUT a clamp at 0 (:122); arg-max
                                           prior; sigmoid CE of the yx logits against frac(yx), 0.5 * (hw - log(box_hw / prior))^2, sigmoid CE of the class
                                           logits against the one-hot label and of the objectness against 1.  No-object term: all priors of the cells that hold
                                           NO box centre; their IoU with every box is taken on a MANGLED prior box -- "yx" := y1x1, "hw" := y2x2 of the prior, then
                                           y1x1 := "yx" - "hw"/2, y2x2 := "yx" + "hw"/2 (:145-148), again without the clamp -- and sigmoid CE against 0 where the
                                           largest IoU is <= 0.6.  loss_i = coord * (yx + hw) + class * cls + obj * obj + noobj * noobj; mean over the batch.
  * optimizer ............................ + weight_decay * sum l2_loss(trainables), MomentumOptimizer(0.9) (:168-175)
  * inference (:177-201) ................. confidence = sigmoid(class) * sigmoid(obj); box centre = (cell + 0.5) + sigmoid(yx), size = prior + exp(hw)
                                           (sums, as written), corners * 32; per class score filter + tf.image.non_max_suppression
All quirks are reproduced, not repaired.  Pinned against the reference's own class run on oracle/tf_shim: tests/golden/yolov2_train.npz
(tests/golden/make_golden_yolov2.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .detect_common import per_class_nms
from .ssd300_ref import BN_EPS, BN_MOMENTUM, MEAN_RGB, conv2d_same, maxpool_same

BACKBONE = [(32, 3), 'P', (64, 3), 'P', (128, 3), (64, 1), (128, 3), 'P', (256, 3), (128, 1), (256, 3), 'P',
            (512, 3), (256, 1), (512, 3), (256, 1), (512, 3), 'P', (1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]
HEAD = [(1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]
STRIDE = 32.0
PRIORS = [[1.08, 1.19], [3.42, 4.41], [6.63, 11.38], [9.42, 5.11], [16.62, 10.52]]


def layer_specs(num_classes=20, num_priors=5):
    """[(name, cin, cout, k, leaky)] in TensorFlow's creation order: b1..b18 (backbone), h1..h5, pred"""
    s, c, i = [], 3, 0
    for l in BACKBONE:
        if l != 'P':
            i += 1
            s.append((f'b{i}', c, l[0], l[1], True)); c = l[0]
    for j, (co, k) in enumerate(HEAD):
        s.append((f'h{j + 1}', c, co, k, True)); c = co
    s.append(('pred', 512 + 1024, (num_classes + 5) * num_priors, 1, False))
    return s


def init_params(seed=0, num_classes=20, num_priors=5):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, cin, cout, k, _ in layer_specs(num_classes, num_priors):
        p[name + '.w'] = torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = torch.zeros(cout)
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(cout, generator=g)
        p[name + '.mmean'] = torch.zeros(cout)
        p[name + '.mvar'] = torch.ones(cout)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(('.mmean', '.mvar'))]


def _layer(p, name, x, training, leaky, stats, taps):
    z = conv2d_same(x, p[name + '.w'], p[name + '.b'])
    if training:
        mean = z.mean(dim=(0, 2, 3))
        var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if stats is not None:
            n = z.shape[0] * z.shape[2] * z.shape[3]
            stats[name] = (mean.detach(), var.detach() * (n / max(n - 1, 1)))
    else:
        mean, var = p[name + '.mmean'], p[name + '.mvar']
    y = (z - mean[None, :, None, None]) * (torch.rsqrt(var + BN_EPS) * p[name + '.gamma'])[None, :, None, None] + p[name + '.beta'][None, :, None, None]
    if leaky:
        y = F.leaky_relu(y, 0.1)
    if taps is not None:
        taps[name] = y
    return y


def forward(p, images_nhwc, training, stats_out=None, taps=None, subtract_mean=True, num_priors=5):
    """-> pred [N, H, W, priors, classes + 5]"""
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    i, passthrough = 0, None
    for l in BACKBONE:
        if l == 'P':
            x = maxpool_same(x, 2, 2)
        else:
            i += 1
            x = _layer(p, f'b{i}', x, training, True, stats_out, taps)
            if i == 17:
                passthrough = x
    for j in range(len(HEAD)):
        x = _layer(p, f'h{j + 1}', x, training, True, stats_out, taps)
    x = torch.cat([passthrough, x], dim=1)
    y = _layer(p, 'pred', x, training, False, stats_out, taps).permute(0, 2, 3, 1)
    n, h, w, c = y.shape
    return y.reshape(n, h, w, num_priors, c // num_priors)


def _bce(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x * z + log(1 + exp(-|x|))"""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))


def image_loss(pred, gt, priors, scales, num_classes):
    """pred [H, W, P, C + 5], gt [pad, 5] (yc, xc, h, w, class in pixels; padded with rows of -1) -> loss of one image (YOLOv2.py:102-166)"""
    coord_scale, noobj_scale, obj_scale, class_scale = scales
    H, W, P, _ = pred.shape
    C = num_classes
    pclass, pyx, phw, pobj = pred[..., :C], pred[..., C:C + 2], pred[..., C + 2:C + 4], pred[..., C + 4:]
    G = int(torch.argmin(gt[:, 0]).item())
    gn = gt[:G] / torch.tensor([STRIDE, STRIDE, STRIDE, STRIDE, 1.0])
    gyx, ghw, glab = gn[:, :2], gn[:, 2:4], gn[:, 4].to(torch.int64)
    cell = torch.floor(gyx).to(torch.int64)
    pri = torch.as_tensor(priors, dtype=torch.float32).view(1, P, 2)
    a_yx = (cell.float() + 0.5).view(-1, 1, 2).expand(-1, P, -1)                      # centres of the box's cell, for the five priors
    a_y1x1, a_y2x2 = a_yx - pri / 2, a_yx + pri / 2
    g_y1x1, g_y2x2 = (gyx - ghw / 2).unsqueeze(1), (gyx + ghw / 2).unsqueeze(1)
    inter = (torch.minimum(g_y2x2, a_y2x2) - torch.maximum(g_y1x1, a_y1x1)).prod(-1)   # no clamp (:122)
    garea = (g_y2x2 - g_y1x1).prod(-1)
    aarea = pri.prod(-1).expand(G, -1)
    iou = inter / (aarea + garea - inter)
    best = torch.argmax(iou, dim=-1)
    ar = torch.arange(G)
    r_yx, r_hw = pyx[cell[:, 0], cell[:, 1]][ar, best], phw[cell[:, 0], cell[:, 1]][ar, best]
    r_cls, r_obj = pclass[cell[:, 0], cell[:, 1]][ar, best], pobj[cell[:, 0], cell[:, 1]][ar, best]
    r_pri = pri[0][best]
    yx_loss = _bce(r_yx, gyx - torch.floor(gyx)).sum()
    hw_loss = 0.5 * ((r_hw - torch.log(ghw / r_pri)) ** 2).sum()
    class_loss = _bce(r_cls, F.one_hot(glab, C).float()).sum()
    obj_loss = _bce(r_obj, torch.ones_like(r_obj)).sum()
    # no-object term over the cells without a box centre, on the mangled prior boxes (:145-148)
    has = torch.zeros(H, W, dtype=torch.bool)
    has[cell[:, 0], cell[:, 1]] = True
    ty, tx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    c_yx = (torch.stack([ty, tx], -1) + 0.5).view(H, W, 1, 2).expand(-1, -1, P, -1)
    c_hw = pri.view(1, 1, P, 2).expand(H, W, -1, -1)
    m_yx, m_hw = c_yx - c_hw / 2., c_yx + c_hw / 2.                                    # "yx" := y1x1, "hw" := y2x2
    n_y1x1, n_y2x2 = (m_yx - m_hw / 2.)[~has].unsqueeze(1), (m_yx + m_hw / 2.)[~has].unsqueeze(1)      # [cells, 1, P, 2]
    gg1, gg2 = g_y1x1.view(1, G, 1, 2), g_y2x2.view(1, G, 1, 2)
    inter2 = (torch.minimum(gg2, n_y2x2) - torch.maximum(gg1, n_y1x1)).prod(-1)
    aarea2 = (n_y2x2 - n_y1x1).prod(-1)
    garea2 = (gg2 - gg1).prod(-1)
    agiou = (inter2 / (aarea2 + garea2 - inter2)).max(dim=1).values                    # [cells, P]
    pobj_nb = pobj[..., 0][~has]
    noobj_loss = (_bce(pobj_nb, torch.zeros_like(pobj_nb)) * (agiou <= 0.6).float()).sum()
    return coord_scale * (yx_loss + hw_loss) + class_scale * class_loss + obj_scale * obj_loss + noobj_scale * noobj_loss


def batch_loss(pred, gt, priors, scales, num_classes):
    return torch.stack([image_loss(pred[i], gt[i], priors, scales, num_classes) for i in range(pred.shape[0])]).mean()


def loss_fn(p, images_nhwc, gt, priors=PRIORS, scales=(1., 1., 5., 1.), weight_decay=1e-4, stats_out=None):
    pred = forward(p, images_nhwc, True, stats_out, num_priors=len(priors))
    C = pred.shape[-1] - 5
    data = batch_loss(pred, gt, priors, scales, C)
    l2 = sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))
    return data + weight_decay * l2, data


def train_step(p, mom, images_nhwc, gt, lr, priors=PRIORS, scales=(1., 1., 5., 1.), weight_decay=1e-4):
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    total, data = loss_fn(p, images_nhwc, gt, priors, scales, weight_decay, stats)
    total.backward()
    grads = {}
    with torch.no_grad():
        for k in names:
            grads[k] = p[k].grad.clone()
            mom[k].mul_(0.9).add_(p[k].grad)
            p[k].sub_(lr * mom[k])
            p[k].requires_grad_(False)
            p[k].grad = None
        for name, (mean, unb) in stats.items():
            p[name + '.mmean'].mul_(BN_MOMENTUM).add_(mean * (1 - BN_MOMENTUM))
            p[name + '.mvar'].mul_(BN_MOMENTUM).add_(unb * (1 - BN_MOMENTUM))
    return float(total.detach()), float(data.detach()), grads


def decode(pred0, priors):
    """pred0 [H, W, P, C + 5] -> confidence [H*W*P, C], boxes y1x1y2x2 [H*W*P, 4] in pixels (YOLOv2.py:177-186)"""
    H, W, P, E = pred0.shape
    C = E - 5
    conf = torch.sigmoid(pred0[..., :C]).reshape(-1, C) * torch.sigmoid(pred0[..., C + 4:]).reshape(-1, 1)
    ty, tx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    a_yx = (torch.stack([ty, tx], -1) + 0.5).view(H, W, 1, 2).expand(-1, -1, P, -1).reshape(-1, 2)
    a_hw = torch.as_tensor(priors, dtype=torch.float32).view(1, 1, P, 2).expand(H, W, -1, -1).reshape(-1, 2)
    yx = a_yx + torch.sigmoid(pred0[..., C:C + 2].reshape(-1, 2))
    hw = a_hw + torch.exp(pred0[..., C + 2:C + 4].reshape(-1, 2))
    return conf, torch.cat([yx - hw / 2., yx + hw / 2.], -1) * STRIDE


def synthetic_gt(n, size, seed, pad=8, max_obj=4, num_classes=20):
    """[n, pad, 5] (yc, xc, h, w, class) in pixels padded with -1; box centres in DISTINCT cells are not required (duplicates are part of the semantics)"""
    g = torch.Generator().manual_seed(seed)
    gt = -torch.ones(n, pad, 5)
    for i in range(n):
        k = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        hw = torch.rand(k, 2, generator=g) * (0.6 * size) + 0.08 * size
        yx = torch.rand(k, 2, generator=g) * (size - hw) + hw / 2
        gt[i, :k] = torch.cat([yx, hw, torch.randint(0, num_classes, (k, 1), generator=g).float()], 1)
    return gt


def test_one_image(p, images_nhwc, priors, score_thr, max_boxes, iou_thr):
    with torch.no_grad():
        pred = forward(p, images_nhwc, False, subtract_mean=False, num_priors=len(priors))
    conf, boxes = decode(pred[0], priors)
    return per_class_nms(conf, boxes, conf.shape[1], score_thr, max_boxes, iou_thr)
