"""CPU fp32 restatement of the reference's whole RetinaNet detection model (TEST INFRASTRUCTURE ONLY): network and training step.

Follows /root/reference/RetinaNet.py:
  * input ................................ images - mean (:101-118; test mode feeds the tensor after the subtraction)
  * stem ................................. conv(7x7, stride 2, `init_conv_filters`) + batch norm + ReLU, 3x3 / stride-2 max pool  :258-271
  * pre-activation units ................. every other conv is batch norm -> ReLU -> conv(bias) (_bn_activation_conv :594-619);
                                           bottleneck = 1x1 f, 3x3 f (stride), 1x1 4f  +  a 3x3 (stride) 4f shortcut conv on EVERY unit
                                           (:634-643); filters f = 7, 14, 28, 56 -- `filters_list` is built from the KERNEL SIZE 7 (:27)
  * pyramid .............................. p5 = 3x3 on the last stage; p4 / p3: 1x1 lateral + tf.image.resize_bilinear(top) (TF-1.x grid),
                                           the SUM is handed down, a 3x3 on the sum is the level's output (:303-319); p6, p7: 3x3 / stride 2 on
                                           p5, p6 (:143-144)
  * subnets .............................. per level, NOT shared: class 4 x 3x3(256) + 3x3(9 * classes, bias = -log((1 - pi) / pi), pi = .01),
                                           box 4 x 3x3(256) + 3x3(36) (:287-301), created in the order p3c, p3r, p4c, ... (:146-155)
  * predictions .......................... pconf [N, A, classes], pbbox [N, A, 4] = (yx, hw) (:321-326), levels concatenated p3..p7 (:184-186)
  * loss / optimizer ..................... sum_i loss_i / batch + wd * l2 over 'feature_extractor' and 'regressor' (:194-213), Momentum 0.9
Layers are l0 .. l121 in creation order; layer k owns conv k and batch norm k ('<layer>.w' [K,R,S,C], '.b', '.gamma', '.beta',
'.mmean', '.mvar'): l0 is conv -> BN -> ReLU, every other layer is BN -> ReLU -> conv (its BN has the conv's INPUT channels).
Pinned against the reference's own class run on oracle/tf_shim: tests/golden/retinanet_train.npz
(tests/golden/make_golden_retinanet_net.py).  Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import retinanet_ref as RR
from .augment_ref import resize_bilinear_legacy
from .ssd300_ref import BN_EPS, BN_MOMENTUM, conv2d_same, maxpool_same

MEAN_RGB = (123.68, 116.779, 103.979)
FILTERS = (7, 14, 28, 56)                      # RetinaNet.py:27 (sic)
PI = 0.01


def layer_specs(block_list=(3, 4, 6, 3), init_filters=16, num_classes=21, num_anchors=9):
    """[(name, cin, cout, k, stride, bn_channels, bias_init)] in creation order"""
    specs = []

    def add(cin, cout, k, s, bias_init=0.):
        specs.append((f'l{len(specs)}', cin, cout, k, s, cin if specs else cout, bias_init))
        return cout
    c = add(3, init_filters, 7, 2)
    stage_out = []
    for i, blocks in enumerate(block_list):
        f = FILTERS[i]
        for j in range(blocks):
            s = 2 if (i > 0 and j == 0) else 1
            add(c, f, 1, 1); add(f, f, 3, s); add(f, 4 * f, 1, 1)
            add(c, 4 * f, 3, s)                                     # the shortcut conv is created AFTER the conv branch (:636-641)
            c = 4 * f
        stage_out.append(c)
    f1, f2, f3 = stage_out[-3:]
    add(f3, 256, 3, 1)                                              # p5
    add(f2, 256, 1, 1); add(256, 256, 3, 1)                         # p4: lateral, smoothing
    add(f1, 256, 1, 1); add(256, 256, 3, 1)                         # p3
    add(256, 256, 3, 2); add(256, 256, 3, 2)                        # p6, p7
    for _ in range(5):
        for _ in range(4):
            add(256, 256, 3, 1)
        add(256, num_classes * num_anchors, 3, 1, -math.log((1 - PI) / PI))
        for _ in range(4):
            add(256, 256, 3, 1)
        add(256, 4 * num_anchors, 3, 1)
    return specs


def init_params(seed=0, **kw):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, cin, cout, k, _, bnc, bias_init in layer_specs(**kw):
        p[name + '.w'] = torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = torch.full((cout,), float(bias_init)) + (0.01 * torch.randn(cout, generator=g) if bias_init == 0. else 0.)
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(bnc, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(bnc, generator=g)
        p[name + '.mmean'] = torch.zeros(bnc)
        p[name + '.mvar'] = torch.ones(bnc)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(('.mmean', '.mvar'))]


class _Net:
    def __init__(self, p, specs, training, stats_out, relu_masks, taps):
        self.p, self.specs, self.training, self.stats, self.masks, self.taps, self.i = p, specs, training, stats_out, relu_masks, taps, 0

    def _bn_relu(self, name, x):
        p = self.p
        if self.training:
            mean = x.mean(dim=(0, 2, 3))
            var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
            if self.stats is not None:
                n = x.shape[0] * x.shape[2] * x.shape[3]
                self.stats[name] = (mean.detach(), var.detach() * (n / max(n - 1, 1)))
        else:
            mean, var = p[name + '.mmean'], p[name + '.mvar']
        y = (x - mean[None, :, None, None]) * (torch.rsqrt(var + BN_EPS) * p[name + '.gamma'])[None, :, None, None] + p[name + '.beta'][None, :, None, None]
        if self.masks is not None:                   # the linear region is dictated (see yolov3_net_ref.forward)
            y = torch.where(self.masks[name], y, torch.zeros_like(y))
        else:
            y = F.relu(y)
        if self.taps is not None:
            self.taps[name] = y
        return y

    def stem(self, x):
        name, _, _, _, stride, _, _ = self.specs[self.i]
        self.i += 1
        return self._bn_relu(name, conv2d_same(x, self.p[name + '.w'], self.p[name + '.b'], stride))

    def conv(self, x):
        """batch norm -> ReLU -> conv(bias)"""
        name, _, _, _, stride, _, _ = self.specs[self.i]
        self.i += 1
        return conv2d_same(self._bn_relu(name, x), self.p[name + '.w'], self.p[name + '.b'], stride)


def _resize(x, h, w):
    return torch.stack([resize_bilinear_legacy(img.permute(1, 2, 0), h, w).permute(2, 0, 1) for img in x])


def forward(p, images_nhwc, training, stats_out=None, subtract_mean=True, relu_masks=None, taps=None, block_list=(3, 4, 6, 3)):
    """-> pconf [N, A, classes], pbbox [N, A, 4] (yx, hw)"""
    C = _num_classes(p, block_list)
    specs = layer_specs(block_list, p['l0.w'].shape[0], C, 9)
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, specs, training, stats_out, relu_masks, taps)
    x = maxpool_same(net.stem(x), 3, 2)
    feats = []
    for i, blocks in enumerate(block_list):
        for _ in range(blocks):
            branch = net.conv(net.conv(net.conv(x)))
            x = branch + net.conv(x)
        feats.append(x)
    f1, f2, f3 = feats[-3:]
    p5 = net.conv(f3)
    lat = net.conv(f2)
    total4 = lat + _resize(p5, lat.shape[2], lat.shape[3])
    p4 = net.conv(total4)
    lat = net.conv(f1)
    total3 = lat + _resize(total4, lat.shape[2], lat.shape[3])
    p3 = net.conv(total3)
    p6 = net.conv(p5)
    p7 = net.conv(p6)
    confs, boxes = [], []
    n = x.shape[0]
    for level in (p3, p4, p5, p6, p7):
        c = level
        for _ in range(5):
            c = net.conv(c)
        r = level
        for _ in range(5):
            r = net.conv(r)
        confs.append(c.permute(0, 2, 3, 1).reshape(n, -1, C))
        boxes.append(r.permute(0, 2, 3, 1).reshape(n, -1, 4))
    assert net.i == len(specs)
    return torch.cat(confs, 1), torch.cat(boxes, 1)


def _num_classes(p, block_list):
    """the first class-subnet output conv (9 * classes channels) sits 4 layers behind the pyramid"""
    idx = 1 + 4 * sum(block_list) + 7 + 4
    return p[f'l{idx}.w'].shape[0] // 9


def loss_fn(p, images_nhwc, ground_truth, weight_decay=1e-4, stats_out=None, alpha=0.25, gamma=2.0, relu_masks=None):
    pconf, pbbox = forward(p, images_nhwc, True, stats_out, relu_masks=relu_masks)
    h, w = images_nhwc.shape[1], images_nhwc.shape[2]
    anc = RR.anchors([h, w, 3], RR.pyramid_shapes(h, w))
    data = RR.batch_loss(pbbox[..., :2], pbbox[..., 2:], pconf, anc, ground_truth, alpha, gamma)
    l2 = sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))
    return data + weight_decay * l2, data


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=1e-4, relu_masks=None):
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    total, data = loss_fn(p, images_nhwc, ground_truth, weight_decay, stats, relu_masks=relu_masks)
    total.backward()
    grads = {}
    with torch.no_grad():
        for k in names:
            grads[k] = p[k].grad.clone()
            mom[k].mul_(0.9).add_(p[k].grad)
            p[k].sub_(lr * mom[k])
            p[k].requires_grad_(False)
            p[k].grad = None
        for name, (mean, var_unbiased) in stats.items():
            p[name + '.mmean'].mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * mean)
            p[name + '.mvar'].mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * var_unbiased)
    return float(total.detach()), float(data.detach()), grads
