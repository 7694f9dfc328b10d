"""CPU fp32 restatement of the reference's whole FCOS model (TEST INFRASTRUCTURE ONLY): network and training step.

Follows /root/reference/FCOS.py:
  * input ................................ images - mean (:51-68)
  * every normalisation is tf.contrib.layers.group_norm(groups=8, epsilon 1e-6): per SAMPLE and group, moments over H, W and the
    group's channels; gamma / beta per channel (:438-446) -- no batch statistics, no moving averages
  * stem ................................. conv(7x7, stride 2, 16) + GN + ReLU, 3x3 / stride-2 max pool (:73-85)
  * pre-activation bottleneck units ...... GN -> ReLU -> conv(bias); 1x1 f, 3x3 f (stride), 1x1 4f + a 3x3 (stride) 4f shortcut conv on every
                                           unit (:504-513); blocks 3, 4, 6, 3 with f = 16, 32, 64, 128 (:29-31)
  * pyramid .............................. c3, c4, c5 = 1x1(256) on the last three stages; p5 = 3x3(c5); p4 / p3: ANOTHER 1x1 on c4 / c3 + bilinear
                                           resize of the level above (TF-1.x grid), the sum is handed down, 3x3 on the sum; p6, p7 = 3x3 / s2
                                           (:98-107, :366-382)
  * heads ................................ 4 x 3x3(256) -> 3x3(classes, pi bias) and 3x3(1, pi bias) centre-ness; 4 x 3x3(256) -> exp(3x3(4))
                                           (:350-364).  ONE set of 6 + 5 layers (convs AND group norms) serves all five levels: _detect_head enters
                                           variable_scope('classifier_head' / 'regress_head', reuse=tf.AUTO_REUSE) once per level, leaving a scope
                                           resets the default-name counters of its sub-scopes (TF 1.x variable_scope.py, close_variable_subscopes),
                                           so every level asks for conv2d, conv2d_1, ... / GroupNorm, GroupNorm_1, ... again and gets the first
                                           level's variables back
  * loss / optimizer ..................... mean_i loss_i + wd * l2(all trainables), Momentum 0.9 (:186-192); per-image loss: oracle/fcos_ref.py
Layers l0 .. l85 in creation order (l75 .. l85: the shared head layers; their ReLU masks / taps are keyed l<k>@<level>); layer k owns conv k and group norm k ('.w' [K,R,S,C], '.b', '.gamma', '.beta'): l0 is conv -> GN -> ReLU,
every other layer GN -> ReLU -> conv (its GN has the conv's INPUT channels).
Pinned against the reference's own class run on oracle/tf_shim: tests/golden/fcos_train.npz (tests/golden/make_golden_fcos_net.py).
Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import fcos_ref as FR
from .augment_ref import resize_bilinear_legacy
from .ssd300_ref import conv2d_same, maxpool_same

MEAN_RGB = (123.68, 116.779, 103.979)
BLOCKS = (3, 4, 6, 3)
FILTERS = (16, 32, 64, 128)
GROUPS, GN_EPS = 8, 1e-6
PI_BIAS = -math.log((1 - 0.01) / 0.01)


def layer_specs(num_classes=20):
    """[(name, cin, cout, k, stride, gn_channels, bias_init)] in creation order"""
    specs = []

    def add(cin, cout, k, s, bias_init=0.):
        specs.append((f'l{len(specs)}', cin, cout, k, s, cin if specs else cout, bias_init))
        return cout
    c = add(3, 16, 7, 2)
    stage_out = []
    for i, blocks in enumerate(BLOCKS):
        f = FILTERS[i]
        for j in range(blocks):
            s = 2 if (i > 0 and j == 0) else 1
            add(c, f, 1, 1); add(f, f, 3, s); add(f, 4 * f, 1, 1)
            add(c, 4 * f, 3, s)
            c = 4 * f
        stage_out.append(c)
    e3, e4, e5 = stage_out[-3:]
    add(e3, 256, 1, 1); add(e4, 256, 1, 1); add(e5, 256, 1, 1)       # c3, c4, c5
    add(256, 256, 3, 1)                                              # p5
    add(256, 256, 1, 1); add(256, 256, 3, 1)                         # p4: 1x1 on c4, 3x3 on the sum
    add(256, 256, 1, 1); add(256, 256, 3, 1)                         # p3
    add(256, 256, 3, 2); add(256, 256, 3, 2)                         # p6, p7
    for _ in range(4):                                               # the heads: one set of layers for all five levels
        add(256, 256, 3, 1)
    add(256, num_classes, 3, 1, PI_BIAS)
    add(256, 1, 3, 1, PI_BIAS)
    for _ in range(4):
        add(256, 256, 3, 1)
    add(256, 4, 3, 1)
    return specs


HEAD_LAYERS = 11


def init_params(seed=0, num_classes=20):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, cin, cout, k, _, gnc, bias_init in layer_specs(num_classes):
        p[name + '.w'] = torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = torch.full((cout,), float(bias_init)) + (0.01 * torch.randn(cout, generator=g) if bias_init == 0. else 0.)
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(gnc, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(gnc, generator=g)
    return p


def group_norm(x, gamma, beta):
    """NCHW; tf.contrib.layers.group_norm: per sample and group, biased variance, epsilon 1e-6"""
    n, c, h, w = x.shape
    xg = x.reshape(n, GROUPS, c // GROUPS, h, w)
    mean = xg.mean(dim=(2, 3, 4), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(2, 3, 4), keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + GN_EPS)).reshape(n, c, h, w)
    return y * gamma[None, :, None, None] + beta[None, :, None, None]


class _Net:
    def __init__(self, p, specs, relu_masks, taps):
        self.p, self.specs, self.masks, self.taps, self.i = p, specs, relu_masks, taps, 0

    def _gn_relu(self, name, x, key=None):
        key = key or name                                # instance key of a shared layer: l<k>@<level>
        y = group_norm(x, self.p[name + '.gamma'], self.p[name + '.beta'])
        y = torch.where(self.masks[key], y, torch.zeros_like(y)) if self.masks is not None else F.relu(y)
        if self.taps is not None:
            self.taps[key] = y
        return y

    def stem(self, x):
        name, _, _, _, stride, _, _ = self.specs[self.i]
        self.i += 1
        return self._gn_relu(name, conv2d_same(x, self.p[name + '.w'], self.p[name + '.b'], stride))

    def conv(self, x):
        name, _, _, _, stride, _, _ = self.specs[self.i]
        self.i += 1
        return conv2d_same(self._gn_relu(name, x), self.p[name + '.w'], self.p[name + '.b'], stride)

    def head(self, j, level, x):
        """shared head layer j (0 .. 10, counted from the first head spec) applied at pyramid level `level`"""
        name, _, _, _, stride, _, _ = self.specs[len(self.specs) - HEAD_LAYERS + j]
        return conv2d_same(self._gn_relu(name, x, f'{name}@{level}'), self.p[name + '.w'], self.p[name + '.b'], stride)


def _resize(x, h, w):
    return torch.stack([resize_bilinear_legacy(img.permute(1, 2, 0), h, w).permute(2, 0, 1) for img in x])


def forward(p, images_nhwc, subtract_mean=True, relu_masks=None, taps=None):
    """-> conf, reg, center: lists of 5 level tensors [N,H,W,classes] logits, [N,H,W,4] distances (after the exp), [N,H,W,1] logits"""
    specs = layer_specs(_num_classes(p))
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, specs, relu_masks, taps)
    x = maxpool_same(net.stem(x), 3, 2)
    feats = []
    for blocks in BLOCKS:
        for _ in range(blocks):
            branch = net.conv(net.conv(net.conv(x)))
            x = branch + net.conv(x)
        feats.append(x)
    c3, c4, c5 = net.conv(feats[-3]), net.conv(feats[-2]), net.conv(feats[-1])
    p5 = net.conv(c5)
    lat = net.conv(c4)
    total4 = lat + _resize(p5, lat.shape[2], lat.shape[3])
    p4 = net.conv(total4)
    lat = net.conv(c3)
    total3 = lat + _resize(total4, lat.shape[2], lat.shape[3])
    p3 = net.conv(total3)
    p6 = net.conv(p5)
    p7 = net.conv(p6)
    conf, reg, center = [], [], []
    assert net.i == len(specs) - HEAD_LAYERS
    for l, level in enumerate((p3, p4, p5, p6, p7)):
        c = level
        for j in range(4):
            c = net.head(j, l, c)
        conf.append(net.head(4, l, c).permute(0, 2, 3, 1))
        center.append(net.head(5, l, c).permute(0, 2, 3, 1))
        r = level
        for j in range(4):
            r = net.head(6 + j, l, r)
        reg.append(torch.exp(net.head(10, l, r)).permute(0, 2, 3, 1))
    return conf, reg, center


def _num_classes(p):
    return p[f'l{1 + 4 * sum(BLOCKS) + 10 + 4}.w'].shape[0]


def trainable_names(p):
    return list(p)


def loss_fn(p, images_nhwc, ground_truth, weight_decay=1e-4, relu_masks=None):
    conf, reg, center = forward(p, images_nhwc, relu_masks=relu_masks)
    data = FR.batch_loss(conf, reg, center, ground_truth)
    l2 = sum((v ** 2).sum() / 2 for v in p.values())
    return data + weight_decay * l2, data


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=1e-4, relu_masks=None):
    for k in p:
        p[k].requires_grad_(True)
        p[k].grad = None
    total, data = loss_fn(p, images_nhwc, ground_truth, weight_decay, relu_masks)
    total.backward()
    grads = {}
    with torch.no_grad():
        for k in p:
            grads[k] = p[k].grad.clone()
            mom[k].mul_(0.9).add_(p[k].grad)
            p[k].sub_(lr * mom[k])
            p[k].requires_grad_(False)
            p[k].grad = None
    return float(total.detach()), float(data.detach()), grads
