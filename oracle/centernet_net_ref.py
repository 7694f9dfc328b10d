"""CPU fp32 restatement of the reference's whole CenterNet model (TEST INFRASTRUCTURE ONLY): DLA backbone, up-sampling tree,
centre detector, training step.

Follows /root/reference/CenterNet.py:
  * input ................................ (images / 255 - mean) / std                                                  :51-65
  * every layer is conv(bias) -> tf.layers.batch_normalization -> ReLU (none on the three outputs)                       :325-340
  * stem ................................. 7x7(16), 3x3(16), 3x3 / s2 (32)                                               :74-91
  * _basic_block(x, f) ................... 3x3(f), 3x3(f) + shortcut; the shortcut is tf.cond(channels == f, x, 1x1(f)(x)):
                                           TensorFlow builds BOTH branches, so the 1x1 conv and its batch norm exist as variables even
                                           where the input already has f channels ("ghost" layers: trainable, inside the L2 term, moved
                                           only by weight decay; their moving statistics never update)                   :378-389
  * _dla_generator(x, f, levels) ......... levels 1: b1 = block(x), b2 = block(b1), 3x3(f)(b1 + b2);  levels 2: the same with
                                           b1 = dla(x, f, 1), b2 = dla(b1, f, 1)                                          :391-402
  * stages ............................... s3 = pool(dla(stem, 64, 1)); s4 = pool(dla(s3, 128, 2)) + avgpool(1x1(128)(s3));
                                           s5 likewise (256, levels 2); s6 (512, levels 1); pool = 2x2 / s2 max            :92-110
  * up-sampling .......................... 1x1(256) on s6 / s5 / s4, 4x4 / s2 transposed convs, 3x3 on the sums           :111-126
  * centre detector ...................... 3x3 -> classes, 3x3 -> 2 (offset), 3x3 -> 2 (size), batch norm, no activation  :131-134
  * loss / optimizer ..................... mean_i loss_i + wd * l2(all trainables, ghosts included), tf.train.AdamOptimizer(lr)  :144-157
Layers c0 .. c65 in creation order = TensorFlow's variable order (tests/golden/centernet_variables.json); layer k owns '.w' [K,R,S,C]
(transposed convs: the kernel of the equivalent forward conv from the OUTPUT to the INPUT, w[ci][r][s][co] = tf_kernel[r][s][co][ci]),
'.b', '.gamma', '.beta', '.mmean', '.mvar'.
Pinned against the reference's own class run on oracle/tf_shim: tests/golden/centernet_train.npz (make_golden_centernet_net.py).
Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import centernet_ref as CR
from .ssd300_ref import BN_EPS, BN_MOMENTUM, conv2d_same, maxpool_same

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8


def layer_specs(num_classes=20):
    """[(name, kind, cin, cout, k, stride, relu, ghost)] in creation order; kind 'conv' | 'dconv'"""
    specs = []

    def add(kind, cin, cout, k, s, relu=True, ghost=False):
        specs.append((f'c{len(specs)}', kind, cin, cout, k, s, relu, ghost))
        return cout

    def block(cin, f):
        add('conv', cin, f, 3, 1); add('conv', f, f, 3, 1)
        add('conv', cin, f, 1, 1, ghost=(cin == f))           # the shortcut conv: unused (but created) when the channels match
        return f

    def dla(cin, f, levels):
        if levels == 1:
            block(cin, f); block(f, f)
        else:
            dla(cin, f, levels - 1); dla(f, f, levels - 1)
        add('conv', f, f, 3, 1)
        return f
    add('conv', 3, 16, 7, 1); add('conv', 16, 16, 3, 1); add('conv', 16, 32, 3, 2)
    dla(32, 64, 1)
    dla(64, 128, 2); add('conv', 64, 128, 1, 1)
    dla(128, 256, 2); add('conv', 128, 256, 1, 1)
    dla(256, 512, 1); add('conv', 256, 512, 1, 1)
    add('conv', 512, 256, 1, 1)
    for _ in range(3):
        add('dconv', 256, 256, 4, 2)
    add('conv', 256, 256, 1, 1); add('conv', 256, 256, 3, 1)
    add('dconv', 256, 256, 4, 2); add('dconv', 256, 256, 4, 2)
    add('conv', 128, 256, 1, 1); add('conv', 256, 256, 3, 1)
    add('dconv', 256, 256, 4, 2)
    add('conv', 256, 256, 3, 1); add('conv', 256, 256, 1, 1)
    add('conv', 256, num_classes, 3, 1, relu=False); add('conv', 256, 2, 3, 1, relu=False); add('conv', 256, 2, 3, 1, relu=False)
    return specs


def init_params(seed=0, num_classes=20):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, kind, cin, cout, k, _, _, _ in layer_specs(num_classes):
        kin, kout = (cin, cout) if kind == 'conv' else (cout, cin)      # dconv: stored as the forward conv output -> input
        p[name + '.w'] = torch.randn(kout, k, k, kin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = torch.zeros(cout)                               # zero, as TensorFlow initialises it: in front of a batch norm it never moves
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(cout, generator=g)
        p[name + '.mmean'] = torch.zeros(cout)
        p[name + '.mvar'] = torch.ones(cout)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(('.mmean', '.mvar'))]


def dconv_same(x, w_fwd, b, stride):
    """tf.layers.conv2d_transpose(padding='same') = the gradient of the SAME forward conv (w_fwd [K = cin_of_dconv][R][S][C = cout_of_dconv])
    with respect to its input; x NCHW [N, K, H, W] -> [N, C, H * stride, W * stride]"""
    from .ssd300_ref import same_pad
    n, _, h, w = x.shape
    k = w_fwd.shape[1]
    oh, ow = h * stride, w * stride
    _, pt, pb = same_pad(oh, k, stride)
    _, pl, pr = same_pad(ow, k, stride)
    wt = w_fwd.permute(0, 3, 1, 2).contiguous()                 # [K, C, R, S]
    full = torch.nn.grad.conv2d_input((n, w_fwd.shape[3], oh + pt + pb, ow + pl + pr), wt, x, stride=stride)
    return full[:, :, pt: pt + oh, pl: pl + ow] + b[None, :, None, None]


class _Net:
    def __init__(self, p, specs, training, stats_out, relu_masks, taps):
        self.p, self.specs, self.training, self.stats, self.masks, self.taps, self.i = p, specs, training, stats_out, relu_masks, taps, 0

    def layer(self, x):
        name, kind, cin, cout, k, stride, relu, ghost = self.specs[self.i]
        self.i += 1
        assert not ghost and x.shape[1] == cin, (name, x.shape, cin)
        p = self.p
        z = conv2d_same(x, p[name + '.w'], p[name + '.b'], stride) if kind == 'conv' else dconv_same(x, p[name + '.w'], p[name + '.b'], stride)
        if self.training:
            mean = z.mean(dim=(0, 2, 3))
            var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
            if self.stats is not None:
                n = z.shape[0] * z.shape[2] * z.shape[3]
                self.stats[name] = (mean.detach(), var.detach() * (n / max(n - 1, 1)))
        else:
            mean, var = p[name + '.mmean'], p[name + '.mvar']
        y = (z - mean[None, :, None, None]) * (torch.rsqrt(var + BN_EPS) * p[name + '.gamma'])[None, :, None, None] + p[name + '.beta'][None, :, None, None]
        if relu:
            y = torch.where(self.masks[name], y, torch.zeros_like(y)) if self.masks is not None else F.relu(y)
        if self.taps is not None:
            self.taps[name] = y
        return y

    def ghost(self):
        assert self.specs[self.i][7]
        self.i += 1

    def block(self, x, f):
        c = self.layer(self.layer(x))
        if x.shape[1] == f:
            self.ghost()
            return c + x
        return c + self.layer(x)

    def dla(self, x, f, levels):
        if levels == 1:
            b1 = self.block(x, f); b2 = self.block(b1, f)
        else:
            b1 = self.dla(x, f, levels - 1); b2 = self.dla(b1, f, levels - 1)
        return self.layer(b1 + b2)


def forward(p, images_nhwc, training, stats_out=None, relu_masks=None, taps=None, normalize=True):
    """-> keypoints [N,H/4,W/4,classes] logits, offset [N,H/4,W/4,2], size [N,H/4,W/4,2]"""
    nc = p['c63.w'].shape[0]
    specs = layer_specs(nc)
    x = images_nhwc.float()
    if normalize:
        x = (x / 255. - torch.tensor(MEAN).view(1, 1, 1, 3)) / torch.tensor(STD).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, specs, training, stats_out, relu_masks, taps)
    x = net.layer(net.layer(net.layer(x)))
    s3 = maxpool_same(net.dla(x, 64, 1), 2, 2)
    stages = [s3]
    for f, levels in ((128, 2), (256, 2), (512, 1)):
        prev = stages[-1]
        d = net.dla(prev, f, levels)
        res = F.avg_pool2d(net.layer(prev), 2, 2)
        stages.append(maxpool_same(d, 2, 2) + res)
    s3, s4, s5, s6 = stages
    u6 = net.layer(s6)
    u6_5 = net.layer(u6); u6_4 = net.layer(u6_5); u6_3 = net.layer(u6_4)
    u5 = net.layer(s5)
    u5_4 = net.layer(net.layer(u5 + u6_5))
    u5_3 = net.layer(u5_4)
    u4 = net.layer(s4)
    u4_3 = net.layer(net.layer(u4 + u5_4 + u6_4))
    feat = net.layer(net.layer(u6_3 + u5_3 + u4_3))
    kp, off, size = net.layer(feat), net.layer(feat), net.layer(feat)
    assert net.i == len(specs)
    return kp.permute(0, 2, 3, 1), off.permute(0, 2, 3, 1), size.permute(0, 2, 3, 1)


def loss_fn(p, images_nhwc, ground_truth, weight_decay=1e-4, stats_out=None, relu_masks=None):
    kp, off, size = forward(p, images_nhwc, True, stats_out, relu_masks)
    data = CR.batch_loss(kp, off, size, ground_truth)
    l2 = sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))
    return data + weight_decay * l2, data


def train_step(p, state, images_nhwc, ground_truth, lr, weight_decay=1e-4, relu_masks=None):
    """one AdamOptimizer step in place; `state` = {'t': 0, 'm': {}, 'v': {}} -> (total loss, data loss, grads incl. the L2 term)"""
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    total, data = loss_fn(p, images_nhwc, ground_truth, weight_decay, stats, relu_masks)
    total.backward()
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
    grads = {}
    with torch.no_grad():
        for k in names:
            g = p[k].grad.clone()
            grads[k] = g
            m = state.setdefault('m', {}).setdefault(k, torch.zeros_like(g))
            v = state.setdefault('v', {}).setdefault(k, torch.zeros_like(g))
            m.mul_(ADAM_B1).add_(g * (1.0 - ADAM_B1))
            v.mul_(ADAM_B2).add_(g * g * (1.0 - ADAM_B2))
            p[k].sub_(lr_t * m / (torch.sqrt(v) + ADAM_EPS))
            p[k].requires_grad_(False)
            p[k].grad = None
        for name, (mean, unb) in stats.items():
            p[name + '.mmean'].mul_(BN_MOMENTUM).add_(mean * (1 - BN_MOMENTUM))
            p[name + '.mvar'].mul_(BN_MOMENTUM).add_(unb * (1 - BN_MOMENTUM))
    return float(total.detach()), float(data.detach()), grads


def test_one_image(p, images_nhwc, score_threshold, top_k):
    with torch.no_grad():
        kp, off, size = forward(p, images_nhwc, False)
    return CR.decode(kp[0], off[0], size[0], score_threshold, top_k)
