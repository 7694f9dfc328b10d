"""CPU fp32 restatement of the reference's whole RefineDet320 model (TEST INFRASTRUCTURE ONLY): network and training step.

Follows /root/reference/RefineDet.py:
  * input ................................ images - mean (:51-68; test mode feeds the tensor after the subtraction)
  * VGG-16 trunk conv1_1 .. conv5_3 ...... tf.nn.conv2d + bias + ReLU, no batch norm, 2x2 / s2 SAME pools, pool5 3x3 / s1 (:232-365)
  * extras conv6 (3x3, dilation 2, 1024), conv7 (1x1, 1024), conv8_1 (1x1, 256), conv8_2 (3x3 / s2, 512), conv9_1 (1x1, 256), conv9_2 (3x3 / s2, 512),
    conv10_1 (1x1, 256), conv10_2 (3x3, stride 1, 256): tf.layers.conv2d(bias) -> batch norm -> ReLU (:366-373, :631-646)
  * features ............................. conv4_3, conv5_3 (both L2-normalised over channels and scaled by ONE learnable scalar each, initial 10 and 8),
                                           conv8_2, conv10_2 (:74-95, :379-385)
  * ARM (per level) ...................... 4 x [3x3(256) + BN + ReLU], then 3x3 -> 4 * 3 box outputs and 3x3 -> 2 * 3 class outputs, BN, no activation (:387-395)
  * TCB (top-down: tcb4, tcb3, tcb2, tcb1)  3x3(256) + BN + ReLU, 3x3(256) + BN; with a higher level: + [4x4 / s2 transposed conv(256) + BN] of that level's
                                           block, ReLU (:397-405)
  * ODM (per level, on the TCB outputs) .. as the ARM with num_classes * 3 class outputs (:407-415)
  * loss / optimizer ..................... sum_i loss_i / batch (oracle/refinedet_ref.py) + wd * l2(all trainables), MomentumOptimizer(0.9) (:160-187)
Parameters by layer name ('.w' [K,R,S,C]; transposed convs as the filter of the stride-2 conv they are the gradient of, [cin][4][4][cout]; '.b';
batch-normalised layers also '.gamma', '.beta', '.mmean', '.mvar'), plus 'feat1_l2_norm', 'feat2_l2_norm'.
Pinned against the reference's own class run on oracle/tf_shim: tests/golden/refinedet_train.npz (tests/golden/make_golden_refinedet_net.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import refinedet_ref as FR
from .centernet_net_ref import dconv_same
from .ssd300_ref import BN_EPS, BN_MOMENTUM, MEAN_RGB, VGG_LAYERS, conv2d_same, maxpool_same

EXTRAS = [("conv6", 512, 1024, 3, 1, 2), ("conv7", 1024, 1024, 1, 1, 1), ("conv8_1", 1024, 256, 1, 1, 1), ("conv8_2", 256, 512, 3, 2, 1),
          ("conv9_1", 512, 256, 1, 1, 1), ("conv9_2", 256, 512, 3, 2, 1), ("conv10_1", 512, 256, 1, 1, 1), ("conv10_2", 256, 256, 3, 1, 1)]
FEAT_CH = [512, 512, 512, 256]
NA = 3


def layer_specs(num_classes=21):
    """[(name, kind, cin, cout, k, stride, dil, relu)] in TensorFlow's creation order; kind 'vgg' (bias + ReLU, no BN) | 'conv' | 'dconv' (both + BN)"""
    s = []
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            s.append((l[0], 'vgg', l[1], l[2], 3, 1, 1, True))
    for (n, ci, co, k, st, d) in EXTRAS:
        s.append((n, 'conv', ci, co, k, st, d, True))

    def head(prefix, cin, ncls):
        c = cin
        for j in range(1, 5):
            s.append((f'{prefix}.c{j}', 'conv', c, 256, 3, 1, 1, True)); c = 256
        s.append((f'{prefix}.loc', 'conv', 256, 4 * NA, 3, 1, 1, False))
        s.append((f'{prefix}.conf', 'conv', 256, ncls * NA, 3, 1, 1, False))
    for l in range(4):
        head(f'arm{l + 1}', FEAT_CH[l], 2)
    for l in (4, 3, 2, 1):
        s.append((f'tcb{l}.c1', 'conv', FEAT_CH[l - 1], 256, 3, 1, 1, True))
        s.append((f'tcb{l}.c2', 'conv', 256, 256, 3, 1, 1, l == 4))          # tcb4: relu(bn(conv)); the others get their ReLU after the sum
        if l < 4:
            s.append((f'tcb{l}.d', 'dconv', 256, 256, 4, 2, 1, False))
    for l in range(4):
        head(f'odm{l + 1}', 256, num_classes)
    return s


def init_params(seed=0, num_classes=21):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, kind, cin, cout, k, _, _, _ in layer_specs(num_classes):
        kout, kin = (cout, cin) if kind != 'dconv' else (cin, cout)
        p[name + '.w'] = torch.randn(kout, k, k, kin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = 0.05 * torch.randn(cout, generator=g) if kind == 'vgg' else torch.zeros(cout)
        if kind != 'vgg':
            p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(cout, generator=g)
            p[name + '.beta'] = 0.1 * torch.randn(cout, generator=g)
            p[name + '.mmean'] = torch.zeros(cout)
            p[name + '.mvar'] = torch.ones(cout)
    p['feat1_l2_norm'] = torch.full((1,), 10.0)
    p['feat2_l2_norm'] = torch.full((1,), 8.0)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(('.mmean', '.mvar'))]


class _Net:
    def __init__(self, p, training, stats, taps):
        self.p, self.training, self.stats, self.taps = p, training, stats, taps
        self.spec = {s[0]: s for s in layer_specs(p['odm1.conf.w'].shape[0] // NA)}

    def __call__(self, name, x):
        _, kind, cin, cout, k, stride, dil, relu = self.spec[name]
        p = self.p
        if kind == 'vgg':
            y = F.relu(conv2d_same(x, p[name + '.w'], p[name + '.b']))
        else:
            z = conv2d_same(x, p[name + '.w'], p[name + '.b'], stride, dil) if kind == 'conv' else dconv_same(x, p[name + '.w'], p[name + '.b'], stride)
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
                y = F.relu(y)
        if self.taps is not None:
            self.taps[name] = y
        return y


def _l2(x, gamma):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12)) * gamma


def forward(p, images_nhwc, training, stats_out=None, taps=None, subtract_mean=True):
    """-> arm_loc [N,A,4], arm_conf [N,A,2], odm_loc [N,A,4], odm_conf [N,A,classes] (level-major, then y, x, anchor)"""
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, training, stats_out, taps)
    feats = {}
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            x = net(l[0], x)
            feats[l[0]] = x
        else:
            x = maxpool_same(x, 3, 1) if l == 'pool5' else maxpool_same(x, 2, 2)
    for e in EXTRAS:
        x = net(e[0], x)
        feats[e[0]] = x
    f = [_l2(feats['conv4_3'], p['feat1_l2_norm']), _l2(feats['conv5_3'], p['feat2_l2_norm']), feats['conv8_2'], feats['conv10_2']]
    if taps is not None:
        taps['feat1'], taps['feat2'] = f[0], f[1]
    n = x.shape[0]

    def head(prefix, x, ncls):
        c = x
        for j in range(1, 5):
            c = net(f'{prefix}.c{j}', c)
        loc = net(f'{prefix}.loc', c).permute(0, 2, 3, 1).reshape(n, -1, 4)
        conf = net(f'{prefix}.conf', c).permute(0, 2, 3, 1).reshape(n, -1, ncls)
        return loc, conf
    arm = [head(f'arm{l + 1}', f[l], 2) for l in range(4)]
    tcb = {}
    for l in (4, 3, 2, 1):
        c2 = net(f'tcb{l}.c2', net(f'tcb{l}.c1', f[l - 1]))
        tcb[l] = c2 if l == 4 else F.relu(c2 + net(f'tcb{l}.d', tcb[l + 1]))
        if taps is not None:
            taps[f'tcb{l}'] = tcb[l]
    ncls = p['odm1.conf.w'].shape[0] // NA
    odm = [head(f'odm{l + 1}', tcb[l + 1], ncls) for l in range(4)]
    return (torch.cat([a[0] for a in arm], 1), torch.cat([a[1] for a in arm], 1), torch.cat([o[0] for o in odm], 1), torch.cat([o[1] for o in odm], 1))


def loss_fn(p, images_nhwc, ground_truth, weight_decay=1e-4, stats_out=None, anchors=None):
    anchors = anchors or FR.anchors(images_nhwc.shape[1])
    al, ac, ol, oc = forward(p, images_nhwc, True, stats_out)
    data = FR.batch_loss(al, ac, ol, oc, anchors, ground_truth, oc.shape[-1])
    l2 = sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))
    return data + weight_decay * l2, data


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=1e-4):
    """one MomentumOptimizer(0.9) step in place -> (total loss, data loss, gradients incl. the L2 term)"""
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    total, data = loss_fn(p, images_nhwc, ground_truth, weight_decay, stats)
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


def test_one_image(p, images_nhwc, score_thr, max_boxes, iou_thr):
    with torch.no_grad():
        al, ac, ol, oc = forward(p, images_nhwc, False, subtract_mean=False)         # the reference's test-mode feed bypasses the mean subtraction
    return FR.detect(al[0], ac[0], ol[0], oc[0], FR.anchors(images_nhwc.shape[1]), score_thr, max_boxes, iou_thr, oc.shape[-1])
