"""CPU fp32 restatement of the reference's whole PFPNetR model (TEST INFRASTRUCTURE ONLY): network and training step.

Follows /root/reference/PFPNetR.py (class PFPNetR; input_size a multiple of 64, testpfpnetr.py uses 320):
  * input ................................ images - mean (:55-72; test mode feeds the tensor after the subtraction)
  * VGG-16 trunk conv1_1 .. conv4_3 ...... tf.nn.conv2d + bias + ReLU, 2x2 / s2 SAME pools after conv1_2, conv2_2, conv3_3 (:231-313) -> fh1, stride 8
  * fh2, fh3, fh4 ........................ tf.image.resize_bilinear(fh1, half / quarter / eighth, align_corners=True) (:315-324)
  * the parallel feature pyramid ......... 85-channel (512 // 6) branches, every layer tf.layers.conv2d / conv2d_transpose (bias) + batch norm (:330-364):
       fl_k   = relu(bn(1x1(fh_k)))                                      k = 1..4
       fl_a_b = relu(bn(1x1( bn(4x4 / s2 transposed conv(fl_a or fl_a_(b+1))) + fl_b )))   up-path (a > b): fl2_1, fl3_2, fl3_1, fl4_3, fl4_2, fl4_1
       fl_a_b = bn(1x1(avg_pool2x2(fl_a or fl_a_(b-1))))                  down-path (a < b): fl1_2, fl1_3, fl1_4, fl2_3, fl2_4, fl3_4 (no activation)
  * features ............................. feat_k = concat over the channels of the four level-k tensors IN SOURCE ORDER (fh_k sits at position k; 512 + 3 * 85 =
                                           767 channels); feat1, feat2 L2-normalised and scaled by one learnable scalar each (10, 8) (:76-94, :366-396)
  * ARM / TCB / ODM, loss, optimizer ..... identical to RefineDet.py (same text): oracle/refinedet_net_ref.py, oracle/refinedet_ref.py (:403-431, :469-610)
Parameters by layer name as in oracle/refinedet_net_ref.py.
Pinned against the reference's own class run on oracle/tf_shim: tests/golden/pfpnet_train.npz (tests/golden/make_golden_pfpnet.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import refinedet_net_ref as NR
from . import refinedet_ref as FR
from .ssd300_ref import BN_MOMENTUM, MEAN_RGB, maxpool_same

VGG = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool1", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool2",
       ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool3",
       ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512)]
CB = 512 // 6                                   # 85 channels per pyramid branch (PFPNetR.py:330)
UP = [(2, 1), (3, 2), (3, 1), (4, 3), (4, 2), (4, 1)]        # creation order of the up-path blocks fl<a>_<b>
DOWN = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
FEAT_C = 512 + 3 * CB
NA = 3


def layer_specs(num_classes=21):
    """[(name, kind, cin, cout, k, stride, dil, relu)] in TensorFlow's creation order; kind 'vgg' (bias + ReLU, no BN) | 'conv' | 'dconv' (both + BN)"""
    s = []
    for l in VGG:
        if isinstance(l, tuple):
            s.append((l[0], 'vgg', l[1], l[2], 3, 1, 1, True))
    for k in range(1, 5):
        s.append((f'fl{k}', 'conv', 512, CB, 1, 1, 1, True))
    for a, b in UP:
        s.append((f'fl{a}_{b}d', 'dconv', CB, CB, 4, 2, 1, False))
        s.append((f'fl{a}_{b}c', 'conv', CB, CB, 1, 1, 1, True))
    for a, b in DOWN:
        s.append((f'fl{a}_{b}', 'conv', CB, CB, 1, 1, 1, False))

    def head(prefix, cin, ncls):
        c = cin
        for j in range(1, 5):
            s.append((f'{prefix}.c{j}', 'conv', c, 256, 3, 1, 1, True)); c = 256
        s.append((f'{prefix}.loc', 'conv', 256, 4 * NA, 3, 1, 1, False))
        s.append((f'{prefix}.conf', 'conv', 256, ncls * NA, 3, 1, 1, False))
    for l in range(4):
        head(f'arm{l + 1}', FEAT_C, 2)
    for l in (4, 3, 2, 1):
        s.append((f'tcb{l}.c1', 'conv', FEAT_C, 256, 3, 1, 1, True))
        s.append((f'tcb{l}.c2', 'conv', 256, 256, 3, 1, 1, l == 4))
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


trainable_names = NR.trainable_names


class _Net(NR._Net):
    def __init__(self, p, training, stats, taps):
        self.p, self.training, self.stats, self.taps = p, training, stats, taps
        self.spec = {s[0]: s for s in layer_specs(p['odm1.conf.w'].shape[0] // NA)}


def resize_bilinear_align_corners(x, oh, ow):
    """tf.image.resize_bilinear(align_corners=True) on NCHW: src = dst * (in - 1) / (out - 1), taps floor(src) and min(floor(src) + 1, in - 1)"""
    h, w = x.shape[2], x.shape[3]
    sy = torch.tensor((h - 1) / (oh - 1) if oh > 1 else h / oh, dtype=torch.float32)
    sx = torch.tensor((w - 1) / (ow - 1) if ow > 1 else w / ow, dtype=torch.float32)
    fy, fx = torch.arange(oh, dtype=torch.float32) * sy, torch.arange(ow, dtype=torch.float32) * sx
    y0, x0 = torch.floor(fy).long(), torch.floor(fx).long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    ly, lx = (fy - y0.float()).view(1, 1, oh, 1), (fx - x0.float()).view(1, 1, 1, ow)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def features(net, x, taps=None):
    """images (NCHW, mean subtracted) -> [feat1 .. feat4] before the L2 normalisation (PFPNetR.py:230-401)"""
    for l in VGG:
        x = net(l[0], x) if isinstance(l, tuple) else maxpool_same(x, 2, 2)
    fh = {1: x}
    h = x.shape[2]
    for k in (2, 3, 4):
        fh[k] = resize_bilinear_align_corners(x, h >> (k - 1), x.shape[3] >> (k - 1))
    fl = {(k, k): net(f'fl{k}', fh[k]) for k in range(1, 5)}
    for a, b in UP:                                            # fl<a>_<b> = relu(bn(1x1(bn(dconv(fl<a>_<b+1>)) + fl<b>)))
        src = fl[(a, b + 1)]
        fl[(a, b)] = net(f'fl{a}_{b}c', net(f'fl{a}_{b}d', src) + fl[(b, b)])
    for a, b in DOWN:                                          # fl<a>_<b> = bn(1x1(avg_pool(fl<a>_<b-1>)))
        fl[(a, b)] = net(f'fl{a}_{b}', F.avg_pool2d(fl[(a, b - 1)], 2, 2))
    feats = []
    for k in range(1, 5):
        feats.append(torch.cat([fh[k] if a == k else fl[(a, k)] for a in range(1, 5)], dim=1))
    if taps is not None:
        for k in range(1, 5):
            taps[f'fh{k}'] = fh[k]
            taps[f'cat{k}'] = feats[k - 1]
    return feats


def forward(p, images_nhwc, training, stats_out=None, taps=None, subtract_mean=True):
    """-> arm_loc [N,A,4], arm_conf [N,A,2], odm_loc [N,A,4], odm_conf [N,A,classes] (level-major, then y, x, anchor)"""
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, training, stats_out, taps)
    f = features(net, x, taps)
    f = [NR._l2(f[0], p['feat1_l2_norm']), NR._l2(f[1], p['feat2_l2_norm']), f[2], f[3]]
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


def anchors(input_size):
    """PFPNetR.py:447-467 on the four levels of stride 8, 16, 32, 64 with size 4 * stride: RefineDet's table"""
    return FR.anchors(input_size)


def loss_fn(p, images_nhwc, ground_truth, weight_decay=1e-4, stats_out=None, anc=None):
    anc = anc or anchors(images_nhwc.shape[1])
    al, ac, ol, oc = forward(p, images_nhwc, True, stats_out)
    data = FR.batch_loss(al, ac, ol, oc, anc, ground_truth, oc.shape[-1])
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
    return FR.detect(al[0], ac[0], ol[0], oc[0], anchors(images_nhwc.shape[1]), score_thr, max_boxes, iou_thr, oc.shape[-1])
