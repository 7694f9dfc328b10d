"""CPU fp32 restatement of the reference's whole YOLOv3 model (TEST INFRASTRUCTURE ONLY): network, training step, inference.

Follows /root/reference/YOLOv3.py:
  * input ................................ images - mean (:61-76; test mode feeds the tensor AFTER the subtraction, the same
                                           re-binding quirk as SSD300: fed pixels bypass the mean, `subtract_mean=False`)
  * every conv = tf.layers.conv2d(same, bias) + batch_normalization (+ leaky_relu 0.1) .......... _conv_layer :493-506
  * DarkNet-53 ........................... _feature_extractor :389-396, _darknet_block :484-491
                                           (3x3/s2 conv, then `blocks` x [1x1 f/2, 3x3 f, residual sum])
  * heads ................................ _yolo3_header :398-417: [lateral 1x1 WITHOUT activation on the previous level's conv5
                                           -> nearest resize -> concat(bottom, lateral)] -> 1x1, 3x3, 1x1, 3x3, 1x1, 3x3 ->
                                           1x1 `final_units` prediction conv -- which is ALSO followed by batch norm and
                                           leaky_relu (is_activation defaults to True): predictions are BN'd, leaky logits
  * head call order ...................... pyd1 = block5 (1024), pyd2 = block4 (256 !), pyd3 = block3 (128)   :83-88
                                           (the reference passes 256 / 128 filters for levels 2 / 3, half of darknet's 512 / 256)
  * loss ................................. .5 * mean_i(loss_i) + wd * sum_v l2_loss(v) over trainables (:311-315);
                                           per-image loss in oracle/yolov3_ref.py
  * optimizer ............................ MomentumOptimizer(lr, 0.9) (:312), BN moving statistics via UPDATE_OPS (:317)
Layers are named c0 .. c74 in creation (= forward) order, parameters '<layer>.w' [K,R,S,C], '.b', '.gamma', '.beta',
'.mmean', '.mvar'.  Pinned against the reference's own _feature_extractor / _yolo3_header run on oracle/tf_shim
(tests/golden/yolov3_net.npz, tests/golden/make_golden_yolov3_net.py).  Only tests/ and the smoke/bench checkers import this.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import yolov3_ref as YR
from .ssd300_ref import BN_EPS, BN_MOMENTUM, conv2d_same

MEAN_RGB = (123.68, 116.779, 103.979)
DARKNET_BLOCKS = ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4))       # :391-395
HEAD_FILTERS = (1024, 256, 128)                                            # :83-86


def layer_specs(num_classes=20, num_priors=3):
    """[(name, cin, cout, k, stride, act)] in creation order; act: 'leaky' or None (the lateral convs)."""
    specs = []

    def add(cin, cout, k, s, act='leaky'):
        specs.append((f'c{len(specs)}', cin, cout, k, s, act))
        return cout
    c = add(3, 32, 3, 1)
    outs = []
    for f, blocks in DARKNET_BLOCKS:
        c = add(c, f, 3, 2)
        for _ in range(blocks):
            add(c, f // 2, 1, 1)
            add(f // 2, f, 3, 1)
        outs.append(c)
    final = (num_classes + 5) * num_priors
    bottoms = [outs[4], outs[3], outs[2]]
    top = None
    for lvl, f in enumerate(HEAD_FILTERS):
        cin = bottoms[lvl]
        if top is not None:
            add(top, f, 1, 1, None)                  # lateral on the previous level's conv5
            cin += f
        add(cin, f // 2, 1, 1); add(f // 2, f, 3, 1); add(f, f // 2, 1, 1); add(f // 2, f, 3, 1)
        top = add(f, f // 2, 1, 1)
        add(f // 2, f, 3, 1)
        add(f, final, 1, 1)
    return specs


def init_params(seed=0, num_classes=20, num_priors=3):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, cin, cout, k, _, _ in layer_specs(num_classes, num_priors):
        p[name + '.w'] = torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k))
        p[name + '.b'] = torch.randn(cout, generator=g) * 0.01
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(cout, generator=g)
        p[name + '.mmean'] = torch.zeros(cout)
        p[name + '.mvar'] = torch.ones(cout)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(('.mmean', '.mvar'))]


class _Net:
    def __init__(self, p, training, stats_out, taps, leaky_masks=None):
        self.p, self.training, self.stats, self.taps, self.i, self.masks = p, training, stats_out, taps, 0, leaky_masks
        self.specs = layer_specs_cache(p)

    def conv(self, x):
        name, _, _, _, stride, act = self.specs[self.i]
        self.i += 1
        p = self.p
        z = conv2d_same(x, p[name + '.w'], p[name + '.b'], stride)
        if self.training:
            mean = z.mean(dim=(0, 2, 3))
            var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
            if self.stats is not None:
                n = z.shape[0] * z.shape[2] * z.shape[3]
                self.stats[name] = (mean.detach(), var.detach() * (n / max(n - 1, 1)))
        else:
            mean, var = p[name + '.mmean'], p[name + '.mvar']
        y = (z - mean[None, :, None, None]) * (torch.rsqrt(var + BN_EPS) * p[name + '.gamma'])[None, :, None, None] \
            + p[name + '.beta'][None, :, None, None]
        if act == 'leaky':
            if self.masks is not None:        # the linear region is dictated (see forward): gradients comparable to 1e-4
                y = torch.where(self.masks[name], y, 0.1 * y)
            else:
                y = F.leaky_relu(y, 0.1)
        if self.taps is not None:
            self.taps[name] = y
        return y


def layer_specs_cache(p):
    final = p['c74.w'].shape[0]
    return layer_specs(final // 3 - 5, 3)


def forward(p, images_nhwc, training, stats_out=None, taps=None, subtract_mean=True, leaky_masks=None):
    """-> [pred1, pred2, pred3] each [N, H_l, W_l, num_priors, C+5] (head 1 = coarsest), YOLOv3.py:82-95.
    leaky_masks {layer: bool NCHW}: which elements take the slope-1 branch.  A pre-activation within float round-off of 0
    may fall on different sides in two implementations; one such element moves every upstream gradient by ~1 / sqrt(elements)
    (about 1 %), so gradient parity is tested on the SAME linear region as the implementation under test."""
    x = images_nhwc.float()
    if subtract_mean:
        x = x - torch.tensor(MEAN_RGB).view(1, 1, 1, 3)
    x = x.permute(0, 3, 1, 2)
    net = _Net(p, training, stats_out, taps, leaky_masks)
    x = net.conv(x)
    outs = []
    for f, blocks in DARKNET_BLOCKS:
        x = net.conv(x)
        for _ in range(blocks):
            x = x + net.conv(net.conv(x))
        outs.append(x)
    preds, top = [], None
    for lvl, bottom in enumerate((outs[4], outs[3], outs[2])):
        x = bottom
        if top is not None:
            lat = net.conv(top)
            lat = F.interpolate(lat, size=bottom.shape[2:], mode='nearest')      # exact 2x: identical to TF's nearest resize
            x = torch.cat([bottom, lat], 1)
        x = net.conv(net.conv(net.conv(net.conv(x))))
        top = net.conv(x)
        pred = net.conv(net.conv(top))
        n, c, h, w = pred.shape
        preds.append(pred.permute(0, 2, 3, 1).reshape(n, h, w, 3, c // 3))
    return preds


def loss_fn(p, images_nhwc, ground_truth, weight_decay=5e-4, stats_out=None, scales=(1., 1., 5., 1.), leaky_masks=None):
    """(total, data): .5 * mean_i loss_i + wd * sum l2_loss(trainables)   (YOLOv3.py:311-315)"""
    preds = forward(p, images_nhwc, True, stats_out, leaky_masks=leaky_masks)
    C = preds[0].shape[-1] - 5
    data = YR.batch_loss(preds, ground_truth, num_classes=C, coord_scale=scales[0], noobj_scale=scales[1], obj_scale=scales[2],
                         class_scale=scales[3])
    l2 = sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))
    return .5 * data + weight_decay * l2, data


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=5e-4, leaky_masks=None):
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    total, data = loss_fn(p, images_nhwc, ground_truth, weight_decay, stats, leaky_masks=leaky_masks)
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


def test_one_image(p, images_nhwc, score_thr=0.5, max_boxes=10, iou_thr=0.5, subtract_mean=False):
    """YOLOv3.py:320-368 on image 0 -> [scores, bbox, class_id]"""
    from . import detect_common as DC
    with torch.no_grad():
        preds = forward(p, images_nhwc, False, subtract_mean=subtract_mean)
    C = preds[0].shape[-1] - 5
    conf, box = YR.decode_candidates([q[0] for q in preds], num_classes=C)
    return DC.per_class_nms(conf, box, C, score_thr, max_boxes, iou_thr)
