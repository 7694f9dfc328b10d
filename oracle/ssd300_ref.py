"""CPU fp32 restatement of the reference SSD300 graph (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/SSD300.py:
  * inputs / mean subtraction ............ SSD300.py:52-69
  * VGG-16 trunk + extra layers .......... SSD300.py:192-314, helpers :506-547
  * L2-norm + scalar scale ............... SSD300.py:74-83
  * heads ................................ SSD300.py:85-110, :316-321
  * priors ............................... SSD300.py:112-127, :323-343
  * per-image loss ....................... SSD300.py:345-456
  * batch loss + L2 + momentum ........... SSD300.py:129-155
  * inference decode + per-class NMS ..... SSD300.py:157-190
TF-1.13 kernel semantics (SAME padding, fused BN, NMSv3, argmax ties, ...) are
restated from memory -- "parity unpinned" at that layer.

Tensors are torch CPU float32.  Activations are handled NCHW internally (torch
conv) but every public input/output uses the reference's NHWC / [rows, C] views.
Conv weights are kept as [Cout, R, S, Cin] ("KRSC"); TF's HWIO is a permutation.
"""
from __future__ import annotations

import math
import os
import ctypes
import subprocess
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

INPUT_SIZE = 300
MEAN_RGB = (123.68, 116.779, 103.979)          # SSD300.py:55 (blue is 103.979 as shipped)
BN_EPS = 1e-3                                   # tf.layers.batch_normalization default
BN_MOMENTUM = 0.99

# (name, cin, cout, k, stride, dilation, has_bn, relu)   SSD300.py:193-313, :85-90
VGG_LAYERS = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool1",
    ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool2",
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool3",
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "pool4",
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), "pool5",
]
EXTRA_LAYERS = [  # name, cin, cout, k, stride, dilation     SSD300.py:304-313
    ("conv6", 512, 1024, 3, 1, 2), ("conv7", 1024, 1024, 1, 1, 1),
    ("conv8_1", 1024, 256, 1, 1, 1), ("conv8_2", 256, 512, 3, 2, 1),
    ("conv9_1", 512, 128, 1, 1, 1), ("conv9_2", 128, 256, 3, 2, 1),
    ("conv10_1", 256, 128, 1, 1, 1), ("conv10_2", 128, 256, 3, 1, 1),
    ("conv11_1", 256, 128, 1, 1, 1), ("conv11_2", 128, 256, 3, 2, 1),
]
FEATS = ["conv4_3", "conv7", "conv8_2", "conv9_2", "conv10_2", "conv11_2"]  # SSD300.py:314
FEAT_CH = [512, 1024, 512, 256, 256, 256]
ANCHORS_PER_CELL = [4, 6, 6, 6, 4, 4]                                       # SSD300.py:85-90
ASPECTS = [[2, 1 / 2], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3],
           [2, 1 / 2, 3, 1 / 3], [2, 1 / 2], [2, 1 / 2]]                     # SSD300.py:114-119


# ----------------------------------------------------------------------------
# TF SAME-padding arithmetic
# ----------------------------------------------------------------------------
def same_pad(in_size: int, k: int, stride: int, dil: int = 1):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x, w_krsc, b, stride=1, dil=1):
    """x NCHW, w [Cout,R,S,Cin]; TF SAME (extra pad bottom/right)."""
    k = w_krsc.shape[1]
    _, pt, pb = same_pad(x.shape[2], k, stride, dil)
    _, pl, pr = same_pad(x.shape[3], k, stride, dil)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_krsc.permute(0, 3, 1, 2).contiguous(), b, stride=stride, dilation=dil)


def maxpool_same(x, k, stride):
    """tf.layers.max_pooling2d SAME: padded cells never win (-inf)."""
    _, pt, pb = same_pad(x.shape[2], k, stride)
    _, pl, pr = same_pad(x.shape[3], k, stride)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, k, stride)


def batch_norm(x, p, name, training, stats_out=None):
    """tf.layers.batch_normalization(axis=C), fused semantics (App. B)."""
    g, bt = p[name + ".gamma"], p[name + ".beta"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))     # biased
        if stats_out is not None:
            n = x.shape[0] * x.shape[2] * x.shape[3]
            stats_out[name] = (mean.detach(), var.detach() * (n / max(n - 1, 1)))
    else:
        mean, var = p[name + ".mmean"], p[name + ".mvar"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + bt[None, :, None, None]


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
def conv_specs():
    """All 29 convs in creation order: (name, cin, cout, k, stride, dil, bn, relu)."""
    specs = []
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            specs.append((l[0], l[1], l[2], 3, 1, 1, False, True))
    for (n, ci, co, k, s, d) in EXTRA_LAYERS:
        specs.append((n, ci, co, k, s, d, True, True))
    for i, (ch, a) in enumerate(zip(FEAT_CH, ANCHORS_PER_CELL)):
        specs.append((f"pred{i + 1}", ch, a * 25, 3, 1, 1, True, False))
    return specs


def init_params(seed=0, num_classes=21):
    """Synthetic init: He-normal conv, zero bias, BN 1/0, l2 scale 20."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for (name, ci, co, k, s, d, bn, relu) in conv_specs():
        fan_in = ci * k * k
        p[name + ".w"] = torch.randn(co, k, k, ci, generator=g) * math.sqrt(2.0 / fan_in)
        p[name + ".b"] = torch.zeros(co)
        if bn:
            p[name + ".gamma"] = torch.ones(co)
            p[name + ".beta"] = torch.zeros(co)
            p[name + ".mmean"] = torch.zeros(co)
            p[name + ".mvar"] = torch.ones(co)
    p["l2norm.gamma"] = torch.full((1,), 20.0)
    return p


def calibrate_bn(p, images_nhwc, subtract_mean=True):
    """Set the BN moving statistics to the batch statistics of `images` (gives inference-mode
    activations a sane scale for randomly initialised weights)."""
    stats = {}
    with torch.no_grad():
        forward(p, images_nhwc, True, stats, subtract_mean=subtract_mean)
    for name, (mean, var_unbiased) in stats.items():
        p[name + ".mmean"] = mean.clone()
        p[name + ".mvar"] = var_unbiased.clone()
    return p


def trainable_names(p):
    return [k for k in p if not (k.endswith(".mmean") or k.endswith(".mvar"))]


# ----------------------------------------------------------------------------
# network forward
# ----------------------------------------------------------------------------
def preprocess(images_nhwc, subtract_mean=True):
    """SSD300.py:52-63: subtract the RGB mean; returns NCHW.

    Reference quirk (reproduced): in TEST mode `self.images` is re-bound to `placeholder - mean`
    (SSD300.py:65-66) and test_one_image feeds THAT tensor (SSD300.py:487) -- a TF feed overrides
    the value of the fed tensor, so the pixels handed to test_one_image bypass the subtraction.
    Training batches (iterator output, SSD300.py:61-63) are mean-subtracted."""
    if subtract_mean:
        mean = torch.tensor(MEAN_RGB, dtype=torch.float32).view(1, 1, 1, 3)
        images_nhwc = images_nhwc - mean
    return images_nhwc.permute(0, 3, 1, 2).contiguous()


def forward(p, images_nhwc, training, stats_out=None, taps=None, subtract_mean=True):
    """Returns pred [N, 8828, 25] (level-major, then y, x, anchor; SSD300.py:316-321).

    `taps`, if a dict, receives intermediate activations as NHWC tensors."""
    x = preprocess(images_nhwc, subtract_mean)
    feats = {}
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            name = l[0]
            x = F.relu(conv2d_same(x, p[name + ".w"], p[name + ".b"]))       # SSD300.py:514-521
            if name in FEATS:
                feats[name] = x
        else:
            if l == "pool5":
                x = maxpool_same(x, 3, 1)                                    # SSD300.py:303
            else:
                x = maxpool_same(x, 2, 2)
        if taps is not None:
            taps[l[0] if isinstance(l, tuple) else l] = x.permute(0, 2, 3, 1)
            if taps.get("_retain") and x.requires_grad:
                x.retain_grad(); taps[(l[0] if isinstance(l, tuple) else l) + ".raw"] = x
    for (name, ci, co, k, s, d) in EXTRA_LAYERS:                             # SSD300.py:523-537
        x = conv2d_same(x, p[name + ".w"], p[name + ".b"], s, d)
        if taps is not None:
            taps[name + ".z"] = x.permute(0, 2, 3, 1)
            if taps.get("_retain") and x.requires_grad:
                x.retain_grad(); taps[name + ".z.raw"] = x
        x = F.relu(batch_norm(x, p, name, training, stats_out))
        if name in FEATS:
            feats[name] = x
        if taps is not None:
            taps[name] = x.permute(0, 2, 3, 1)
            if taps.get("_retain") and x.requires_grad:
                x.retain_grad(); taps[name + ".raw"] = x
    # L2 normalise conv4_3 across channels, one learnable scalar (SSD300.py:74-83)
    f1 = feats["conv4_3"]
    ss = (f1 * f1).sum(dim=1, keepdim=True)
    f1 = f1 * torch.rsqrt(torch.clamp(ss, min=1e-12)) * p["l2norm.gamma"]
    if taps is not None:
        taps["feat1"] = f1.permute(0, 2, 3, 1)
    srcs = [f1] + [feats[n] for n in FEATS[1:]]
    preds = []
    for i, f in enumerate(srcs):
        name = f"pred{i + 1}"
        z = conv2d_same(f, p[name + ".w"], p[name + ".b"])
        z = batch_norm(z, p, name, training, stats_out)                      # BN, no activation
        z = z.permute(0, 2, 3, 1)                                            # NHWC
        preds.append(z.reshape(z.shape[0], -1, 25))                          # SSD300.py:316-317
    return torch.cat(preds, dim=1)


# ----------------------------------------------------------------------------
# priors (SSD300.py:112-127, 323-343)
# ----------------------------------------------------------------------------
def feature_sizes():
    """side of the feature maps FEATS: conv4_3 after three 2x2 / s2 SAME pools, conv7 after the fourth, then the strides of the extra layers"""
    s = INPUT_SIZE
    for _ in range(3):            # pool1..3 (2x2 s2 SAME)
        s = -(-s // 2)
    at = {"conv4_3": s}           # 38
    s = -(-s // 2)                # pool4; pool5 is 3x3 / s1
    for (n, ci, co, k, st, d) in EXTRA_LAYERS:
        s = -(-s // st)           # conv10_2 has stride 1 (SSD300.py:311): 5 again
        at[n] = s
    return [at[n] for n in FEATS]


def prior_scales():
    """SSD300.py:112-113 (python doubles)"""
    s = [(0.2 + (0.9 - 0.2) / 5 * (i - 1)) * INPUT_SIZE for i in range(1, 8)]
    return [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 6)]


def priors():
    """Returns y1x1, y2x2, yx, hw each [8828, 2] float32, following the exact
    op order of _get_abbox so results are bit-identical to float32 TF math."""
    f32 = np.float32
    s = prior_scales()
    outs = [[], [], [], []]
    for lvl, (f, size, ar) in enumerate(zip(feature_sizes(), s, ASPECTS)):
        ty = (np.arange(0., f, dtype=f32).reshape(-1, 1, 1, 1) + f32(0.5))
        tx = (np.arange(0., f, dtype=f32).reshape(1, -1, 1, 1) + f32(0.5))
        ty = np.tile(ty, [1, f, 1, 1]) * f32(INPUT_SIZE) / f32(f)
        tx = np.tile(tx, [f, 1, 1, 1]) * f32(INPUT_SIZE) / f32(f)
        tyx = np.concatenate([ty, tx], -1)
        tyx = np.tile(tyx, [1, 1, len(ar) + 2, 1])
        pr = [[size[0], size[0]], [size[1], size[1]]]
        for a in ar:
            pr.append([size[0] * (a ** 0.5), size[0] / (a ** 0.5)])
        pr = np.asarray(pr, dtype=np.float64).astype(f32).reshape(1, 1, -1, 2)
        y1x1 = (tyx - pr / f32(2.)).reshape(-1, 2)
        y2x2 = (tyx + pr / f32(2.)).reshape(-1, 2)
        yx = y1x1 / f32(2.) + y2x2 / f32(2.)
        hw = y2x2 - y1x1
        for o, v in zip(outs, (y1x1, y2x2, yx, hw)):
            o.append(v.astype(f32))
    return tuple(torch.from_numpy(np.concatenate(o, 0)) for o in outs)


# ----------------------------------------------------------------------------
# NMS: tf.image.non_max_suppression (NonMaxSuppressionV3) restated in C++
# ----------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_NMS_LIB = None


def build_native(force=False):
    """Compile oracle/nms_ref.cpp -> oracle/_build/libnms_ref.so (gcc only)."""
    out = os.path.join(_HERE, "_build", "libnms_ref.so")
    src = os.path.join(_HERE, "nms_ref.cpp")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-shared", "-fPIC",
                               src, "-o", out])
    return out


def _nms_lib():
    global _NMS_LIB
    if _NMS_LIB is None:
        lib = ctypes.CDLL(build_native())
        lib.nms_ref_v3.restype = ctypes.c_int
        lib.nms_ref_v3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        _NMS_LIB = lib
    return _NMS_LIB


def nms(boxes, scores, max_output, iou_threshold, score_threshold=float("-inf")):
    """boxes [n,4] (y1,x1,y2,x2) f32, scores [n] f32 -> int32 indices in pick order."""
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float32).reshape(-1, 4))
    scores = np.ascontiguousarray(np.asarray(scores, dtype=np.float32).reshape(-1))
    n = scores.shape[0]
    out = np.zeros(max(int(max_output), 1), dtype=np.int32)
    if n == 0 or max_output <= 0:
        return out[:0]
    cnt = _nms_lib().nms_ref_v3(boxes.ctypes.data, scores.ctypes.data, n, int(max_output),
                                float(iou_threshold), float(score_threshold), out.ctypes.data)
    return out[:cnt].copy()


def nms_python(boxes, scores, max_output, iou_threshold):
    """Brute-force greedy NMS (stable: ties -> lower index first); cross-check only."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32)
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    sel = []
    f = np.float32

    def iou(i, j):
        yi0, yi1 = min(boxes[i, 0], boxes[i, 2]), max(boxes[i, 0], boxes[i, 2])
        xi0, xi1 = min(boxes[i, 1], boxes[i, 3]), max(boxes[i, 1], boxes[i, 3])
        yj0, yj1 = min(boxes[j, 0], boxes[j, 2]), max(boxes[j, 0], boxes[j, 2])
        xj0, xj1 = min(boxes[j, 1], boxes[j, 3]), max(boxes[j, 1], boxes[j, 3])
        ai = f(f(yi1 - yi0) * f(xi1 - xi0)); aj = f(f(yj1 - yj0) * f(xj1 - xj0))
        if ai <= 0 or aj <= 0:
            return f(0)
        ih = max(f(min(yi1, yj1) - max(yi0, yj0)), f(0)); iw = max(f(min(xi1, xj1) - max(xi0, xj0)), f(0))
        inter = f(ih * iw)
        return f(inter / f(f(ai + aj) - inter))
    for i in order:
        if len(sel) >= max_output:
            break
        if not np.isfinite(scores[i]) and scores[i] < 0:
            continue
        if all(not (iou(i, j) > f(iou_threshold)) for j in sel):
            sel.append(i)
    return np.asarray(sel, dtype=np.int32)


# ----------------------------------------------------------------------------
# per-image loss (SSD300.py:345-456); returns a dict with every intermediate
# ----------------------------------------------------------------------------
def smooth_l1(x):
    return torch.where(torch.abs(x) < 1., 0.5 * x * x, torch.abs(x) - 0.5)


def sparse_softmax_ce(logits, labels):
    """tf sparse_softmax_cross_entropy_with_logits: log(sum exp(z-max)) - (z_l - max)."""
    m = logits.max(dim=1, keepdim=True).values
    sh = logits - m
    lse = torch.log(torch.exp(sh).sum(dim=1))
    return lse - sh.gather(1, labels.view(-1, 1).long()).squeeze(1)


def match(anchors, gt):
    """Steps 1-6 of App. A.4. gt [pad,5] = [yc,xc,h,w,cls], padded rows -1.
    Integer outputs are the bit-exact contract for the GPU kernels."""
    a_y1x1, a_y2x2, a_yx, a_hw = anchors
    G = int(torch.argmin(gt[:, 0]).item())                                # :347 first min
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
    iou = inter / (aarea + garea - inter)                                 # [G, A]
    best = torch.argmax(iou, dim=1)                                       # first max
    A = a_yx.shape[0]
    othermask = torch.ones(A, dtype=torch.bool)
    othermask[best] = False
    other_iou = iou.t()[othermask]                                        # [A', G]
    m = other_iou.max(dim=1).values
    r = torch.argmax(other_iou, dim=1)
    pos = m > 0.5
    return dict(G=G, g_yx=g_yx, g_hw=g_hw, label=label, iou=iou, best=best,
                othermask=othermask, pos=pos, rgindex=r, max_iou=m)


def one_image_loss(p_yx, p_hw, pconf, anchors, gt, num_classes=21, detail=False):
    a_y1x1, a_y2x2, a_yx, a_hw = anchors
    mt = match(anchors, gt)
    G, best, othermask, pos = mt["G"], mt["best"], mt["othermask"], mt["pos"]
    neg = ~pos
    o_pyx, o_phw, o_conf = p_yx[othermask], p_hw[othermask], pconf[othermask]
    o_ayx, o_ahw = a_yx[othermask], a_hw[othermask]
    pos_r = mt["rgindex"][pos]
    neg_conf = o_conf[neg]
    neg_ayx, neg_ahw = o_ayx[neg], o_ahw[neg]
    neg_boxes = torch.cat([neg_ayx - neg_ahw / 2., neg_ayx + neg_ahw / 2.], dim=-1)
    num_pos = G + int(pos.sum().item())
    num_neg = int(neg.sum().item())
    k = 3 * num_pos if num_neg > 3 * num_pos else num_neg                 # :426
    neg_label = torch.full((num_neg,), num_classes - 1, dtype=torch.int64)
    total_neg_loss = sparse_softmax_ce(neg_conf, neg_label)               # :430
    sel = nms(neg_boxes.detach().numpy(), total_neg_loss.detach().numpy(), k, 0.7)   # :431
    sel_t = torch.from_numpy(sel.astype(np.int64))
    neg_loss = total_neg_loss[sel_t].mean()                               # :434

    t_pyx = torch.cat([p_yx[best], o_pyx[pos]], 0)
    t_phw = torch.cat([p_hw[best], o_phw[pos]], 0)
    t_conf = torch.cat([pconf[best], o_conf[pos]], 0)
    t_label = torch.cat([mt["label"], mt["label"][pos_r]], 0)
    t_gyx = torch.cat([mt["g_yx"], mt["g_yx"][pos_r]], 0)
    t_ghw = torch.cat([mt["g_hw"], mt["g_hw"][pos_r]], 0)
    t_ayx = torch.cat([a_yx[best], o_ayx[pos]], 0)
    t_ahw = torch.cat([a_hw[best], o_ahw[pos]], 0)
    pos_conf_loss = sparse_softmax_ce(t_conf, t_label.long()).mean()      # :445 (MEAN)
    tgt_yx = (t_gyx - t_ayx) / t_ahw
    tgt_hw = torch.log(t_ghw / t_ahw)
    yx_l = smooth_l1(t_pyx - tgt_yx).sum(-1)
    hw_l = smooth_l1(t_phw - tgt_hw).sum(-1)
    coord = (yx_l + hw_l).mean()
    total = neg_loss + pos_conf_loss + coord                              # :452
    if not detail:
        return total
    other_idx = torch.nonzero(othermask).squeeze(1)
    neg_idx = other_idx[neg]                       # original anchor ids of the negatives
    pos_idx = other_idx[pos]
    return dict(total=total, neg_loss=neg_loss, pos_conf_loss=pos_conf_loss, coord=coord,
                num_pos=num_pos, num_neg=num_neg, k=k, sel_local=sel, sel_anchor=neg_idx[sel_t],
                neg_anchor=neg_idx, pos_anchor=pos_idx, total_neg_loss=total_neg_loss,
                neg_boxes=neg_boxes, match=mt)


def batch_loss(pred, anchors, ground_truth, num_classes=21):
    """SSD300.py:129-148: sequential sum over images / batch_size."""
    n = pred.shape[0]
    loss = torch.zeros(())
    for i in range(n):
        loss = loss + one_image_loss(pred[i, :, num_classes:num_classes + 2],
                                     pred[i, :, num_classes + 2:], pred[i, :, :num_classes],
                                     anchors, ground_truth[i], num_classes)
    return loss / n


def l2_term(p, weight_decay):
    """SSD300.py:150-152: wd * sum_v ||v||^2 / 2 over ALL trainables."""
    return weight_decay * sum((p[k] ** 2).sum() / 2 for k in trainable_names(p))


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=1e-4, anchors=None):
    """One reference training step (SSD300.py:148-155). Mutates p/mom in place.
    Returns (total loss incl. L2, data loss)."""
    anchors = anchors or priors()
    names = trainable_names(p)
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    stats = {}
    pred = forward(p, images_nhwc, True, stats)
    data_loss = batch_loss(pred, anchors, ground_truth)
    loss = data_loss + l2_term(p, weight_decay)
    loss.backward()
    with torch.no_grad():
        for k in names:
            g = p[k].grad
            mom[k].mul_(0.9).add_(g)                  # accum = 0.9*accum + grad
            p[k].sub_(lr * mom[k])                    # var -= lr*accum
            p[k].requires_grad_(False)
            p[k].grad = None
        for name, (mean, var_unbiased) in stats.items():       # moving-stat update ops
            p[name + ".mmean"].mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * mean)
            p[name + ".mvar"].mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * var_unbiased)
    return float(loss.detach()), float(data_loss.detach())


# ----------------------------------------------------------------------------
# inference (SSD300.py:157-190)
# ----------------------------------------------------------------------------
def decode(pred0, anchors, num_classes=21):
    """pred0 [8828,25] -> (conf [K',20], boxes [K',4], kept row ids)."""
    _, _, a_yx, a_hw = anchors
    conf = torch.softmax(pred0[:, :num_classes], dim=-1)
    cls = torch.argmax(conf, dim=-1)
    keep = cls < num_classes - 1
    p_yx = pred0[keep, num_classes:num_classes + 2]
    p_hw = pred0[keep, num_classes + 2:]
    conf = conf[keep][:, :num_classes - 1]
    ayx, ahw = a_yx[keep], a_hw[keep]
    yx = p_yx * ahw + ayx
    hw = ahw * torch.exp(p_hw)
    boxes = torch.cat([yx - hw / 2., yx + hw / 2.], dim=-1)
    return conf, boxes, torch.nonzero(keep).squeeze(1)


def detect(pred0, anchors, score_thr, max_boxes, iou_thr, num_classes=21):
    conf, boxes, _ = decode(pred0, anchors, num_classes)
    scores, bbox, cid = [], [], []
    for c in range(num_classes - 1):
        m = conf[:, c] >= score_thr
        sc, bb = conf[m, c], boxes[m]
        idx = nms(bb.numpy(), sc.numpy(), max_boxes, iou_thr)
        idx = torch.from_numpy(idx.astype(np.int64))
        scores.append(sc[idx]); bbox.append(bb[idx])
        cid.append(torch.full((len(idx),), c, dtype=torch.int32))
    return torch.cat(scores), torch.cat(bbox, 0), torch.cat(cid)


def test_one_image(p, images_nhwc, score_thr=0.5, max_boxes=20, iou_thr=0.5, anchors=None):
    anchors = anchors or priors()
    with torch.no_grad():
        pred = forward(p, images_nhwc, False, subtract_mean=False)      # see preprocess(): test-mode quirk
        s, b, c = detect(pred[0], anchors, score_thr, max_boxes, iou_thr)
    return [s.numpy(), b.numpy(), c.numpy()]


# ----------------------------------------------------------------------------
# synthetic VOC-shaped batch
# ----------------------------------------------------------------------------
def synthetic_batch(batch, seed=0, pad_truth_to=60, max_obj=6):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(batch, INPUT_SIZE, INPUT_SIZE, 3, generator=g) * 255.
    gt = torch.full((batch, pad_truth_to, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = torch.rand(n, generator=g) * 240. + 30.
        w = torch.rand(n, generator=g) * 240. + 30.
        yc = h / 2 + torch.rand(n, generator=g) * (INPUT_SIZE - h)
        xc = w / 2 + torch.rand(n, generator=g) * (INPUT_SIZE - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return images, gt
