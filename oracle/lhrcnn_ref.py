"""CPU fp32 restatement of the reference's Light-Head R-CNN (TEST INFRASTRUCTURE ONLY): network, both losses, the training step, inference.

Follows /root/reference/LH_RCNN.py (class LHRCNN; driver testlhrcnn.py: 700 x 1100, batch 32):
  * input ................................ images / 127.5 - 1 (:58-72)
  * backbone 'feature_extractor' ......... :203-231  conv1 3x3/s2 (24) + 3x3/s2 SAME max pool; three stages of [3x3/s2 conv] + 3 / 7 / 3 separable 3x3 blocks
                                                      (144, 288, 576 channels); every block conv(bias) | separable(no bias) -> batch norm -> ReLU (:537-567); stride 32
  * RPN 'rpn' ............................ :77-97    3x3 conv (256) + BN + ReLU; 3x3 conv -> 15 * 2 scores and 3x3 conv -> 15 * 4 box codes, both + BN, no activation;
                                                      15 anchors per cell (5 scales x 3 ratios, :31-33, :240-261), only anchors inside the picture are kept (:87-97)
  * light head 'rcnn' .................... :99-104   two branches of separable [1x15] (256) + separable [15x1] (490), BN + ReLU each, summed: 490 = 10 * 7 * 7 channels
  * RPN loss per image ................... :263-440  (rpn_one_image below)
  * R-CNN stage .......................... :140-170  proposals clamped to the picture, tf.image.crop_and_resize to 7 x 7 (bilinear, NOT position sensitive), flatten,
                                                      dense 2048 + ReLU, dense classes + 1 and dense 4; softmax cross entropy (background = LAST class) + smooth L1
  * optimizer ............................ :171-201  ONE MomentumOptimizer(0.9); rpn_loss (+ wd * l2 of the backbone's and the RPN's trainables) drives the
                                                      backbone + RPN variables, rcnn_loss (+ wd * l2 of the 'rcnn' trainables) drives the 'rcnn' variables
  * inference ............................ :134-138, :153-164, :203-236

Three things in that file do not do what they look like; each is restated AS TENSORFLOW EXECUTES IT (TF 1.13, GPU kernels), not as it reads:
  1. :194-197 `tf.case([(step < rpn_first_step, lambda: train_rpn_op), ...])`: both train ops were built outside the case, so neither is gated by a
     predicate -- a TF-1.x graph runs BOTH of them on every step (cond_v1 only hangs a control edge from the selected branch's pivot onto the op).  Every
     step therefore updates backbone + RPN from rpn_loss AND the light head from rcnn_loss, and global_step (incremented by train_rcnn_op, :192) advances
     every step.  Only the REPORTED loss (:200-203, a tensor selected by the same predicates) follows the schedule.
  2. :337 `best_rcnn_label = tf.gather(rcnn_label, best_raindex)` indexes the G ground-truth labels with ANCHOR indices.  tf.gather's documented device
     difference: the CPU kernel raises InvalidArgument, the GPU kernel stores 0 for an out-of-range index.  Restated with the GPU behaviour (label 0, or
     label[index] when the anchor index happens to be < G); the CPU behaviour is "the step aborts".
  3. :430 the R-CNN centre target is (g_yx - proposal_yx) / proposal_yx -- divided by the proposal's CENTRE, not its size.  Kept.
Parameters by layer name: conv '.w' [K,R,S,C] + '.b'; separable '.dw' [kh,kw,C] + '.w' [K,1,1,C] (no bias); both with '.gamma', '.beta', '.mmean', '.mvar';
dense '.w' [units, in] + '.b'.  Pinned against the reference's own class run on oracle/tf_shim with the GPU gather switch on:
tests/golden/lhrcnn_train.npz, lhrcnn_detect.npz (tests/golden/make_golden_lhrcnn.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .ssd300_ref import BN_EPS, conv2d_same, maxpool_same, nms, smooth_l1, sparse_softmax_ce

ANCHOR_SCALES = [32, 64, 128, 256, 512]
ANCHOR_RATIOS = [0.5, 1.0, 2.0]
NA = len(ANCHOR_SCALES) * len(ANCHOR_RATIOS)
STRIDE = 32.0
CROP = 7
HEAD_CH = 490


def layer_specs(num_classes=21):
    """[(name, kind, cin, cout, kh, kw, stride, relu)] in TensorFlow's creation order; kind 'conv' | 'sep' (both + batch norm) | 'dense'"""
    s = [('conv1', 'conv', 3, 24, 3, 3, 2, True)]
    c = 24
    for stage, ch, nsep in ((2, 144, 3), (3, 288, 7), (4, 576, 3)):
        s.append((f'stage{stage}_sconv1', 'conv', c, ch, 3, 3, 2, True))
        for j in range(2, nsep + 2):
            s.append((f'stage{stage}_sconv{j}', 'sep', ch, ch, 3, 3, 1, True))
        c = ch
    s.append(('rpn_conv', 'conv', 576, 256, 3, 3, 1, True))
    s.append(('rpn_conf', 'conv', 256, NA * 2, 3, 3, 1, False))
    s.append(('rpn_pbbox', 'conv', 256, NA * 4, 3, 3, 1, False))
    for b in (1, 2):
        s.append((f'state5_conv{b}_1', 'sep', 576, 256, 1, 15, 1, True))
        s.append((f'state5_conv{b}_2', 'sep', 256, HEAD_CH, 15, 1, 1, True))
    s.append(('roi_feat_dense', 'dense', CROP * CROP * HEAD_CH, 2048, 1, 1, 1, True))
    s.append(('rcnn_pconf', 'dense', 2048, num_classes, 1, 1, 1, False))
    s.append(('rcnn_pbbox', 'dense', 2048, 4, 1, 1, 1, False))
    return s


BACKBONE = [s[0] for s in layer_specs() if s[0].startswith(('conv1', 'stage'))]
RPN_LAYERS = ['rpn_conv', 'rpn_conf', 'rpn_pbbox']
RCNN_LAYERS = [s[0] for s in layer_specs() if s[0].startswith(('state5', 'roi_feat', 'rcnn_'))]


def init_params(seed=0, num_classes=21, dense_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, kind, cin, cout, kh, kw, _, _ in layer_specs(num_classes):
        if kind == 'dense':
            p[name + '.w'] = torch.randn(cout, cin, generator=g) * (dense_scale * math.sqrt(2.0 / cin))
            p[name + '.b'] = 0.05 * torch.randn(cout, generator=g)
            continue
        if kind == 'sep':
            p[name + '.dw'] = torch.randn(kh, kw, cin, generator=g) * math.sqrt(2.0 / (kh * kw))
            p[name + '.w'] = torch.randn(cout, 1, 1, cin, generator=g) * math.sqrt(2.0 / cin)
        else:
            p[name + '.w'] = torch.randn(cout, kh, kw, cin, generator=g) * math.sqrt(2.0 / (cin * kh * kw))
            p[name + '.b'] = torch.zeros(cout)
        p[name + '.gamma'] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        p[name + '.beta'] = 0.1 * torch.randn(cout, generator=g)
        p[name + '.mmean'] = torch.zeros(cout)
        p[name + '.mvar'] = torch.ones(cout)
    return p


def trainable_names(p, group=None):
    """group None | 'rpn' (backbone + RPN: the variables train_rpn_op moves, :188) | 'rcnn' (:190)"""
    rc = tuple(RCNN_LAYERS)
    out = []
    for k in p:
        if k.endswith(('.mmean', '.mvar')):
            continue
        is_rcnn = k.rsplit('.', 1)[0] in rc
        if group is None or (group == 'rcnn') == is_rcnn:
            out.append(k)
    return out


# ---------------------------------------------------------------------------- network
class _Net:
    def __init__(self, p, training, stats, taps):
        self.p, self.training, self.stats, self.taps = p, training, stats, taps
        self.spec = {s[0]: s for s in layer_specs(p['rcnn_pconf.w'].shape[0])}

    def __call__(self, name, x):
        _, kind, cin, cout, kh, kw, stride, relu = self.spec[name]
        p = self.p
        if kind == 'sep':
            wd = p[name + '.dw'].permute(2, 0, 1).unsqueeze(1)                          # [C,1,kh,kw]
            z = F.conv2d(F.pad(x, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2)), wd, None, groups=cin)   # SAME, stride 1
            z = F.conv2d(z, p[name + '.w'].permute(0, 3, 1, 2), None)
        else:
            z = conv2d_same(x, p[name + '.w'], p[name + '.b'], stride, 1)
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


def preprocess(images_nhwc):
    return images_nhwc.float() / 127.5 - 1.


def backbone(net, images_nhwc, normalise=True):
    x = (preprocess(images_nhwc) if normalise else images_nhwc.float()).permute(0, 3, 1, 2)
    x = net('conv1', x)
    x = maxpool_same(x, 3, 2)
    for name in BACKBONE[1:]:
        x = net(name, x)
    return x


def forward(p, images_nhwc, training, stats_out=None, taps=None, detach_c4_for_head=True, normalise=True):
    """-> c4 [N,576,fh,fw], rpn_conf [N, fh*fw*15, 2], rpn_bbox [N, fh*fw*15, 4], rcnn_feat NHWC [N,fh,fw,490].
    detach_c4_for_head: the light head's loss never reaches the backbone (its optimizer only holds the 'rcnn' variables)."""
    net = _Net(p, training, stats_out, taps)
    c4 = backbone(net, images_nhwc, normalise)
    r = net('rpn_conv', c4)
    n = c4.shape[0]
    conf = net('rpn_conf', r).permute(0, 2, 3, 1).reshape(n, -1, 2)
    bbox = net('rpn_pbbox', r).permute(0, 2, 3, 1).reshape(n, -1, 4)
    h = c4.detach() if detach_c4_for_head else c4
    feat = net('state5_conv1_2', net('state5_conv1_1', h)) + net('state5_conv2_2', net('state5_conv2_1', h))
    return c4, conf, bbox, feat.permute(0, 2, 3, 1)


def anchors(fh, fw, img_h, img_w):
    """:240-261 + the inside-the-picture mask of :87-97.  -> dict(y1x1, y2x2, yx, hw: [A', 2] of the kept anchors; keep: [fh*fw*15] bool).
    Anchor (i, j, k) sits at ((i + .5) * 32, (j + .5) * 32) with size (s * sqrt(r), s / sqrt(r)), k = scale-major over (scale, ratio); kept when
    y1, x1 >= 0 and y2 <= (H - 1) - 1, x2 <= (W - 1) - 1 (self.h = H - 1, and the mask subtracts one more, :88-89)."""
    f32 = torch.float32
    cy = (torch.arange(fh, dtype=f32) + 0.5).view(fh, 1, 1, 1).expand(fh, fw, NA, 1)
    cx = (torch.arange(fw, dtype=f32) + 0.5).view(1, fw, 1, 1).expand(fh, fw, NA, 1)
    centre = torch.cat([cy, cx], -1) * STRIDE
    sizes = torch.tensor([[s * (r ** 0.5), s / (r ** 0.5)] for s in ANCHOR_SCALES for r in ANCHOR_RATIOS], dtype=f32).view(1, 1, NA, 2)
    y1x1 = (centre - sizes / 2.).reshape(-1, 2)
    y2x2 = (centre + sizes / 2.).reshape(-1, 2)
    yx = y1x1 / 2. + y2x2 / 2.
    hw = y2x2 - y1x1
    h, w = float(img_h - 1), float(img_w - 1)
    keep = (y1x1[:, 0] >= 0.) & (y1x1[:, 1] >= 0.) & (y2x2[:, 0] <= h - 1) & (y2x2[:, 1] <= w - 1)
    return dict(y1x1=y1x1[keep], y2x2=y2x2[keep], yx=yx[keep], hw=hw[keep], keep=keep)


# ---------------------------------------------------------------------------- RPN loss of one image (:263-440)
def num_real(gt):
    """:265 `tf.argmin(ground_truth, axis=0)[0]`: the first row holding the smallest yc -- the first -1 padding row"""
    col = gt[:, 0]
    return int(torch.nonzero(col == col.min())[0, 0])


def rpn_one_image(p_yx, p_hw, pconf, anc, gt, gather_oob_zero=True, detail=False):
    """p_yx, p_hw [A',2], pconf [A',2] of the kept anchors; gt [P,5] = yc, xc, h, w, class padded with -1.
    -> loss, pos_proposal [Kp,4], pos_label [Kp] int, rcnn_truth [Kp,4], neg_proposal [Kn,4]  (+ the index lists with detail=True)"""
    G = num_real(gt)
    g = gt[:G]
    g_yx, g_hw, label = g[:, 0:2], g[:, 2:4], g[:, 4].to(torch.int32)
    g_y1x1, g_y2x2 = g_yx - g_hw / 2., g_yx + g_hw / 2.
    a_y1x1, a_y2x2, a_yx, a_hw = anc['y1x1'], anc['y2x2'], anc['yx'], anc['hw']
    A = a_yx.shape[0]
    lo = torch.maximum(a_y1x1.unsqueeze(0), g_y1x1.unsqueeze(1))                      # [G, A, 2]
    hi = torch.minimum(a_y2x2.unsqueeze(0), g_y2x2.unsqueeze(1))
    inter = torch.clamp(hi - lo, min=0).prod(-1)
    iou = inter / (a_hw.prod(-1).unsqueeze(0) + g_hw.prod(-1).unsqueeze(1) - inter + 1e-8)
    best_a = iou.argmax(dim=1)                                                        # first maximum: one anchor per ground-truth box (duplicates allowed)
    if gather_oob_zero:
        ok = best_a < G
        best_label = torch.where(ok, label[best_a.clamp(max=max(G - 1, 0))], torch.zeros_like(label))   # :337 on the GPU gather kernel
    else:
        best_label = label[best_a]                                                    # IndexError = the CPU kernel's InvalidArgument
    is_best = torch.zeros(A, dtype=torch.bool)
    is_best[best_a] = True
    other = torch.nonzero(~is_best).squeeze(1)                                        # anchor order
    o_iou = iou.t()[other]                                                            # [A'', G]
    o_max, o_arg = o_iou.max(dim=1).values, o_iou.argmax(dim=1)
    pos_o, neg_o = other[o_max > 0.5], other[o_max < 0.3]
    pos_g = o_arg[o_max > 0.5]
    pos_a = torch.cat([best_a, pos_o])                                                # candidate positives: the G best anchors, then the IoU > 0.5 ones
    pos_gi = torch.cat([torch.arange(G), pos_g])
    pos_label = torch.cat([best_label, label[pos_g]])
    n_pos, n_neg = pos_a.shape[0], neg_o.shape[0]
    k_pos = min(n_pos, 128)
    k_neg = min(n_neg, 256 - k_pos)

    def box(idx):
        return torch.cat([a_yx[idx] - a_hw[idx] / 2., a_yx[idx] + a_hw[idx] / 2.], -1)
    pos_ce = sparse_softmax_ce(pconf[pos_a], torch.zeros(n_pos, dtype=torch.long))    # "object" is class 0 of the RPN's two outputs
    sel_p = torch.from_numpy(nms(box(pos_a).detach().numpy(), torch.softmax(pconf[pos_a], -1)[:, 0].detach().numpy(), k_pos, 0.7).astype(np.int64))
    neg_ce = sparse_softmax_ce(pconf[neg_o], torch.ones(n_neg, dtype=torch.long))
    sel_n = torch.from_numpy(nms(box(neg_o).detach().numpy(), neg_ce.detach().numpy(), k_neg, 0.7).astype(np.int64))
    pa, pg = pos_a[sel_p], pos_gi[sel_p]
    na_ = neg_o[sel_n]
    t_yx = (g_yx[pg] - a_yx[pa]) / a_hw[pa]
    t_hw = torch.log(g_hw[pg] / a_hw[pa])
    coord = (smooth_l1(p_yx[pa] - t_yx).sum(-1) + smooth_l1(p_hw[pa] - t_hw).sum(-1)).mean()
    loss = neg_ce[sel_n].mean() + pos_ce[sel_p].mean() + 10. * coord
    pr_yx = a_hw[pa] * p_yx[pa] + a_yx[pa]
    pr_hw = torch.exp(p_hw[pa]) * a_hw[pa]
    truth = torch.cat([(g_yx[pg] - pr_yx) / pr_yx, torch.log(g_hw[pg] / pr_hw)], -1)  # :430: divided by the proposal's centre
    nr_yx = a_hw[na_] * p_yx[na_] + a_yx[na_]
    nr_hw = torch.exp(p_hw[na_]) * a_hw[na_]
    out = (loss, torch.cat([pr_yx - pr_hw / 2., pr_yx + pr_hw / 2.], -1), pos_label[sel_p], truth, torch.cat([nr_yx - nr_hw / 2., nr_yx + nr_hw / 2.], -1))
    if detail:
        return out + (dict(G=G, best_a=best_a, pos_a=pos_a, pos_gi=pos_gi, neg_o=neg_o, sel_p=sel_p, sel_n=sel_n, pa=pa, pg=pg, na=na_, iou=iou),)
    return out


# ---------------------------------------------------------------------------- R-CNN stage
def crop_and_resize(feat_nhwc, boxes, box_ind, crop=CROP):
    """tf.image.crop_and_resize, bilinear, extrapolation 0 (crop_and_resize_op.cc): sample row y of a box = y1 (H-1) + y (y2 - y1)(H-1) / (crop - 1);
    outside [0, H-1] -> 0; top/bottom = floor/ceil rows, value = top + (bottom - top) * lerp of the x-interpolated rows.  Differentiable in feat."""
    n, H, W, C = feat_nhwc.shape
    R = boxes.shape[0]
    if R == 0:
        return feat_nhwc.new_zeros((0, crop, crop, C))
    b = boxes.detach()
    grid = torch.arange(crop, dtype=torch.float32, device=b.device).view(1, crop)
    if crop > 1:
        # the divisor is a TENSOR on purpose: torch turns `tensor / python_scalar` into a multiplication by the rounded reciprocal on the GPU, one ulp off the
        # division the TensorFlow kernel (and odtk_crop_and_resize_*) performs -- enough to move a sample across the picture's last row when a box ends at 1.0
        steps = torch.full((1, 1), float(crop - 1), dtype=torch.float32, device=b.device)
        in_y = b[:, 0:1] * (H - 1) + grid * ((b[:, 2:3] - b[:, 0:1]) * (H - 1) / steps)
        in_x = b[:, 1:2] * (W - 1) + grid * ((b[:, 3:4] - b[:, 1:2]) * (W - 1) / steps)
    else:                                                         # one sample at the centre of the box
        in_y, in_x = 0.5 * (b[:, 0:1] + b[:, 2:3]) * (H - 1), 0.5 * (b[:, 1:2] + b[:, 3:4]) * (W - 1)
    ok = ((in_y >= 0) & (in_y <= H - 1)).view(R, crop, 1, 1) & ((in_x >= 0) & (in_x <= W - 1)).view(R, 1, crop, 1)
    y0, x0 = torch.floor(in_y), torch.floor(in_x)
    ly, lx = (in_y - y0).view(R, crop, 1, 1), (in_x - x0).view(R, 1, crop, 1)
    y0i, y1i = y0.clamp(0, H - 1).long().view(R, crop, 1), torch.ceil(in_y).clamp(0, H - 1).long().view(R, crop, 1)
    x0i, x1i = x0.clamp(0, W - 1).long().view(R, 1, crop), torch.ceil(in_x).clamp(0, W - 1).long().view(R, 1, crop)
    bi = box_ind.long().view(R, 1, 1)
    tl, tr, bl, br = feat_nhwc[bi, y0i, x0i], feat_nhwc[bi, y0i, x1i], feat_nhwc[bi, y1i, x0i], feat_nhwc[bi, y1i, x1i]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return torch.where(ok, top + (bot - top) * ly, feat_nhwc.new_zeros(()))


def head(p, roi_flat):
    x = F.relu(roi_flat @ p['roi_feat_dense.w'].t() + p['roi_feat_dense.b'])
    return x @ p['rcnn_pconf.w'].t() + p['rcnn_pconf.b'], x @ p['rcnn_pbbox.w'].t() + p['rcnn_pbbox.b']


def l2_term(p, names):
    return sum((p[k] * p[k]).sum() / 2 for k in names)


def losses(p, images_nhwc, ground_truth, weight_decay=1e-4, stats_out=None, gather_oob_zero=True, detail=False):
    """-> rpn_loss, rcnn_loss (each with its weight-decay term, :176-183) of a training-mode pass"""
    n, H, W, _ = images_nhwc.shape
    c4, conf, bbox, feat = forward(p, images_nhwc, True, stats_out)
    anc = anchors(c4.shape[2], c4.shape[3], H, W)
    keep = anc['keep']
    per, pos_prop, pos_lab, truth, neg_prop, pos_ind, neg_ind, det = [], [], [], [], [], [], [], []
    for i in range(n):
        r = rpn_one_image(bbox[i, keep, :2], bbox[i, keep, 2:], conf[i, keep], anc, ground_truth[i], gather_oob_zero, detail)
        per.append(r[0]); pos_prop.append(r[1]); pos_lab.append(r[2]); truth.append(r[3]); neg_prop.append(r[4])
        pos_ind.append(torch.full((r[1].shape[0],), i)); neg_ind.append(torch.full((r[4].shape[0],), i))
        if detail:
            det.append(r[5])
    rpn_loss = torch.stack(per).mean() + weight_decay * l2_term(p, trainable_names(p, 'rpn'))
    h, w = float(H - 1), float(W - 1)
    lim = torch.tensor([h, w, h, w])
    pos_prop, neg_prop = torch.cat(pos_prop).detach(), torch.cat(neg_prop).detach()
    boxes = torch.cat([torch.minimum(torch.clamp(pos_prop, min=0.), lim), torch.minimum(torch.clamp(neg_prop, min=0.), lim)]) / lim
    ind = torch.cat(pos_ind + neg_ind)
    num_classes = p['rcnn_pconf.w'].shape[0]
    labels = torch.cat([torch.cat(pos_lab).long(), torch.full((neg_prop.shape[0],), num_classes - 1, dtype=torch.long)])
    roi = crop_and_resize(feat, boxes, ind).reshape(boxes.shape[0], -1)
    pconf, pbbox = head(p, roi)
    kp = pos_prop.shape[0]
    rcnn_loss = sparse_softmax_ce(pconf, labels).mean() + smooth_l1(pbbox[:kp] - torch.cat(truth).detach()).sum(-1).mean()
    rcnn_loss = rcnn_loss + weight_decay * l2_term(p, trainable_names(p, 'rcnn'))
    if detail:
        return rpn_loss, rcnn_loss, dict(images=det, boxes=boxes, box_ind=ind, labels=labels, truth=torch.cat(truth).detach(), num_pos=kp, roi=roi, pconf=pconf,
                                         pbbox=pbbox, rpn_conf=conf, rpn_bbox=bbox, feat=feat, keep=keep)
    return rpn_loss, rcnn_loss


def train_step(p, mom, images_nhwc, ground_truth, lr, weight_decay=1e-4, gather_oob_zero=True):
    """one step as the TF-1.x graph executes it: BOTH optimizer ops (header, item 1), then the batch-norm moving statistics.  -> (rpn_loss, rcnn_loss)"""
    names_rpn, names_rcnn = trainable_names(p, 'rpn'), trainable_names(p, 'rcnn')
    for k in names_rpn + names_rcnn:
        p[k].requires_grad_(True)
    stats = {}
    rpn_loss, rcnn_loss = losses(p, images_nhwc, ground_truth, weight_decay, stats, gather_oob_zero)
    g_rpn = torch.autograd.grad(rpn_loss, [p[k] for k in names_rpn], retain_graph=True, allow_unused=True)
    g_rcnn = torch.autograd.grad(rcnn_loss, [p[k] for k in names_rcnn], allow_unused=True)
    with torch.no_grad():
        for names, grads in ((names_rpn, g_rpn), (names_rcnn, g_rcnn)):
            for k, g in zip(names, grads):
                g = torch.zeros_like(p[k]) if g is None else g
                mom[k].mul_(0.9).add_(g)
                p[k].sub_(lr * mom[k])
        for name, (mean, unb) in stats.items():
            p[name + '.mmean'].mul_(0.99).add_(mean * 0.01)
            p[name + '.mvar'].mul_(0.99).add_(unb * 0.01)
    for k in names_rpn + names_rcnn:
        p[k].requires_grad_(False)
    return float(rpn_loss.detach()), float(rcnn_loss.detach())


# ---------------------------------------------------------------------------- inference (:134-138, :153-164, :203-236)
def detect(p, images_nhwc, score_thr=0.5, max_boxes=20, iou_thr=0.45, post_nms_proposal=500, detail=False, normalise=False):
    """one image [1,H,W,3] -> scores [K], bbox [K,4] (y1, x1, y2, x2), class_id [K] int32, class-major as the reference concatenates them.
    normalise=False is the reference: test_one_image feeds `self.images`, which at that point names the tensor AFTER `/ 127.5 - 1` (:68-69, :467) -- the fed
    pixels bypass the normalisation, the caller has to hand over normalised pictures."""
    assert images_nhwc.shape[0] == 1
    _, H, W, _ = images_nhwc.shape
    with torch.no_grad():
        c4, conf, bbox, feat = forward(p, images_nhwc, False, normalise=normalise)
        anc = anchors(c4.shape[2], c4.shape[3], H, W)
        keep = anc['keep']
        yx = bbox[0, keep, :2] * anc['hw'] + anc['yx']
        hw = torch.exp(bbox[0, keep, 2:]) * anc['hw']
        h, w = float(H - 1), float(W - 1)
        lim = torch.tensor([h, w, h, w])
        prop = torch.minimum(torch.clamp(torch.cat([yx - hw / 2., yx + hw / 2.], -1), min=0.), lim)
        score = torch.softmax(conf[0, keep], -1)[:, 0]
        sel = torch.from_numpy(nms(prop.numpy(), score.numpy(), post_nms_proposal, 0.7).astype(np.int64))
        prop = prop[sel]
        p_yx, p_hw = prop[:, 0:2] / 2. + prop[:, 2:4] / 2., prop[:, 2:4] - prop[:, 0:2]
        roi = crop_and_resize(feat, prop / lim, torch.zeros(prop.shape[0])).reshape(prop.shape[0], -1)
        pconf, pbbox = head(p, roi)
        cf = torch.softmax(pconf, -1)
        num_classes = cf.shape[1]
        fg = cf.argmax(-1) < num_classes - 1
        cf, pb, p_yx, p_hw = cf[fg], pbbox[fg], p_yx[fg], p_hw[fg]
        d_yx = pb[:, 0:2] * p_hw + p_yx
        d_hw = p_hw * torch.exp(pb[:, 2:4])
        boxes = torch.cat([d_yx - d_hw / 2., d_yx + d_hw / 2.], -1)
        s_out, b_out, c_out = [], [], []
        for c in range(num_classes - 1):
            m = cf[:, c] >= score_thr
            sc, bx = cf[m, c], boxes[m]
            k = torch.from_numpy(nms(bx.numpy(), sc.numpy(), max_boxes, iou_thr).astype(np.int64))
            s_out.append(sc[k]); b_out.append(bx[k]); c_out.append(torch.full((k.shape[0],), c, dtype=torch.int32))
        res = (torch.cat(s_out), torch.cat(b_out, 0).reshape(-1, 4), torch.cat(c_out))
    if detail:
        return res + (dict(proposals=prop, selected=sel, pconf=pconf, pbbox=pbbox, rpn_conf=conf, rpn_bbox=bbox, feat=feat, foreground=fg, keep=keep),)
    return res


def synthetic_gt(batch, H, W, seed, pad=6, max_obj=3):
    """[batch, pad, 5] = yc, xc, h, w, class, padded with -1 (at least one padding row: :265 finds the count by arg-min)"""
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = (0.25 + 0.45 * torch.rand(n, generator=g)) * H
        w = (0.25 + 0.45 * torch.rand(n, generator=g)) * W
        yc = h / 2 + torch.rand(n, generator=g) * (H - 1 - h)
        xc = w / 2 + torch.rand(n, generator=g) * (W - 1 - w)
        gt[i, :n] = torch.stack([yc, xc, h, w, torch.randint(0, 20, (n,), generator=g).float()], 1)
    return gt
