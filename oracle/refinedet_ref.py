"""CPU fp32 restatement of the RefineDet box side of the reference (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/RefineDet.py (class RefineDet320; input 320, or 512):
  * anchors .............................. _get_abbox(size = 4 * stride, stride, shape), ratios 0.5 / 1 / 2, (h, w) = (size sqrt(r), size / sqrt(r)),
                                           centres (i + 0.5) * stride                                                      :399-420, :140-143
                                           levels conv4_3 / conv5_3 / conv8_2 / conv10_2 with strides 8 / 16 / 32 / 64 (conv10_2 has stride 1: its
                                           map is conv9_2's, but its anchors are laid out with stride 64)                  :379-385
  * matching ............................. best anchor per ground truth (first arg-max); the others positive above IoU 0.5, negative BELOW 0.4,
                                           ignored between                                                                :422-490
  * anchor refinement module (ARM) ....... 2-way softmax, class 0 = object, class 1 = background: cross entropy of every negative against 1, hard negatives
                                           mined by tf.image.non_max_suppression (IoU 0.7, at most min(3 * positives, negatives)), mean over the mined;
                                           mean cross entropy of the positives against 0; smooth-L1 on (gt - anchor) / anchor_hw, log(gt_hw / anchor_hw),
                                           summed over the four coordinates, mean over the positives                        :519-548
  * object detection module (ODM) ........ negatives = the MINED ARM negatives whose ARM background LOGIT (sic: the logit, not the probability) is below
                                           0.99, mean cross entropy against the background class (last index); positives: mean cross entropy against
                                           the label, smooth-L1 against targets relative to the ARM-REFINED anchor
                                           arm_yx = p_arm_yx * a_hw + a_yx, arm_hw = exp(p_arm_hw) * a_hw -- no stop_gradient: the ODM box loss also
                                           back-propagates into the ARM's box outputs                                      :549-564
  * per image: arm loss + odm loss; batch: mean                                                                            :565-567, :160-180
  * inference ............................ keep anchors with softmax(arm)[1] < 0.99 and arg-max(softmax(odm)) not background; decode through both stages;
                                           per class score >= threshold, NMS                                               :189-230
Pinned against the reference's own functions run on oracle/tf_shim: tests/golden/refinedet_*.npz (tests/golden/make_golden_refinedet.py).
Only tests/ and the smoke / bench checkers may import this file.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .ssd300_ref import nms, smooth_l1

RATIOS = (0.5, 1.0, 2.0)
STRIDES = (8, 16, 32, 64)


def level_shapes(input_size):
    """feature-map sides of conv4_3, conv5_3, conv8_2, conv10_2 (three / four 2x2 SAME pools, then two stride-2 convs; conv10_2 stride 1)"""
    s = input_size
    for _ in range(3):
        s = -(-s // 2)
    out = [s]
    s = -(-s // 2); out.append(s)
    s = -(-s // 2); out.append(s)
    s = -(-s // 2); out.append(s)
    return out


def anchors(input_size=320):
    """y1x1, y2x2, yx, hw each [A, 2] float32, in the op order of _get_abbox (bit-identical to float32 TF math)"""
    f32 = np.float32
    outs = [[], [], [], []]
    for f, stride in zip(level_shapes(input_size), STRIDES):
        size = stride * 4
        ty = (np.arange(0., f, dtype=f32).reshape(-1, 1, 1, 1) + f32(0.5))
        tx = (np.arange(0., f, dtype=f32).reshape(1, -1, 1, 1) + f32(0.5))
        ty = np.tile(ty, [1, f, 1, 1]) * f32(stride)
        tx = np.tile(tx, [f, 1, 1, 1]) * f32(stride)
        tyx = np.tile(np.concatenate([ty, tx], -1), [1, 1, len(RATIOS), 1])
        pr = np.asarray([[size * (r ** 0.5), size / (r ** 0.5)] for r in RATIOS], dtype=np.float64).astype(f32).reshape(1, 1, -1, 2)
        y1x1 = (tyx - pr / f32(2.)).reshape(-1, 2)
        y2x2 = (tyx + pr / f32(2.)).reshape(-1, 2)
        for o, v in zip(outs, (y1x1, y2x2, y1x1 / f32(2.) + y2x2 / f32(2.), y2x2 - y1x1)):
            o.append(v.astype(f32))
    return tuple(torch.from_numpy(np.concatenate(o, 0)) for o in outs)


def match(anc, gt):
    """indices the GPU kernels must reproduce bit for bit: best anchor per box, status per anchor (0 ignore, 1 positive, 2 negative, 3 best), arg-max box"""
    a_y1x1, a_y2x2, a_yx, a_hw = anc
    G = int(torch.argmin(gt[:, 0]).item())
    g = gt[:G]
    g_y1x1, g_y2x2 = g[:, 0:2] - g[:, 2:4] / 2., g[:, 0:2] + g[:, 2:4] / 2.
    inter = torch.clamp(torch.minimum(a_y2x2[None], g_y2x2[:, None]) - torch.maximum(a_y1x1[None], g_y1x1[:, None]), min=0).prod(-1)
    iou = inter / (a_hw.prod(-1)[None] + g[:, 2:4].prod(-1)[:, None] - inter)          # [G, A]
    best = torch.argmax(iou, dim=1)
    A = a_yx.shape[0]
    other = torch.ones(A, dtype=torch.bool); other[best] = False
    m = iou.t().max(dim=1).values
    r = torch.argmax(iou.t(), dim=1)
    status = torch.zeros(A, dtype=torch.uint8)
    status[other & (m > 0.5)] = 1
    status[other & (m < 0.4)] = 2
    status[best] = 3
    return dict(G=G, g=g, best=best, status=status, rgindex=r, iou=iou)


def one_image_loss(arm_yx, arm_hw, arm_conf, odm_yx, odm_hw, odm_conf, anc, gt, num_classes=21, detail=False):
    """RefineDet.py:422-567; arm_conf [A,2], odm_conf [A,num_classes], the four box tensors [A,2]"""
    a_y1x1, a_y2x2, a_yx, a_hw = anc
    mt = match(anc, gt)
    g, best, status, rg = mt['g'], mt['best'], mt['status'], mt['rgindex']
    label = g[:, 4].long()
    pos = torch.nonzero(status == 1).squeeze(1)
    neg = torch.nonzero(status == 2).squeeze(1)
    tp = torch.cat([best, pos])                                                       # total positives: one row per box, then the IoU positives
    tp_label = torch.cat([label, label[rg[pos]]])
    tp_gyx = torch.cat([g[:, 0:2], g[rg[pos], 0:2]])
    tp_ghw = torch.cat([g[:, 2:4], g[rg[pos], 2:4]])
    num_pos, num_neg = tp.shape[0], neg.shape[0]
    chosen = min(3 * num_pos, num_neg)
    neg_arm_ce = F.cross_entropy(arm_conf[neg], torch.ones(num_neg, dtype=torch.long), reduction='none')
    neg_boxes = torch.cat([a_yx[neg] - a_hw[neg] / 2., a_yx[neg] + a_hw[neg] / 2.], -1)
    sel = torch.from_numpy(nms(neg_boxes.detach().numpy(), neg_arm_ce.detach().numpy(), chosen, 0.7).astype(np.int64))
    neg_armloss = neg_arm_ce[sel].mean()
    sel_rows = neg[sel]
    keep = arm_conf[sel_rows, 1] < 0.99                                                # the LOGIT, as written (:535)
    odm_neg_rows = sel_rows[keep]
    neg_odmloss = F.cross_entropy(odm_conf[odm_neg_rows], torch.full((odm_neg_rows.shape[0],), num_classes - 1, dtype=torch.long))
    pos_armconf = F.cross_entropy(arm_conf[tp], torch.zeros(num_pos, dtype=torch.long))
    t_yx = (tp_gyx - a_yx[tp]) / a_hw[tp]
    t_hw = torch.log(tp_ghw / a_hw[tp])
    pos_coord_arm = (smooth_l1(arm_yx[tp] - t_yx).sum(-1) + smooth_l1(arm_hw[tp] - t_hw).sum(-1)).mean()
    r_yx = arm_yx[tp] * a_hw[tp] + a_yx[tp]                                            # the ARM-refined anchors (gradient flows through them)
    r_hw = torch.exp(arm_hw[tp]) * a_hw[tp]
    pos_odmconf = F.cross_entropy(odm_conf[tp], tp_label)
    o_yx = (tp_gyx - r_yx) / r_hw
    o_hw = torch.log(tp_ghw / r_hw)
    pos_coord_odm = (smooth_l1(odm_yx[tp] - o_yx).sum(-1) + smooth_l1(odm_hw[tp] - o_hw).sum(-1)).mean()
    total = (neg_armloss + pos_armconf + pos_coord_arm) + (neg_odmloss + pos_odmconf + pos_coord_odm)
    if not detail:
        return total
    return dict(total=total, sel_rows=sel_rows, odm_neg_rows=odm_neg_rows, num_pos=num_pos, num_neg=num_neg, match=mt,
                parts=(neg_armloss, pos_armconf, pos_coord_arm, neg_odmloss, pos_odmconf, pos_coord_odm))


def batch_loss(arm_loc, arm_conf, odm_loc, odm_conf, anc, ground_truth, num_classes=21):
    """arm_loc / odm_loc [N,A,4] (yx, hw), arm_conf [N,A,2], odm_conf [N,A,num_classes]: sum of the per-image losses / N (:160-180)"""
    n = arm_loc.shape[0]
    tot = 0.
    for i in range(n):
        tot = tot + one_image_loss(arm_loc[i, :, :2], arm_loc[i, :, 2:], arm_conf[i], odm_loc[i, :, :2], odm_loc[i, :, 2:], odm_conf[i], anc,
                                   ground_truth[i], num_classes)
    return tot / n


def decode(arm_loc, arm_conf, odm_loc, odm_conf, anc, num_classes=21):
    """RefineDet.py:189-206 for one image: (confidence [A, num_classes - 1], boxes [A, 4] y1x1y2x2, keep mask [A]) for EVERY anchor"""
    _, _, a_yx, a_hw = anc
    armc = torch.softmax(arm_conf, -1)
    odmc = torch.softmax(odm_conf, -1)
    keep = (armc[:, 1] < 0.99) & (torch.argmax(odmc, -1) < num_classes - 1)
    r_yx = arm_loc[:, :2] * a_hw + a_yx
    r_hw = torch.exp(arm_loc[:, 2:]) * a_hw
    o_yx = odm_loc[:, :2] * r_hw + r_yx
    o_hw = torch.exp(odm_loc[:, 2:]) * r_hw
    return odmc[:, : num_classes - 1], torch.cat([o_yx - o_hw / 2., o_yx + o_hw / 2.], -1), keep


def detect(arm_loc, arm_conf, odm_loc, odm_conf, anc, score_thr, max_boxes, iou_thr, num_classes=21):
    conf, boxes, keep = decode(arm_loc, arm_conf, odm_loc, odm_conf, anc, num_classes)
    conf, boxes = conf[keep], boxes[keep]
    scores, bbox, cid = [], [], []
    for c in range(num_classes - 1):
        m = conf[:, c] >= score_thr
        sc, bb = conf[m, c], boxes[m]
        idx = torch.from_numpy(nms(bb.numpy(), sc.numpy(), max_boxes, iou_thr).astype(np.int64))
        scores.append(sc[idx]); bbox.append(bb[idx]); cid.append(torch.full((len(idx),), c, dtype=torch.int32))
    return torch.cat(scores), torch.cat(bbox, 0), torch.cat(cid)


def synthetic_gt(batch, input_size, seed, pad=60, max_obj=6):
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = torch.rand(n, generator=g) * (input_size * 0.7) + input_size * 0.1
        w = torch.rand(n, generator=g) * (input_size * 0.7) + input_size * 0.1
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        gt[i, :n] = torch.stack([yc, xc, h, w, torch.randint(0, 20, (n,), generator=g).float()], 1)
    return gt
