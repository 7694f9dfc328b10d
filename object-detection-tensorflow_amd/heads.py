"""Inference tails of the other detector classes of the reference on libodtk: decode kernel -> per-class score threshold ->
per-class NMS (the SSD300 NMS path) -> [scores f32[K], bbox f32[K,4] y1x1y2x2 px, class_id i32[K]], i.e. what each class
stores in `self.detection_pred` (RetinaNet.py:224-256, YOLOv3.py:320-368, FCOS.py:197-265, CenterNet.py:159-185, RefineDet.py:189-230);
for RefineDet also the training-side chain matching -> ARM hard-negative mining -> two-stage loss (`refinedet_loss`).
Inputs are the head outputs of ONE image as device tensors.  Product path: no CPU fallback, no use of the test oracles."""
from __future__ import annotations

import torch

from . import ops

NMS_MAX_CANDIDATES = 32768          # odtk_nms_batched: boxes per problem


def _per_class_nms(conf, boxes, cand, num_classes, max_boxes, iou_thr):
    """conf [L, ld] scores (classes 0..num_classes-1 in the first columns), boxes [L, 4], cand [L, ld] uint8."""
    dev = conf.device
    L, ld = conf.shape
    rows = None
    if L > NMS_MAX_CANDIDATES:
        # keep the rows that are a candidate for at least one class, in order (ties in NMS follow the row order)
        rows = torch.nonzero(cand[:, :num_classes].any(dim=1)).flatten()
        if rows.numel() > NMS_MAX_CANDIDATES:
            raise ValueError(f'{rows.numel()} candidate boxes exceed the NMS capacity of {NMS_MAX_CANDIDATES}; raise the score threshold')
        if rows.numel() == 0:
            return (torch.empty(0, device=dev), torch.empty(0, 4, device=dev), torch.empty(0, dtype=torch.int32, device=dev))
        conf, boxes, cand = conf[rows].contiguous(), boxes[rows].contiguous(), cand[rows].contiguous()
        L = rows.numel()
    cap = max(int(max_boxes), 1)
    out_idx = torch.zeros(num_classes, cap, dtype=torch.int32, device=dev)
    out_cnt = torch.zeros(num_classes, dtype=torch.int32, device=dev)
    ops.nms_batched(boxes, 0, conf, 1, ld, cand, 1, ld, 1, L, num_classes, None, 0, int(max_boxes), float(iou_thr), out_idx, cap, out_cnt)
    cnt = out_cnt.cpu().tolist()
    scores, bbox, cid = [], [], []
    for c in range(num_classes):                      # ascending class id, NMS pick order inside (the reference's concat order)
        ids = out_idx[c, : cnt[c]].long()
        scores.append(conf[ids, c]); bbox.append(boxes[ids])
        cid.append(torch.full((cnt[c],), c, dtype=torch.int32, device=dev))
    return torch.cat(scores), torch.cat(bbox, 0).reshape(-1, 4), torch.cat(cid)


def retina_detect(pconf, pbox, anchors_yx, anchors_hw, score_thr, max_boxes, iou_thr):
    """RetinaNet.py:224-256.  pconf [A, C] logits (last class = background), pbox [A, 4] = (dy, dx, log h, log w)."""
    conf, boxes, _, cand = ops.retina_decode(pconf, pbox, anchors_yx, anchors_hw, score_thr)
    return _per_class_nms(conf, boxes, cand, pconf.shape[1] - 1, max_boxes, iou_thr)


def refinedet_anchors(input_size, device):
    """RefineDet.py:140-143, :399-420: levels conv4_3 / conv5_3 / conv8_2 / conv10_2, strides 8 / 16 / 32 / 64, size = 4 * stride, ratios 0.5 / 1 / 2.
    Returns (y1x1, y2x2, yx, hw, nmsbox) device tensors.  The SSD300 prior kernel is reused: (i + 0.5) * input / side equals (i + 0.5) * stride bit for
    bit here (input / side = stride exactly, every product exact in float32)."""
    s = input_size
    sides = []
    for _ in range(3):
        s = -(-s // 2)
    sides.append(s)
    for _ in range(3):
        s = -(-s // 2)
        sides.append(s)
    flat = []
    for side, stride in zip(sides, (8, 16, 32, 64)):
        assert side * stride == input_size, "RefineDet: the input size must be a multiple of 64"
        size = 4 * stride
        for r in (0.5, 1.0, 2.0):
            flat += [size * (r ** 0.5), size / (r ** 0.5)]
    return ops.ssd_priors(input_size, sides, [3, 3, 3, 3], flat, device)


class RefineDetLoss:
    """RefineDet.py:422-567 for a batch: odtk_retina_match -> odtk_softmax_ce_const (ARM background cross entropy) -> odtk_nms_batched (hard negatives)
    -> odtk_refinedet_loss.  Buffers are allocated once per (N, A, C, P)."""

    def __init__(self, anchors, N, num_classes, P, device):
        self.anc, self.N, self.C, self.P = anchors, N, num_classes, P
        A = anchors[0].shape[0]
        self.A = A
        i32 = dict(dtype=torch.int32, device=device)
        self.ngt = torch.zeros(N, **i32); self.best = torch.zeros(N, P, **i32)
        self.status = torch.zeros(N, A, dtype=torch.uint8, device=device); self.rg = torch.zeros(N, A, **i32)
        self.counts = torch.zeros(N, 4, **i32)
        self.ws = ops.retina_match_workspace(A, N, P, device)
        self.negloss = torch.zeros(N, A, device=device)
        self.sel_idx = torch.zeros(N, A, **i32); self.sel_cnt = torch.zeros(N, **i32)
        self.loss_parts = torch.zeros(N, 8, device=device)
        self.d_arm_loc = torch.zeros(N, A, 4, device=device); self.d_arm_conf = torch.zeros(N, A, 2, device=device)
        self.d_odm_loc = torch.zeros(N, A, 4, device=device); self.d_odm_conf = torch.zeros(N, A, num_classes, device=device)

    def __call__(self, arm_loc, arm_conf, odm_loc, odm_conf, gt, grad_scale):
        """-> loss_parts [N, 8] (column 6 = the per-image loss); the gradients are in self.d_*"""
        y1x1, y2x2, yx, hw, nmsbox = self.anc
        N, A = self.N, self.A
        ops.retina_match(y1x1, y2x2, hw, gt, self.ngt, self.best, self.status, self.rg, self.counts, self.ws)
        ops.softmax_ce_const(arm_conf, N * A, 2, 2, 1, self.negloss)
        ops.nms_batched(nmsbox, 0, self.negloss, A, 1, self.status, A, 1, 2, A, N, self.counts[:, 2:], 4, 0, 0.7, self.sel_idx, A, self.sel_cnt)
        ops.refinedet_loss(arm_loc, arm_conf, odm_loc, odm_conf, yx, hw, gt, self.ngt, self.best, self.status, self.rg, self.counts, self.negloss,
                           self.sel_idx, self.sel_cnt, grad_scale, self.loss_parts, self.d_arm_loc, self.d_arm_conf, self.d_odm_loc, self.d_odm_conf)
        return self.loss_parts


def refinedet_detect(arm_loc, arm_conf, odm_loc, odm_conf, anchors_yx, anchors_hw, score_thr, max_boxes, iou_thr):
    """RefineDet.py:189-230 for one image: arm_loc / odm_loc [A, 4], arm_conf [A, 2], odm_conf [A, C] (last class = background)."""
    conf, boxes, _, cand = ops.refinedet_decode(arm_loc, arm_conf, odm_loc, odm_conf, anchors_yx, anchors_hw, score_thr)
    return _per_class_nms(conf, boxes, cand, odm_conf.shape[1] - 1, max_boxes, iou_thr)


def fcos_detect(conf, reg, center, score_thr, max_boxes, iou_thr):
    """FCOS.py:197-265.  conf / reg / center: the five level tensors [H, W, C | 4 | 1] of one image; classes 0..C-2."""
    pconf, pbbox = ops.fcos_decode_candidates(conf, reg, center)
    cand = (pconf >= score_thr).to(torch.uint8)
    return _per_class_nms(pconf, pbbox, cand, pconf.shape[1] - 1, max_boxes, iou_thr)


def yolov3_detect(preds, priors_flat, score_thr, max_boxes, iou_thr, decode_scale=(32., 32., 16.)):
    """YOLOv3.py:320-368.  preds: three [H, W, P, C + 5] tensors (head 1 = coarsest); decode_scale as in the reference (sic)."""
    conf, bbox = ops.yolov3_decode_candidates(preds, priors_flat, decode_scale)
    cand = (conf >= score_thr).to(torch.uint8)
    return _per_class_nms(conf, bbox, cand, conf.shape[1], max_boxes, iou_thr)


def yolov2_detect(pred0, priors_flat, score_thr, max_boxes, iou_thr, stride=32.0):
    """YOLOv2.py:177-201.  pred0: [H, W, P, C + 5] of one image."""
    conf, bbox = ops.yolov2_decode_candidates(pred0, priors_flat, stride)
    cand = (conf >= score_thr).to(torch.uint8)
    return _per_class_nms(conf, bbox, cand, conf.shape[1], max_boxes, iou_thr)


class YOLOv2Loss:
    """YOLOv2.py:102-167 for a batch (odtk_yolov2_loss); the gradient of the prediction tensor is in self.d_pred"""

    def __init__(self, N, H, W, P, C, priors_flat, scales, device):
        self.priors_flat, self.scales = priors_flat, scales
        self.loss_parts = torch.zeros(N, 5, device=device)
        self.d_pred = torch.zeros(N, H * W * P, C + 5, device=device)
        self.shape = (N, H, W, P, C + 5)

    def __call__(self, pred, gt, grad_scale, stride=32.0):
        ops.yolov2_loss(pred.view(self.shape), self.priors_flat, stride, gt, self.scales, grad_scale, self.loss_parts, self.d_pred)
        return self.loss_parts


def centernet_detect(keypoints, offset, size, score_thr, top_k, stride=4.0, workspace=None):
    """CenterNet.py:159-185 (no NMS: 3x3 peak test + top-k)."""
    H, W, C = keypoints.shape
    ws = workspace if workspace is not None else ops.centernet_workspace(1, H, W, C, keypoints.device)
    return ops.centernet_decode(keypoints, offset, size, stride, score_thr, top_k, ws)
