"""CPU fp32 restatement of the YOLOv3 box side of the reference (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/YOLOv3.py:
  * priors ............................... _get_priors :419-433, config :37-42 (priors[i] / stride[i], stride = 8,16,32)
  * ground truth in level units .......... _get_normlized_gn :435-442
  * per-image loss ....................... the body of the batch loop, :117-310
  * batch loss ........................... :311-315 (mean over images; the class then adds 0.5 x and the L2 term)
  * inference decode ..................... :320-350 (candidates before the per-class NMS loop :351-368)
Reference behaviour reproduced on purpose (the GPU kernels are drop-ins, not repairs):
  - head 1 (coarsest map, stride 32) is paired with priors[0] = the SMALLEST anchor triple divided by 8, head 3
    (stride 8) with the largest triple divided by 32 (:111-113 with :37-42);
  - a ground-truth box is assigned to head 1 only if its best prior IoU there is STRICTLY larger than on heads 2
    and 3, to head 2 only if strictly larger than 1 and 3, else to head 3 (:186-190);
  - intersections are products of UNCLAMPED side differences (:167-169, :287-289): two negative sides give a
    positive "area";
  - the no-object prior boxes are built from (y1x1, y2x2) as if they were (centre, size): y1x1 - y2x2/2 and
    y1x1 + y2x2/2 (:251-262);
  - a cell is removed from the no-object set if ANY ground-truth centre falls into it, on every head (:126-131);
  - decode: size = prior + exp(t) (a sum, :337-342), heads 1 AND 2 are scaled by 32, head 3 by 16 (:343-348).
Pinned against the reference's own source lines executed on oracle/tf_shim: tests/golden/yolov3_loss.npz
(tests/golden/make_golden_yolov3.py).  Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

STRIDE = [8., 16., 32.]                                                    # YOLOv3.py:38
PRIORS_PX = [[[10., 13.], [16, 30.], [33., 23.]], [[30., 61.], [62., 45.], [59., 119.]],
             [[116., 90.], [156., 198.], [373., 326.]]]                     # testYOLOv3.py:36-38
HEAD_STRIDE = [STRIDE[-1], STRIDE[-2], STRIDE[-3]]                          # gn1 / gn2 / gn3 (:119-121)
NUM_PRIORS = 3


def head_priors(priors_px=PRIORS_PX):
    """[3][num_priors, 2] (h, w) in the units the reference uses for head 1, 2, 3 (= priors[i] / stride[i])."""
    return [torch.tensor(priors_px[i], dtype=torch.float32) / STRIDE[i] for i in range(3)]


def grids(H, W, prior_hw):
    """_get_priors: a_yx [H,W,P,2] = cell + 0.5, a_hw, y1x1, y2x2."""
    ty = torch.arange(0., float(H)).view(H, 1, 1, 1).expand(H, W, 1, 1)
    tx = torch.arange(0., float(W)).view(1, W, 1, 1).expand(H, W, 1, 1)
    yx = (torch.cat([ty, tx], -1) + 0.5).expand(H, W, prior_hw.shape[0], 2)
    hw = prior_hw.view(1, 1, -1, 2).expand(H, W, -1, 2)
    return yx, hw, yx - hw / 2, yx + hw / 2


def _bce(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x z + log(1 + exp(-|x|))."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def one_image_loss(preds, gt, num_classes=20, coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.,
                   priors_px=PRIORS_PX, detail=False):
    """preds: [pred1, pred2, pred3], each [H_l, W_l, P, C + 5] = class(C), yx(2), hw(2), obj(1); gt [pad, 5]."""
    C = num_classes
    pri = head_priors(priors_px)
    G = int(torch.argmin(gt[:, 0]).item())
    g = gt[:G]
    lv = []
    for l in range(3):
        H, W = preds[l].shape[0], preds[l].shape[1]
        a_yx, a_hw, a_y1, a_y2 = grids(H, W, pri[l])
        gn = g / torch.tensor([HEAD_STRIDE[l]] * 4 + [1.])
        yx, hw, lab = gn[:, :2], gn[:, 2:4], gn[:, 4].to(torch.int64)
        cell = torch.floor(yx).long()
        ra_y1, ra_y2, ra_hw = a_y1[cell[:, 0], cell[:, 1]], a_y2[cell[:, 0], cell[:, 1]], a_hw[cell[:, 0], cell[:, 1]]     # [G,P,2]
        g_y1, g_y2 = (yx - hw / 2.).unsqueeze(1), (yx + hw / 2.).unsqueeze(1)                                             # [G,1,2]
        inter = (torch.minimum(g_y2, ra_y2) - torch.maximum(g_y1, ra_y1)).prod(-1)                                        # unclamped
        iou = inter / (ra_hw.prod(-1) + (g_y2 - g_y1).prod(-1) - inter)                                                   # [G,P]
        lv.append(dict(H=H, W=W, a_yx=a_yx, a_hw=a_hw, yx=yx, hw=hw, lab=lab, cell=cell, ra_hw=ra_hw, g_y1=g_y1, g_y2=g_y2,
                       best=iou.argmax(dim=-1), iou_max=iou.max(dim=-1).values))
    m1 = (lv[0]['iou_max'] > lv[1]['iou_max']) & (lv[0]['iou_max'] > lv[2]['iou_max'])
    m2 = (lv[1]['iou_max'] > lv[0]['iou_max']) & (lv[1]['iou_max'] > lv[2]['iou_max'])
    masks = [m1, m2, ~(m1 | m2)]
    coord = torch.zeros(()); cls_l = torch.zeros(()); obj_l = torch.zeros(()); noobj = torch.zeros(())
    for l in range(3):
        d, m, p = lv[l], masks[l], preds[l]
        cell, k = d['cell'][m], d['best'][m]
        r = p[cell[:, 0], cell[:, 1], k]                                          # [Gm, C+5]
        yx_t = d['yx'][m] - torch.floor(d['yx'][m])
        hw_t = torch.log(d['hw'][m] / d['ra_hw'][m][torch.arange(k.shape[0]), k])
        coord = coord + _bce(r[:, C:C + 2], yx_t).sum() + 0.5 * ((r[:, C + 2:C + 4] - hw_t) ** 2).sum()
        cls_l = cls_l + _bce(r[:, :C], F.one_hot(d['lab'][m], C).float()).sum()
        obj_l = obj_l + _bce(r[:, C + 4:], torch.ones_like(r[:, C + 4:])).sum()
        # no-object term: cells without ANY ground-truth centre, every prior
        H, W = d['H'], d['W']
        occupied = torch.zeros(H, W, dtype=torch.bool)
        occupied[d['cell'][:, 0], d['cell'][:, 1]] = True
        free = ~occupied.reshape(-1)
        q1 = (d['a_yx'] - d['a_hw'] / 2.).reshape(H * W, -1, 2)[free].unsqueeze(1)     # "yx_nobest"  [A,1,P,2]
        q2 = (d['a_yx'] + d['a_hw'] / 2.).reshape(H * W, -1, 2)[free].unsqueeze(1)     # "hw_nobest"
        b1, b2 = q1 - q2 / 2., q1 + q2 / 2.
        gy1, gy2 = d['g_y1'].unsqueeze(0), d['g_y2'].unsqueeze(0)                       # [1,G,1,2]
        inter = (torch.minimum(gy2, b2) - torch.maximum(gy1, b1)).prod(-1)              # [A,G,P]
        iou = inter / ((b2 - b1).prod(-1) + (gy2 - gy1).prod(-1) - inter)
        keep = (iou.max(dim=1).values <= 0.5).float()                                   # [A,P]
        o = p[..., C + 4].reshape(H * W, -1)[free]
        noobj = noobj + (_bce(o, torch.zeros_like(o)) * keep).sum()
    Gf = float(G)
    total = (coord_scale * coord + class_scale * cls_l + obj_scale * obj_l) / Gf + noobj_scale * noobj / Gf
    if not detail:
        return total
    return dict(total=total, coord=coord, cls=cls_l, obj=obj_l, noobj=noobj, masks=masks, levels=lv, G=G)


def batch_loss(preds, ground_truth, **kw):
    """preds[l] [N,H,W,P,C+5]; YOLOv3.py:311: mean over images."""
    n = ground_truth.shape[0]
    return torch.stack([one_image_loss([p[i] for p in preds], ground_truth[i], **kw) for i in range(n)]).mean()


def decode_candidates(preds, num_classes=20, priors_px=PRIORS_PX):
    """YOLOv3.py:320-350 for ONE image: confidence [L, C] = sigmoid(class) * sigmoid(obj), bbox [L, 4] y1x1y2x2 px."""
    C = num_classes
    pri = head_priors(priors_px)
    scale = [STRIDE[-1], STRIDE[-1], STRIDE[-2]]
    conf, box = [], []
    for l in range(3):
        p = preds[l]
        H, W = p.shape[0], p.shape[1]
        a_yx, a_hw, _, _ = grids(H, W, pri[l])
        yx = a_yx.reshape(-1, 2) + torch.sigmoid(p[..., C:C + 2].reshape(-1, 2))
        hw = a_hw.reshape(-1, 2) + torch.exp(p[..., C + 2:C + 4].reshape(-1, 2))
        box.append(torch.cat([yx - hw / 2., yx + hw / 2.], -1) * scale[l])
        conf.append(torch.sigmoid(p[..., :C].reshape(-1, C)) * torch.sigmoid(p[..., C + 4:].reshape(-1, 1)))
    return torch.cat(conf, 0), torch.cat(box, 0)


def synthetic_gt(batch, input_size, seed, pad=60, max_obj=6):
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = torch.exp(torch.rand(n, generator=g) * 3.0 + 2.5).clamp(max=input_size * 0.9)
        w = torch.exp(torch.rand(n, generator=g) * 3.0 + 2.5).clamp(max=input_size * 0.9)
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return gt
