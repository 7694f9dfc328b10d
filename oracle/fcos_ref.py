"""CPU fp32 restatement of the FCOS box side of the reference (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/FCOS.py:
  * level grids / strides ................ s = 8,16,32,64,128; grid = integer cell indices   FCOS.py:134-150
  * ground truth -> pyramid level ........ sqrt(h*w) in [0,64] [64,128] [128,256] [256,512] [512,inf), both ends
                                           inclusive (a box of size exactly 64 trains levels 3 AND 4)    FCOS.py:154-164
  * per-level loss ....................... _compute_one_image_loss                            FCOS.py:266-348
      quirks reproduced: the box of minimum area wins a location, ties keep the per-side MAXIMUM of the tied
      boxes (:293-305); the centre-ness BCE is summed over ALL locations (:324-326); everything is divided by
      sum(heatmap_gt) (:346), which is 0/0 -> nan when a level's boxes cover no grid point
  * batch loss ........................... sum over levels with >= 1 box, mean over images    FCOS.py:165-187
  * inference decode ..................... sigmoid(conf) * sigmoid(centre-ness), boxes = (grid -/+ reg) * stride,
                                           order y1,x1,y2,x2 from reg = l,r,t,b; classes 0..C-2          FCOS.py:192-265
Pinned against the reference's own code run on oracle/tf_shim: tests/golden/fcos_*.npz
(tests/golden/make_golden_centernet_fcos.py).  Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

STRIDES = (8, 16, 32, 64, 128)                                # FCOS.py:134


def level_shapes(input_h, input_w):
    """p3..p7 feature-map sizes: ResNet stem s2 + pool s2 (SAME -> ceil), three stride-2 stages, two stride-2 convs."""
    c = lambda v: -(-v // 2)
    h, w = c(c(input_h)), c(c(input_w))
    out = []
    for _ in range(5):
        h, w = c(h), c(w)
        out.append((h, w))
    return out


def level_mask(gt_valid):
    """[5, G] bool: which ground-truth rows train which level (FCOS.py:158-163)."""
    sz = torch.sqrt(gt_valid[:, 2] * gt_valid[:, 3])
    return torch.stack([sz <= 64., (sz >= 64.) & (sz <= 128.), (sz >= 128.) & (sz <= 256.), (sz >= 256.) & (sz <= 512.), sz >= 512.])


def level_targets(g, H, W, C, stride):
    """FCOS.py:268-305, :329-342 for the boxes g [G,5] of one level."""
    gy, gx, gh, gw = (g[:, k] / stride for k in range(4))
    cls = g[:, 4].to(torch.int32)
    y1, y2, x1, x2 = gy - gh / 2., gy + gh / 2., gx - gw / 2., gx + gw / 2.
    grid_y = torch.arange(0., float(H)).view(H, 1, 1)
    grid_x = torch.arange(0., float(W)).view(1, W, 1)
    dl = (grid_x - x1.view(1, 1, -1)).expand(H, W, -1)
    dr = (x2.view(1, 1, -1) - grid_x).expand(H, W, -1)
    dt = (grid_y - y1.view(1, 1, -1)).expand(H, W, -1)
    db = (y2.view(1, 1, -1) - grid_y).expand(H, W, -1)
    heat = ((dt > 0.) & (db > 0.)).float() * ((dl > 0.) & (dr > 0.)).float()
    dl, dr, dt, db = dl * heat, dr * heat, dt * heat, db * heat
    loc = heat.max(dim=-1).values
    area = (dl + dr) * (dt + db)
    area_min = (area + (1. - heat) * 1e8).min(dim=-1, keepdim=True).values
    dmask = (area == area_min).float() * loc.unsqueeze(-1)
    dl, dr, dt, db = ((d * dmask).max(dim=-1).values for d in (dl, dr, dt, db))
    hm = torch.zeros(H, W, C)
    for c in range(C):
        m = cls == c
        if bool(m.any()):
            hm[..., c] = heat[..., m].max(dim=-1).values
    return dict(loc=loc, dl=dl, dr=dr, dt=dt, db=db, heatmap_gt=hm)


def level_loss(heatmap_pred, dist_pred, center_pred, g, stride, detail=False):
    """FCOS.py:266-348.  heatmap_pred [H,W,C] logits, dist_pred [H,W,4] = l,r,t,b (> 0), center_pred [H,W,1] logits."""
    H, W, C = heatmap_pred.shape
    t = level_targets(g, H, W, C, stride)
    dl, dr, dt, db, loc = t['dl'], t['dr'], t['dt'], t['db'], t['loc']
    pl, pr, pt, pb = (dist_pred[..., k] for k in range(4))
    iw = torch.minimum(dl, pl) + torch.minimum(dr, pr)
    ih = torch.minimum(dt, pt) + torch.minimum(db, pb)
    inter = iw * ih
    union = (dl + dr) * (dt + db) + (pl + pr) * (pt + pb) - inter
    iou = inter / (union + 1e-12)
    iou_loss = (-torch.log(iou + 1e-12) * loc).sum()
    lr_min, tb_min = torch.minimum(dl, dr), torch.minimum(dt, db)
    lr_max, tb_max = torch.maximum(dl, dr), torch.maximum(dt, db)
    cgt = torch.sqrt(lr_min * tb_min / (lr_max * tb_max + 1e-12))
    x = center_pred.reshape(H, W)
    center_loss = (torch.clamp(x, min=0) - x * cgt + torch.log1p(torch.exp(-x.abs()))).sum()
    s = torch.sigmoid(heatmap_pred)
    ls = F.logsigmoid(heatmap_pred)
    hg = t['heatmap_gt']
    pos = -.25 * torch.pow(1. - s, 2.) * ls * hg
    neg = -.25 * torch.pow(s, 2.) * (-heatmap_pred + ls) * (1. - hg)
    heat_loss = pos.sum() + neg.sum()
    total = (iou_loss + heat_loss + center_loss) / hg.sum()
    if not detail:
        return total
    return dict(total=total, iou_loss=iou_loss, heatmap_loss=heat_loss, center_loss=center_loss, center_gt=cgt, **t)


def one_image_loss(conf, reg, center, gt):
    """conf / reg / center: lists of 5 per-level tensors [H,W,C] / [H,W,4] / [H,W,1] of ONE image; gt [P,5]."""
    G = int(torch.argmin(gt[:, 0]).item())
    g = gt[:G]
    lm = level_mask(g)
    total = torch.zeros(())
    for l in range(5):
        if bool(lm[l].any()):
            total = total + level_loss(conf[l], reg[l], center[l], g[lm[l]], float(STRIDES[l]))
    return total


def batch_loss(conf, reg, center, ground_truth):
    """conf[l] [N,H,W,C] ...; FCOS.py:186: mean over images."""
    n = ground_truth.shape[0]
    return torch.stack([one_image_loss([c[i] for c in conf], [r[i] for r in reg], [c[i] for c in center], ground_truth[i])
                        for i in range(n)]).mean()


def decode_candidates(conf, reg, center):
    """FCOS.py:192-246 for ONE image: pconf [L, C] = sigmoid(conf) * sigmoid(centre-ness), pbbox [L, 4] y1,x1,y2,x2 px
    (levels concatenated p3..p7, row-major inside a level)."""
    pc, pb = [], []
    for l in range(5):
        H, W, C = conf[l].shape
        pc.append((torch.sigmoid(conf[l]) * torch.sigmoid(center[l])).reshape(-1, C))
        gy = torch.arange(0., float(H)).view(H, 1, 1).expand(H, W, 1)
        gx = torch.arange(0., float(W)).view(1, W, 1).expand(H, W, 1)
        r = reg[l]
        box = torch.cat([gy - r[..., 2:3], gx - r[..., 0:1], gy + r[..., 3:4], gx + r[..., 1:2]], -1).reshape(-1, 4) * STRIDES[l]
        pb.append(box)
    return torch.cat(pc, 0), torch.cat(pb, 0)


def synthetic_gt(batch, input_size, seed, pad=60, max_obj=6):
    """boxes spread over the FCOS size bands (sqrt(h*w) from ~20 to ~input_size)."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        sz = torch.exp(torch.rand(n, generator=g) * (torch.log(torch.tensor(input_size * 0.9)) - 3.0) + 3.0)
        ar = torch.exp((torch.rand(n, generator=g) - 0.5) * 1.2)
        h = torch.clamp(sz * ar.sqrt(), max=input_size * 0.95)
        w = torch.clamp(sz / ar.sqrt(), max=input_size * 0.95)
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return gt
