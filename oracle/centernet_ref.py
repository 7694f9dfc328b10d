"""CPU fp32 restatement of the CenterNet box side of the reference (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/CenterNet.py:
  * grids ................................ tf.meshgrid(range(W), range(H))        CenterNet.py:136-138
  * per-image loss ....................... _compute_one_image_loss               CenterNet.py:187-209
  * penalty-reduced focal keypoint loss .. _keypoints_loss                       CenterNet.py:211-251
  * gaussian radius ...................... _gaussian_radius                      CenterNet.py:254-270
      (reference quirk, reproduced: tf.reduce_min([r1, r2, r3]) packs the three [G] vectors and reduces over
       EVERYTHING, so one scalar sigma = the smallest radius of any ground-truth box serves the whole image)
  * batch loss ........................... mean over images                      CenterNet.py:144-152
  * inference decode ..................... CenterNet.py:158-185 (sigmoid, arg-max class, 3x3 peak test on the
      class-reduced score map, score > threshold, top-k with lower index first on ties)
Pinned against the reference's own code run on oracle/tf_shim: tests/golden/centernet_*.npz
(tests/golden/make_golden_centernet_fcos.py executes _compute_one_image_loss and, for the decode, the
reference's inline source lines read at generation time).  TF kernel semantics underneath the shim are
"parity unpinned" exactly as for SSD300.  Only tests/ and the smoke/bench checkers may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

STRIDE = 4.0                                                   # CenterNet.py:126


def gaussian_radius(h, w, min_overlap=0.7):
    """CenterNet.py:254-270 -- ONE scalar (min over the three roots and over all boxes)."""
    a1 = 1.
    b1 = h + w
    c1 = w * h * (1. - min_overlap) / (1. + min_overlap)
    r1 = (b1 + torch.sqrt(b1 ** 2. - 4. * a1 * c1)) / 2.
    a2 = 4.
    b2 = 2. * (h + w)
    c2 = (1. - min_overlap) * w * h
    r2 = (b2 + torch.sqrt(b2 ** 2. - 4. * a2 * c2)) / 2.
    a3 = 4. * min_overlap
    b3 = -2. * min_overlap * (h + w)
    c3 = (min_overlap - 1.) * w * h
    r3 = (b3 + torch.sqrt(b3 ** 2. - 4. * a3 * c3)) / 2.
    return torch.stack([r1, r2, r3]).min()


def targets(gt, H, W, C, stride=STRIDE):
    """reduction [H,W,C] (per-class max of the gaussian penalties), gt_keypoints [H,W,C] (1 at floor(centre)),
    cell indices [G,2], offset_gt [G,2], size_gt [G,2]   (CenterNet.py:189-199, :213-245)."""
    G = int(torch.argmin(gt[:, 0]).item())
    g = gt[:G]
    yx = g[:, 0:2] / stride
    fl = torch.floor(yx)
    idx = fl.long()
    cls = g[:, 4].to(torch.int32)
    sigma = gaussian_radius(g[:, 2] / stride, g[:, 3] / stride, 0.7)
    my = torch.arange(0., float(H)).view(1, H, 1)
    mx = torch.arange(0., float(W)).view(1, 1, W)
    gy = (g[:, 0] / stride).view(-1, 1, 1)
    gx = (g[:, 1] / stride).view(-1, 1, 1)
    pen = torch.exp(-((gy - my) ** 2 + (gx - mx) ** 2) / (2 * sigma ** 2))            # [G,H,W]
    red = torch.zeros(H, W, C)
    kp = torch.zeros(H, W, C)
    for c in range(C):
        m = cls == c
        if bool(m.any()):
            red[..., c] = pen[m].max(dim=0).values
            kp[idx[m, 0], idx[m, 1], c] = 1.
    return dict(G=G, idx=idx, offset_gt=yx - fl, size_gt=g[:, 2:4] / stride, reduction=red, gt_keypoints=kp, sigma=sigma)


def one_image_loss(keypoints, offset, size, gt, stride=STRIDE, detail=False):
    """CenterNet.py:187-251.  keypoints [H,W,C] logits, offset / size [H,W,2], gt [P,5] = yc,xc,h,w,cls (pad -1)."""
    H, W, C = keypoints.shape
    t = targets(gt, H, W, C, stride)
    G = float(t['G'])
    s = torch.sigmoid(keypoints)
    ls = F.logsigmoid(keypoints)
    pos = -torch.pow(1. - s, 2.) * ls * t['gt_keypoints']
    neg = -torch.pow(1. - t['reduction'], 4) * torch.pow(s, 2.) * (-keypoints + ls) * (1. - t['gt_keypoints'])
    kl = pos.sum() / G + neg.sum() / G
    o = offset[t['idx'][:, 0], t['idx'][:, 1]]
    z = size[t['idx'][:, 0], t['idx'][:, 1]]
    off_l = (t['offset_gt'] - o).abs().mean()
    size_l = (t['size_gt'] - z).abs().mean()
    total = kl + 0.1 * size_l + off_l
    if not detail:
        return total
    return dict(total=total, keypoints_loss=kl, offset_loss=off_l, size_loss=size_l, **t)


def batch_loss(keypoints, offset, size, ground_truth, stride=STRIDE):
    """CenterNet.py:144-152: mean of the per-image losses."""
    n = keypoints.shape[0]
    return torch.stack([one_image_loss(keypoints[i], offset[i], size[i], ground_truth[i], stride) for i in range(n)]).mean()


def decode(keypoints, offset, size, score_threshold, top_k, stride=STRIDE):
    """CenterNet.py:158-185 for ONE image: keypoints [H,W,C] logits -> scores [K], bbox [K,4] y1x1y2x2 px, class [K]."""
    H, W, C = keypoints.shape
    kp = torch.sigmoid(keypoints)
    cat = torch.zeros(H, W, dtype=torch.int64)
    best = kp[..., 0].clone()
    for c in range(1, C):                                   # tf.argmax: first maximum
        m = kp[..., c] > best
        cat[m] = c
        best = torch.where(m, kp[..., c], best)
    peak = F.max_pool2d(best.view(1, 1, H, W), 3, 1, 1).view(H, W)     # 'same' padding pads with -inf
    sc = (best * (best == peak).float()).reshape(-1)
    cy = torch.arange(0., float(H)).view(H, 1).expand(H, W)
    cx = torch.arange(0., float(W)).view(1, W).expand(H, W)
    centre = torch.stack([cy, cx], -1)
    yx = (centre + offset).reshape(-1, 2)
    hw = size.reshape(-1, 2)
    keep = sc > score_threshold
    sc_k, cls_k = sc[keep], cat.reshape(-1)[keep]
    box = torch.cat([yx[keep] - hw[keep] / 2., yx[keep] + hw[keep] / 2.], -1) * stride
    k = min(int(top_k), int(sc_k.shape[0]))
    order = torch.sort(sc_k, descending=True, stable=True).indices[:k]     # tf.nn.top_k: lower index first on ties
    return sc_k[order], box[order], cls_k[order].to(torch.int32)


def synthetic_gt(batch, input_size, seed, pad=60, max_obj=6):
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = torch.rand(n, generator=g) * (input_size * 0.6) + input_size * 0.08
        w = torch.rand(n, generator=g) * (input_size * 0.6) + input_size * 0.08
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return gt
