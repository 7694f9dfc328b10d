"""GPU parity of the RetinaNet box-side kernels against oracle/retinanet_ref.py, which is
pinned to the reference's own RetinaNet.py functions by tests/golden/retina_*.npz.  Through the C-ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import retinanet_ref as RR  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _ops():
    import odtk  # noqa: F401
    from odtk import ops
    return ops


def _gpu_anchors(ops, dev, data_shape):
    shapes = RR.pyramid_shapes(data_shape[0], data_shape[1])
    flat = []
    for s in RR.ANCHOR_SIZES:
        for h, w in RR.level_priors(s):
            flat += [h, w]
    return ops.retina_anchors(data_shape[1], shapes, [RR.NUM_ANCHORS] * 5, flat, dev), shapes


@pytest.mark.parametrize("data_shape", [[320, 256, 3], [500, 500, 3], [800, 800, 3]])
def test_retina_anchors_bit_exact(data_shape, dev):
    ops = _ops()
    got, shapes = _gpu_anchors(ops, dev, data_shape)
    ref = RR.anchors(data_shape, shapes)
    torch.cuda.synchronize()
    for g, r in zip(got, ref):
        assert torch.equal(g.cpu(), r)
    if data_shape[0] == 800:
        assert got[0].shape[0] == 120087


def _match_gpu(ops, dev, anc, gt):
    N, P, _ = gt.shape
    A = anc[0].shape[0]
    i32 = dict(dtype=torch.int32, device=dev)
    ngt = torch.zeros(N, **i32); best = torch.full((N, P), -1, **i32)
    status = torch.zeros(N, A, dtype=torch.uint8, device=dev); rg = torch.zeros(N, A, **i32)
    counts = torch.zeros(N, 4, **i32)
    ws = ops.retina_match_workspace(A, N, P, dev)
    ops.retina_match(anc[0], anc[1], anc[3], gt.to(dev), ngt, best, status, rg, counts, ws)
    return ngt, best, status, rg, counts


@pytest.mark.parametrize("data_shape,seed", [([320, 256, 3], 1), ([500, 500, 3], 2), ([800, 800, 3], 3)])
def test_retina_match_indices_bit_exact(data_shape, seed, dev):
    ops = _ops()
    anc_d, shapes = _gpu_anchors(ops, dev, data_shape)
    anc = RR.anchors(data_shape, shapes)
    N = 3
    gt = RR.synthetic_gt(N, min(data_shape[:2]), seed)
    gt[1, 1] = gt[1, 0]; gt[1, 2:] = -1                       # duplicate GT -> duplicate best anchors
    ngt, best, status, rg, counts = _match_gpu(ops, dev, anc_d, gt)
    torch.cuda.synchronize()
    for i in range(N):
        mt = RR.match(anc, gt[i])
        G = mt["G"]
        assert int(ngt[i]) == G
        assert best[i, :G].cpu().tolist() == mt["best"].tolist()
        st = torch.full((anc[0].shape[0],), 3, dtype=torch.uint8)
        other = torch.nonzero(mt["othermask"]).squeeze(1)
        st[other] = torch.where(mt["pos"], torch.tensor(1, dtype=torch.uint8),
                                torch.where(mt["neg"], torch.tensor(2, dtype=torch.uint8), torch.tensor(0, dtype=torch.uint8)))
        assert torch.equal(status[i].cpu(), st)
        assert torch.equal(rg[i].cpu().long()[other], mt["rgindex"])
        assert counts[i].cpu().tolist()[:2] == [G + int(mt["pos"].sum()), int(mt["neg"].sum())]


def _loss_gpu(ops, dev, anc_d, pconf, pbox, gt, alpha=0.25, gamma=2.0):
    N, A, C = pconf.shape
    ngt, best, status, rg, counts = _match_gpu(ops, dev, anc_d, gt)
    parts = torch.empty(N, 2, device=dev)
    dconf = torch.full((N, A, C), 7.0, device=dev); dbox = torch.full((N, A, 4), 7.0, device=dev)
    ops.retina_loss(pconf.to(dev), pbox.to(dev), anc_d[2], anc_d[3], gt.to(dev), ngt, best, status, rg, counts, alpha, gamma,
                    1.0 / N, parts, dconf, dbox)
    torch.cuda.synchronize()
    return parts.cpu(), dconf.cpu(), dbox.cpu()


def test_retina_loss_vs_reference_golden(dev):
    """Same inputs as the fixture produced by the reference's _compute_one_image_loss on the shim."""
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'retina_loss.npz'))
    anc_d, _ = _gpu_anchors(ops, dev, [320, 256, 3])
    pconf = torch.from_numpy(g['pconf'].astype(np.float32)); pbox = torch.from_numpy(g['pbox'].astype(np.float32))
    gt = torch.from_numpy(g['gt'])
    parts, _, _ = _loss_gpu(ops, dev, anc_d, pconf, pbox, gt)
    for i in range(pconf.shape[0]):
        got = float(parts[i].sum())
        assert abs(got - float(g['loss'][i])) <= 2e-5 * abs(float(g['loss'][i])), (i, got, float(g['loss'][i]))


@pytest.mark.parametrize("data_shape,gamma", [([320, 256, 3], 2.0), ([500, 500, 3], 1.5)])
def test_retina_loss_and_grad_vs_oracle(data_shape, gamma, dev):
    ops = _ops()
    anc_d, shapes = _gpu_anchors(ops, dev, data_shape)
    anc = RR.anchors(data_shape, shapes)
    A = anc[0].shape[0]
    N = 2
    g = torch.Generator().manual_seed(A)
    pconf = torch.randn(N, A, 21, generator=g) * 2
    pbox = torch.randn(N, A, 4, generator=g) * 0.7
    gt = RR.synthetic_gt(N, min(data_shape[:2]), 5)
    parts, dconf, dbox = _loss_gpu(ops, dev, anc_d, pconf, pbox, gt, 0.25, gamma)
    pc = pconf.clone().requires_grad_(True); pb = pbox.clone().requires_grad_(True)
    tot = 0
    for i in range(N):
        d = RR.one_image_loss(pb[i, :, :2], pb[i, :, 2:], pc[i], anc, gt[i], 0.25, gamma, detail=True)
        tot = tot + d["total"]
        assert abs(float(parts[i, 0]) - float(d["conf_loss"])) <= 2e-5 * abs(float(d["conf_loss"])) + 1e-6
        assert abs(float(parts[i, 1]) - float(d["coord"])) <= 2e-5 * abs(float(d["coord"])) + 1e-6
    (tot / N).backward()
    assert float((dconf - pc.grad).abs().max()) <= 1e-6 + 1e-4 * float(pc.grad.abs().max())
    assert float((dbox - pb.grad).abs().max()) <= 1e-6 + 1e-4 * float(pb.grad.abs().max())


def test_retina_decode_candidates(dev):
    """RetinaNet.py:224-238 (same arithmetic as SSD300.py:157-171, pinned there) on the 320x256 anchor set."""
    ops = _ops()
    got, shapes = _gpu_anchors(ops, dev, [320, 256, 3])
    anc = RR.anchors([320, 256, 3], shapes)
    A = anc[0].shape[0]
    g = torch.Generator().manual_seed(4)
    pconf = torch.randn(A, 21, generator=g) * 2
    pbox = torch.randn(A, 4, generator=g) * 0.5
    conf, boxes, keep, cand = ops.retina_decode(pconf.to(dev), pbox.to(dev), got[2], got[3], 0.3)
    rc, rb, rk, rcand = RR.decode_candidates(pbox[:, :2], pbox[:, 2:], pconf, anc, 0.3)
    assert float((conf.cpu() - rc).abs().max()) <= 1e-6
    assert float((boxes.cpu() - rb).abs().max()) <= 1e-3
    near = ((rc - 0.3).abs() < 1e-6).any(dim=1) | ((rc.max(dim=1).values - torch.softmax(pconf, -1)[:, 20]).abs() < 1e-6)
    assert torch.equal(keep.cpu().bool()[~near], rk[~near])
    assert torch.equal(cand.cpu().bool()[~near], rcand[~near])
