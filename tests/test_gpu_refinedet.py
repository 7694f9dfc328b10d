"""GPU parity of the RefineDet box side against oracle/refinedet_ref.py, which is pinned to the reference's own RefineDet.py
functions by tests/golden/refinedet.npz: anchors bit for bit, matching indices bit for bit, the two-stage loss against the reference's own numbers,
all four gradients against autograd of the oracle (incl. the ODM box term's path into the ARM outputs), the inference tail against the
reference's own detections.  Through the C-ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import refinedet_ref as FR  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _heads():
    import odtk  # noqa: F401
    from odtk import heads
    return heads


@pytest.mark.parametrize("size", [320, 512])
def test_anchors_bit_exact(size, dev):
    got = _heads().refinedet_anchors(size, dev)
    ref = FR.anchors(size)
    torch.cuda.synchronize()
    for g, r in zip(got[:4], ref):
        assert torch.equal(g.cpu(), r)
    if size == 320:
        gold = np.load(os.path.join(GOLD, 'refinedet.npz'))
        assert got[0].shape[0] == 6375 and np.array_equal(got[2].cpu().numpy(), gold['yx'])       # the reference's own _get_abbox


def _inputs(dev):
    g = np.load(os.path.join(GOLD, 'refinedet.npz'))
    f = lambda k: torch.from_numpy(g[k].astype(np.float32))
    return g, f('arm_loc'), f('arm_conf'), f('odm_loc'), f('odm_conf'), torch.from_numpy(g['gt'])


def test_loss_matches_reference_numbers_and_oracle_gradients(dev):
    heads = _heads()
    g, arm_loc, arm_conf, odm_loc, odm_conf, gt = _inputs(dev)
    anc_d = heads.refinedet_anchors(320, dev)
    anc = FR.anchors(320)
    N, A = arm_loc.shape[:2]
    L = heads.RefineDetLoss(anc_d, N, 21, gt.shape[1], dev)
    parts = L(arm_loc.to(dev), arm_conf.to(dev), odm_loc.to(dev), odm_conf.to(dev), gt.to(dev), 1.0 / N)
    torch.cuda.synchronize()
    tot = parts[:, 6].cpu().numpy()
    assert np.allclose(tot, g['loss'], rtol=2e-5), (tot, g['loss'])                          # the reference's own _compute_one_image_loss
    # matching + mined negatives, bit for bit
    leaves = [t.clone().requires_grad_(True) for t in (arm_loc, arm_conf, odm_loc, odm_conf)]
    total = 0.
    for i in range(N):
        d = FR.one_image_loss(leaves[0][i, :, :2], leaves[0][i, :, 2:], leaves[1][i], leaves[2][i, :, :2], leaves[2][i, :, 2:], leaves[3][i], anc, gt[i],
                              detail=True)
        mt = d['match']
        assert int(L.ngt[i]) == mt['G'] and L.best[i, : mt['G']].cpu().tolist() == mt['best'].tolist()
        assert torch.equal(L.status[i].cpu(), mt['status'])
        other = mt['status'] != 3
        assert torch.equal(L.rg[i].cpu().long()[other], mt['rgindex'][other])
        assert L.counts[i].cpu().tolist()[:3] == [d['num_pos'], d['num_neg'], min(3 * d['num_pos'], d['num_neg'])]
        assert L.sel_idx[i, : int(L.sel_cnt[i])].cpu().tolist() == d['sel_rows'].tolist()
        assert int(round(float(parts[i, 7]))) == int(d['odm_neg_rows'].shape[0])
        for j, want in enumerate(d['parts']):
            assert abs(float(parts[i, j]) - float(want)) <= 2e-5 * abs(float(want)) + 1e-6, (i, j)
        total = total + d['total']
    (total / N).backward()
    for got, leaf, tag in zip((L.d_arm_loc, L.d_arm_conf, L.d_odm_loc, L.d_odm_conf), leaves, ('arm_loc', 'arm_conf', 'odm_loc', 'odm_conf')):
        err = float((got.cpu() - leaf.grad).abs().max()) / (float(leaf.grad.abs().max()) + 1e-12)
        assert err < 1e-4, (tag, err)
    assert float(L.d_arm_loc.abs().sum()) > 0


def test_inference_tail_matches_reference_detections(dev):
    heads = _heads()
    g, arm_loc, arm_conf, odm_loc, odm_conf, _ = _inputs(dev)
    anc_d = heads.refinedet_anchors(320, dev)
    s, b, c = heads.refinedet_detect(arm_loc[0].to(dev).contiguous(), arm_conf[0].to(dev).contiguous(), odm_loc[0].to(dev).contiguous(),
                                     odm_conf[0].to(dev).contiguous(), anc_d[2], anc_d[3], 0.12, 10, 0.45)
    torch.cuda.synchronize()
    assert c.cpu().tolist() == g['det_class'].tolist() and len(s) == 200
    assert np.allclose(s.cpu().numpy(), g['det_scores'], atol=1e-6)
    assert np.allclose(b.cpu().numpy(), g['det_bbox'], rtol=1e-4, atol=2e-3)
    # decode of every anchor against the oracle
    import odtk  # noqa: F401
    from odtk import ops
    conf, boxes, keep, cand = ops.refinedet_decode(arm_loc[0].to(dev).contiguous(), arm_conf[0].to(dev).contiguous(), odm_loc[0].to(dev).contiguous(),
                                                   odm_conf[0].to(dev).contiguous(), anc_d[2], anc_d[3], 0.12)
    cr, br, kr = FR.decode(arm_loc[0], arm_conf[0], odm_loc[0], odm_conf[0], FR.anchors(320))
    assert torch.equal(keep.cpu().bool(), kr) and float((conf.cpu() - cr).abs().max()) < 1e-6
    assert float(((boxes.cpu() - br).abs() / (br.abs() + 1.0)).max()) < 1e-4
