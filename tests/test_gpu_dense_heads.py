"""GPU parity of the CenterNet / FCOS box-side kernels against oracle/centernet_ref.py
and oracle/fcos_ref.py, which are pinned to the reference's own code by tests/golden/{centernet,fcos}_loss.npz.
Gradients are checked against autograd of the oracle.  Through the C-ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import centernet_ref as CR  # noqa: E402
from oracle import fcos_ref as FR       # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _ops():
    import odtk  # noqa: F401
    from odtk import ops
    return ops


def _rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


# ------------------------------------------------------------------------------------------------ CenterNet
def _centernet_case(kp, off, size, gt, dev):
    ops = _ops()
    N, H, W, C = kp.shape
    kd, od, zd, gd = (t.to(dev).contiguous() for t in (kp, off, size, gt))
    parts = torch.zeros(N, 4, device=dev)
    dk, do, dz = torch.empty_like(kd), torch.empty_like(od), torch.empty_like(zd)
    ws = ops.centernet_workspace(N, H, W, C, dev)
    ops.centernet_loss(kd, od, zd, gd, CR.STRIDE, 1.0 / N, parts, dk, do, dz, ws)
    torch.cuda.synchronize()
    kr, orr, zr = (t.clone().requires_grad_(True) for t in (kp, off, size))
    dets = [CR.one_image_loss(kr[i], orr[i], zr[i], gt[i], detail=True) for i in range(N)]
    total = torch.stack([d['total'] for d in dets]).mean()
    total.backward()
    got = parts.cpu()
    for i, d in enumerate(dets):
        for j, key in enumerate(('keypoints_loss', 'offset_loss', 'size_loss', 'total')):
            r = float(d[key].detach())
            assert abs(float(got[i, j]) - r) <= 2e-5 * abs(r) + 1e-6, (i, key, float(got[i, j]), r)
    assert _rel(dk.cpu(), kr.grad) <= 2e-4
    assert _rel(do.cpu(), orr.grad) <= 1e-5 and _rel(dz.cpu(), zr.grad) <= 1e-5
    return got


def test_centernet_loss_golden_inputs(dev):
    g = np.load(os.path.join(GOLD, 'centernet_loss.npz'))
    kp, off, size = (torch.from_numpy(g[k].astype(np.float32)) for k in ('keypoints', 'offset', 'size'))
    got = _centernet_case(kp, off, size, torch.from_numpy(g['gt']), dev)
    for i in range(kp.shape[0]):                                      # and directly against the reference's numbers
        assert abs(float(got[i, 3]) - float(g['loss'][i])) <= 3e-5 * abs(float(g['loss'][i]))


def test_centernet_loss_config5_shape(dev):
    """BASELINE config 5 geometry: 512 x 512 input -> 128 x 128 x 20 heat map."""
    g = torch.Generator().manual_seed(3)
    N, H, W, C = 4, 128, 128, 20
    kp = torch.randn(N, H, W, C, generator=g) * 1.5 - 2.0
    off = torch.rand(N, H, W, 2, generator=g)
    size = torch.rand(N, H, W, 2, generator=g) * 40
    _centernet_case(kp, off, size, CR.synthetic_gt(N, 512, 77), dev)


@pytest.mark.parametrize("img,shift,thr,topk", [(0, 0.0, 0.1, 100), (1, 3.0, 0.1, 100), (0, 0.0, 0.6, 100), (2, -6.0, 0.3, 100), (2, 3.0, 0.05, 7)])
def test_centernet_decode(img, shift, thr, topk, dev):
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'centernet_loss.npz'))
    kp = torch.from_numpy(g['keypoints'].astype(np.float32))[img] + shift
    off = torch.from_numpy(g['offset'].astype(np.float32))[img]
    size = torch.from_numpy(g['size'].astype(np.float32))[img]
    H, W, C = kp.shape
    ws = ops.centernet_workspace(1, H, W, C, dev)
    s, b, c = ops.centernet_decode(kp.to(dev), off.to(dev), size.to(dev), CR.STRIDE, thr, topk, ws)
    rs, rb, rc = CR.decode(kp, off, size, thr, topk)
    assert s.shape[0] == rs.shape[0]
    assert torch.equal(c.cpu(), rc)                                   # same cells in the same order
    assert float((s.cpu() - rs).abs().max()) <= 1e-6 if rs.numel() else True
    assert float((b.cpu() - rb).abs().max()) <= 1e-4 if rs.numel() else True
    if thr == 0.1 and topk == 100:                                    # the reference's own outputs
        i = img
        assert np.array_equal(c.cpu().numpy(), g[f'det{i}_class_id'])
        assert np.abs(b.cpu().numpy() - g[f'det{i}_bbox']).max() <= 1e-4


# ------------------------------------------------------------------------------------------------ FCOS
def _fcos_case(conf, reg, cen, gt, dev):
    ops = _ops()
    N = gt.shape[0]
    cd, rd, zd = ([t.to(dev).contiguous() for t in lst] for lst in (conf, reg, cen))
    loss = torch.zeros(N, device=dev)
    dc, dr, dz = ([torch.full_like(t, 7.0) for t in lst] for lst in (cd, rd, zd))
    ws = ops.fcos_workspace(cd, N, dev)
    ops.fcos_loss(cd, rd, zd, gt.to(dev), 1.0 / N, loss, dc, dr, dz, ws)
    torch.cuda.synchronize()
    cr, rr, zr = ([t.clone().requires_grad_(True) for t in lst] for lst in (conf, reg, cen))
    per = [FR.one_image_loss([c[i] for c in cr], [r[i] for r in rr], [c[i] for c in zr], gt[i]) for i in range(N)]
    torch.stack(per).mean().backward()
    for i in range(N):
        r = float(per[i].detach())
        assert abs(float(loss[i]) - r) <= 3e-5 * abs(r) + 1e-6, (i, float(loss[i]), r)
    for l in range(5):
        for got, ref, name in ((dc, cr, 'conf'), (dr, rr, 'reg'), (dz, zr, 'center')):
            rg = ref[l].grad if ref[l].grad is not None else torch.zeros_like(ref[l])
            scale = max(float(r_.grad.abs().max()) if r_.grad is not None else 0.0 for r_ in ref) + 1e-12
            assert float((got[l].cpu() - rg).abs().max()) <= 3e-4 * scale, (l, name)
    return loss.cpu()


def test_fcos_loss_golden_inputs(dev):
    g = np.load(os.path.join(GOLD, 'fcos_loss.npz'))
    conf = [torch.from_numpy(g[f'conf{l}'].astype(np.float32)) for l in range(5)]
    reg = [torch.from_numpy(g[f'reg{l}'].astype(np.float32)) for l in range(5)]
    cen = [torch.from_numpy(g[f'center{l}'].astype(np.float32)) for l in range(5)]
    got = _fcos_case(conf, reg, cen, torch.from_numpy(g['gt']), dev)
    for i in range(got.shape[0]):                                     # and directly against the reference's numbers
        assert abs(float(got[i]) - float(g['loss'][i])) <= 5e-5 * abs(float(g['loss'][i]))


def test_fcos_loss_config5_shape(dev):
    """BASELINE config 5 geometry: 512 x 512 input -> 64/32/16/8/4 maps (5 456 locations), 20 + 1 classes."""
    shapes = FR.level_shapes(512, 512)
    assert sum(h * w for h, w in shapes) == 5456
    g = torch.Generator().manual_seed(8)
    N = 4
    conf = [torch.randn(N, h, w, 21, generator=g) * 1.5 - 2.0 for h, w in shapes]
    reg = [torch.exp(torch.randn(N, h, w, 4, generator=g)) * 2 for h, w in shapes]
    cen = [torch.randn(N, h, w, 1, generator=g) for h, w in shapes]
    _fcos_case(conf, reg, cen, FR.synthetic_gt(N, 512, 5), dev)


def test_fcos_decode_candidates(dev):
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'fcos_loss.npz'))
    conf = [torch.from_numpy(g[f'conf{l}'].astype(np.float32))[0] for l in range(5)]
    reg = [torch.from_numpy(g[f'reg{l}'].astype(np.float32))[0] for l in range(5)]
    cen = [torch.from_numpy(g[f'center{l}'].astype(np.float32))[0] for l in range(5)]
    pc, pb = ops.fcos_decode_candidates([t.to(dev).contiguous() for t in conf], [t.to(dev).contiguous() for t in reg],
                                        [t.to(dev).contiguous() for t in cen])
    rc, rb = FR.decode_candidates(conf, reg, cen)
    assert float((pc.cpu() - rc).abs().max()) <= 1e-6
    assert torch.equal(pb.cpu(), rb)                                  # pure add / multiply: bit-exact
    assert np.abs(pc.cpu().numpy()[::3] - g['pconf']).max() <= 1e-6 and np.array_equal(pb.cpu().numpy()[::3], g['pbbox'])


# ------------------------------------------------------------------------------------------------ YOLOv3
from oracle import yolov3_ref as YR  # noqa: E402


def _yolo_priors_flat():
    return [float(v) for l in YR.head_priors() for v in l.reshape(-1).tolist()]


def _yolo_case(preds, gt, dev):
    ops = _ops()
    N = gt.shape[0]
    pd = [p.to(dev).contiguous() for p in preds]
    parts = torch.zeros(N, 5, device=dev)
    dp = [torch.full_like(p, 7.0) for p in pd]
    ws = ops.yolov3_workspace(pd, N, dev)
    ops.yolov3_loss(pd, _yolo_priors_flat(), YR.HEAD_STRIDE, gt.to(dev), (1., 1., 5., 1.), 1.0 / N, parts, dp, ws)
    torch.cuda.synchronize()
    pr = [p.clone().requires_grad_(True) for p in preds]
    dets = [YR.one_image_loss([p[i] for p in pr], gt[i], detail=True) for i in range(N)]
    torch.stack([d['total'] for d in dets]).mean().backward()
    got = parts.cpu()
    for i, d in enumerate(dets):
        for j, key in enumerate(('coord', 'cls', 'obj', 'noobj', 'total')):
            r = float(d[key].detach())
            assert abs(float(got[i, j]) - r) <= 3e-5 * abs(r) + 1e-5, (i, key, float(got[i, j]), r)
    scale = max(float(p.grad.abs().max()) for p in pr)
    for l in range(3):
        assert float((dp[l].cpu() - pr[l].grad).abs().max()) <= 2e-5 * scale, l
    return got


def test_yolov3_loss_golden_inputs(dev):
    g = np.load(os.path.join(GOLD, 'yolov3_loss.npz'))
    preds = [torch.from_numpy(g[f'pred{l + 1}'].astype(np.float32)) for l in range(3)]
    got = _yolo_case(preds, torch.from_numpy(g['gt']), dev)
    for i in range(got.shape[0]):                                     # and directly against the reference's numbers
        assert abs(float(got[i, 4]) - float(g['loss'][i])) <= 5e-5 * abs(float(g['loss'][i]))


def test_yolov3_loss_config4_shape(dev):
    """BASELINE config 4 geometry: 416 x 416 -> 13 / 26 / 52 grids x 3 priors = 10 647 predictions, 20 classes."""
    g = torch.Generator().manual_seed(17)
    N = 3
    preds = [torch.randn(N, h, h, 3, 25, generator=g) * 1.2 for h in (13, 26, 52)]
    assert sum(p.shape[1] * p.shape[2] * 3 for p in preds) == 10647
    _yolo_case(preds, YR.synthetic_gt(N, 416, 29), dev)


def test_yolov3_decode_candidates(dev):
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'yolov3_loss.npz'))
    preds = [torch.from_numpy(g[f'pred{l + 1}'].astype(np.float32))[0] for l in range(3)]
    conf, box = ops.yolov3_decode_candidates([p.to(dev).contiguous() for p in preds], _yolo_priors_flat(), [32., 32., 16.])
    rc, rb = YR.decode_candidates(preds)
    assert float((conf.cpu() - rc).abs().max()) <= 1e-6
    assert float((box.cpu() - rb).abs().max()) <= 1e-3 * 32 / 16         # exp / sigmoid round-off x stride
    assert np.abs(conf.cpu().numpy()[::3] - g['confidence']).max() <= 1e-6
    assert np.abs(box.cpu().numpy()[::3] - g['bbox']).max() <= 2e-3


# ------------------------------------------------------------------------------------------------ full inference tails
def _same_detections(got, scores, bbox, class_id, stol=1e-6, btol=2e-3):
    s, b, c = (t.cpu().numpy() for t in got)
    assert np.array_equal(c, class_id)                                   # same classes, same pick order
    assert np.abs(s - scores).max() <= stol and np.abs(b - bbox).max() <= btol


def test_full_inference_tails_vs_reference(dev):
    """odtk.heads.*_detect = decode kernel -> threshold -> per-class NMS, against the detections the reference's own
    inference branches produce on the same head outputs (tests/golden/*: YOLOv3.py:320-368, FCOS.py:197-265,
    RetinaNet.py:224-256 run on the shim)."""
    from odtk import heads
    g = np.load(os.path.join(GOLD, 'yolov3_loss.npz'))
    preds = [torch.from_numpy(g[f'pred{l + 1}'].astype(np.float32))[0].to(dev).contiguous() for l in range(3)]
    _same_detections(heads.yolov3_detect(preds, _yolo_priors_flat(), 0.45, 10, 0.5), g['det_scores'], g['det_bbox'], g['det_class_id'])
    g = np.load(os.path.join(GOLD, 'fcos_loss.npz'))
    conf, reg, cen = ([torch.from_numpy(g[f'{n}{l}'].astype(np.float32))[0].to(dev).contiguous() for l in range(5)] for n in ('conf', 'reg', 'center'))
    _same_detections(heads.fcos_detect(conf, reg, cen, 0.2, 10, 0.5), g['det_scores'], g['det_bbox'], g['det_class_id'])
    from oracle import retinanet_ref as RR
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'retina_loss.npz'))
    d = np.load(os.path.join(GOLD, 'retina_det.npz'))
    shapes = RR.pyramid_shapes(320, 256)
    flat = [v for s in RR.ANCHOR_SIZES for hw in RR.level_priors(s) for v in hw]
    anc = ops.retina_anchors(256, shapes, [RR.NUM_ANCHORS] * 5, flat, dev)
    pconf = torch.from_numpy(g['pconf'].astype(np.float32))[0].to(dev).contiguous()
    pbox = torch.from_numpy(g['pbox'].astype(np.float32))[0].to(dev).contiguous()
    _same_detections(heads.retina_detect(pconf, pbox, anc[2], anc[3], 0.35, 10, 0.5), d['scores'], d['bbox'], d['class_id'])
    g = np.load(os.path.join(GOLD, 'centernet_loss.npz'))
    kp, off, size = (torch.from_numpy(g[k].astype(np.float32))[0].to(dev) for k in ('keypoints', 'offset', 'size'))
    s, b, c = heads.centernet_detect(kp, off, size, 0.1, 100)
    assert np.array_equal(c.cpu().numpy(), g['det0_class_id']) and np.abs(b.cpu().numpy() - g['det0_bbox']).max() <= 1e-4
