"""GPU parity of YOLOv2 through the C-ABI against oracle/yolov2_ref.py, which is pinned on two training steps and the test graph of the
reference's own class (tests/golden/yolov2_train.npz).  Box side: odtk_yolov2_loss (the four sums, the total, the gradient against autograd of the oracle,
duplicate boxes in one cell, determinism) and odtk_yolov2_decode_candidates; whole model (f32 engine): predictions, loss, every gradient, momentum update,
moving statistics, detections, class surface, a bf16 run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import yolov2_ref as YR            # noqa: E402

SCALES = (1., 1., 5., 1.)
CONFIG = {'mode': 'train', 'is_pretraining': False, 'data_shape': [416, 416, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'data_format': 'channels_last', 'batch_size': 2, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'nms_score_threshold': 0.5,
          'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'rescore_confidence': False, 'priors': YR.PRIORS, 'verbose': False, 'compute_dtype': 'f32'}


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return float((a - b).norm()) / (float(b.norm()) + 1e-30)


@pytest.mark.parametrize("geom", [(3, 13, 13, 5, 20), (2, 15, 15, 5, 20), (2, 6, 7, 3, 4)])
def test_loss_kernel_matches_oracle(geom, dev):
    from odtk import ops
    N, H, W, P, C = geom
    g = torch.Generator().manual_seed(7)
    pred = torch.randn(N, H, W, P, C + 5, generator=g)
    gt = YR.synthetic_gt(N, min(H, W) * 32, 8, pad=10, max_obj=6, num_classes=C)
    gt[0, 1, :2] = gt[0, 0, :2] + 3.0                                    # two box centres in ONE cell: both write, gradients add
    gt[0, 1, 2:4] = gt[0, 0, 2:4] * 0.5
    priors = YR.PRIORS[:P]
    flat = [v for hw in priors for v in hw]
    x = pred.clone().requires_grad_(True)
    per = torch.stack([YR.image_loss(x[i], gt[i], priors, SCALES, C) for i in range(N)])
    per.sum().backward()
    parts = torch.zeros(N, 5, device=dev)
    d = torch.full((N, H * W * P, C + 5), 9.0, device=dev)
    ops.yolov2_loss(pred.to(dev), flat, 32.0, gt.to(dev), SCALES, 0.5, parts, d)
    torch.cuda.synchronize()
    np.testing.assert_allclose(parts[:, 4].cpu().numpy(), per.detach().numpy(), rtol=2e-5)
    want = (x.grad * 0.5).reshape(N, -1, C + 5)
    assert float((d.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7
    assert abs(float((SCALES[0] * parts[:, 0] + SCALES[3] * parts[:, 1] + SCALES[2] * parts[:, 2] + SCALES[1] * parts[:, 3] - parts[:, 4]).abs().max())) < 1e-3
    d2 = torch.zeros_like(d); parts2 = torch.zeros_like(parts)
    ops.yolov2_loss(pred.to(dev), flat, 32.0, gt.to(dev), SCALES, 0.5, parts2, d2)
    assert torch.equal(d, d2) and torch.equal(parts, parts2)             # fixed summation order


def test_decode_candidates_and_detections(dev):
    from odtk import heads, ops
    g = torch.Generator().manual_seed(9)
    pred0 = torch.randn(13, 13, 5, 25, generator=g) * 2
    flat = [v for hw in YR.PRIORS for v in hw]
    conf, bbox = ops.yolov2_decode_candidates(pred0.to(dev), flat, 32.0)
    wc, wb = YR.decode(pred0, YR.PRIORS)
    np.testing.assert_allclose(conf.cpu().numpy(), wc.numpy(), atol=1e-6)
    np.testing.assert_allclose(bbox.cpu().numpy(), wb.numpy(), rtol=1e-5, atol=1e-3)
    from oracle.detect_common import per_class_nms
    got = heads.yolov2_detect(pred0.to(dev), flat, 0.5, 10, 0.5)
    want = per_class_nms(wc, wb, 20, 0.5, 10, 0.5)
    assert len(want[0]) > 0 and np.array_equal(got[2].cpu().numpy(), want[2].numpy())
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0].numpy(), atol=1e-6)
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1].numpy(), rtol=1e-5, atol=1e-3)


def _batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 416, 416, 3, generator=g) * 255).round(), YR.synthetic_gt(n, 416, seed + 1, pad=8, max_obj=4)


def _model(mode, batch, provider=None, **kw):
    import odtk
    return odtk.YOLOv2(dict(CONFIG, mode=mode, batch_size=batch, **kw), provider)


def _provider(batches):
    return {'data_shape': [416, 416, 3], 'num_train': sum(b[0].shape[0] for b in batches), 'num_val': 0, 'train_generator': batches, 'val_generator': None}


def test_f32_model_matches_oracle_forward_loss_gradients_and_step(dev):
    torch.set_num_threads(16)
    p = YR.init_params(43)
    imgs, gt = _batch(2, 310)
    m = _model('train', 2, _provider([(imgs, gt)]))
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.001).item())
    torch.cuda.synchronize()
    with torch.no_grad():
        want = YR.forward(p, imgs, True)
    assert float((m.pred.cpu().view(want.shape) - want).abs().max()) < 2e-3 * (float(want.abs().max()) + 1)
    q = {k: v.clone() for k, v in p.items()}
    mom = {k: torch.zeros_like(p[k]) for k in YR.trainable_names(p)}
    total, data, grads = YR.train_step(q, mom, imgs, gt, 0.001)
    assert abs(loss - total) < 2e-3 * abs(total), (loss, total)
    errs, worst = [], ('', 0.)
    for k in YR.trainable_names(p):
        if k.endswith('.b'):
            assert float(m.get_param(k, m.G).abs().max()) == 0.0
            continue
        err = _rel(m.get_param(k, m.G), grads[k] - 1e-4 * p[k])
        errs.append(err)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err < 3e-2, (k, err)
    errs.sort()
    print('relative gradient error: median', errs[len(errs) // 2], 'worst', worst)
    after = m.export_params()
    for k in q:
        if k.endswith('.b'):
            continue
        step = q[k] - p[k]
        if float(step.norm()) > 1e-12:
            assert _rel(after[k] - p[k], step) < 3e-2, k


def test_inference_class_surface_and_bf16(dev, tmp_path):
    torch.set_num_threads(16)
    p = YR.init_params(83)                                               # the calibration of tests/golden/make_golden_yolov2.py
    g = torch.Generator().manual_seed(1100)
    img = (torch.rand(1, 416, 416, 3, generator=g) * 255).round()
    stats = {}
    with torch.no_grad():
        YR.forward(p, img, True, stats_out=stats, subtract_mean=False)
    for name, (mean, unb) in stats.items():
        p[name + '.mmean'], p[name + '.mvar'] = mean.clone(), unb.clone()
    p['pred.beta'] = p['pred.beta'] + 1.5
    m = _model('test', 1)
    m.load_oracle_params(p)
    got = m.test_one_image(img.numpy())
    gold = np.load(__file__.rsplit('/', 1)[0] + '/golden/yolov2_train.npz')
    assert len(gold['det_scores']) > 0 and np.array_equal(got[2], gold['det_class'])          # the REFERENCE's own detections
    np.testing.assert_allclose(got[0], gold['det_scores'], atol=2e-3)
    assert ((np.abs(got[1] - gold['det_bbox']) <= 2.0 + 5e-3 * np.abs(gold['det_bbox'])).all(axis=1)).mean() >= 0.95
    batches = [_batch(2, 330), _batch(2, 332)]
    t = _model('train', 2, _provider(batches))
    l0 = t.train_one_epoch(0.001)
    assert np.isfinite(l0) and t.global_step == 2
    path = str(tmp_path / 'y' / 'yolov2')
    t.save_weight('latest', path)
    t2 = _model('train', 2, _provider(batches), seed=5)
    t2.load_weight(path + '-2')
    a, b = t.export_params(), t2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(t.Mom, t2.Mom) and t2.global_step == 2
    losses = {}
    for dt in ('f32', 'bf16'):
        mm = _model('train', 2, _provider(batches), compute_dtype=dt, seed=3)
        mm.set_batch(*batches[0])
        losses[dt] = [float(mm.train_step(0.0005).item()) for _ in range(6)]
    assert abs(losses['bf16'][0] - losses['f32'][0]) < 6e-2 * losses['f32'][0], losses
    assert losses['bf16'][-1] < losses['bf16'][0] and losses['f32'][-1] < losses['f32'][0], losses
