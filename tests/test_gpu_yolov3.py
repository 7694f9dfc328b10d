"""GPU parity of the whole YOLOv3 model (BASELINE config 4) through the C-ABI against oracle/yolov3_net_ref.py,
which is pinned on the reference's own network code (tests/golden/yolov3_net.npz):
  * glue kernels: batch norm with leaky_relu(0.1) forward / backward, residual sum / pitched copy, 2x nearest up-sampling;
  * f32 engine: predictions (train and inference mode), loss, EVERY gradient, the parameters / moving statistics after a step;
  * bf16 engine (the production dtype): loss and update direction;
  * class surface: train_one_epoch, test_one_image (detections equal to the oracle's), checkpoint round trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import yolov3_net_ref as NR   # noqa: E402
from oracle import yolov3_ref as YR       # noqa: E402

CONFIG = {
    'mode': 'train', 'data_shape': [64, 64, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
    'batch_size': 2, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3,
    'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'priors': YR.PRIORS_PX, 'verbose': False,
}


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _batch(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    imgs = (torch.rand(n, size, size, 3, generator=g) * 255).round()
    return imgs, YR.synthetic_gt(n, size, seed + 1, max_obj=3)


def _model(mode, dtype, batch, size, provider=None, **kw):
    import odtk
    cfg = dict(CONFIG, mode=mode, compute_dtype=dtype, batch_size=batch, data_shape=[size, size, 3], **kw)
    return odtk.YOLOv3(cfg, provider)


def _provider(batches):
    return {'num_train': sum(b[0].shape[0] for b in batches), 'train_generator': batches, 'val_generator': None, 'num_val': 0}


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_glue_kernels(dev, dt):
    import odtk  # noqa: F401
    from odtk import ops
    tdt = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(0)
    N, H, W, C1, C2 = 2, 5, 7, 16, 24
    bottom = torch.randn(N * 4 * H * W, C1, generator=g).to(tdt).to(dev)
    lat = torch.randn(N * H * W, C2, generator=g).to(tdt).to(dev)
    cat = torch.zeros(N * 4 * H * W, C1 + C2, dtype=tdt, device=dev)
    ops.add2d(bottom, C1, None, 0, cat, C1 + C2, bottom.shape[0], C1)
    ops.upsample2x_fwd(lat, C2, cat[:, C1:], C1 + C2, N, H, W, C2)
    up = lat.view(N, H, W, C2).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1, C2)
    assert torch.equal(cat[:, :C1], bottom) and torch.equal(cat[:, C1:], up)
    s = torch.zeros_like(bottom)
    ops.add2d(bottom, C1, cat[:, :C1], C1 + C2, s, C1, bottom.shape[0], C1)
    assert torch.equal(s, (bottom.float() * 2).to(tdt))
    dlat = torch.zeros_like(lat)
    ops.upsample2x_bwd(cat[:, C1:], C1 + C2, dlat, C2, N, H, W, C2, False)
    torch.testing.assert_close(dlat.float(), lat.float() * 4, rtol=1e-2 if dt == 'bf16' else 1e-6, atol=0)
    ops.upsample2x_bwd(cat[:, C1:], C1 + C2, dlat, C2, N, H, W, C2, True)
    torch.testing.assert_close(dlat.float(), lat.float() * 8, rtol=2e-2 if dt == 'bf16' else 1e-6, atol=0)
    # batch norm + leaky_relu(0.1), forward and backward against autograd
    M, C = 300, 24
    z = torch.randn(M, C, generator=g).to(tdt).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    mm, mv, sm, si = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y = torch.zeros_like(z)
    ws = torch.zeros(ops.bn_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    ops.bn_fwd(z, M, C, C, gamma, beta, mm, mv, sm, si, True, 2, y, C, M, 0, ws)
    zr = z.float().cpu().requires_grad_(True); gr, br = gamma.cpu().requires_grad_(True), beta.cpu().requires_grad_(True)
    mean, var = zr.mean(0), zr.var(0, unbiased=False)
    yr = torch.nn.functional.leaky_relu((zr - mean) * torch.rsqrt(var + 1e-3) * gr + br, 0.1)
    tol = 2e-2 if dt == 'bf16' else 2e-5
    torch.testing.assert_close(y.float().cpu(), yr.detach(), rtol=tol, atol=tol)
    dy = torch.randn(M, C, generator=g).to(tdt)
    yr.backward(dy.float())
    dz, dg, db = torch.zeros_like(z), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_bwd(z, y, dy.to(dev), M, C, C, C, M, 0, gamma, sm, si, 2, dz, dg, db, ws)
    if dt == 'f32':                       # in bf16 the sign of a rounded y near 0 may differ from autograd's: compare in f32 only
        torch.testing.assert_close(dz.cpu(), zr.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)
    else:
        assert float((dz.float().cpu() - zr.grad).norm() / zr.grad.norm()) < 3e-2


def test_f32_model_matches_oracle_forward_loss_gradients_and_step(dev):
    torch.set_num_threads(16)
    p = NR.init_params(5)
    imgs, gt = _batch(2, 64, 40)
    m = _model('train', 'f32', 2, 64, _provider([(imgs, gt)]))
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.01).item())
    mom = {k: torch.zeros_like(v) for k, v in p.items() if k in NR.trainable_names(p)}
    q = {k: v.clone() for k, v in p.items()}
    preds_ref = [t.detach() for t in NR.forward(q, imgs, True)]
    # gradients are compared on the linear region the GPU took: which side of 0 a pre-activation of ~1e-6 falls on is round-off,
    # and ONE such element moves every upstream gradient by ~1 % (oracle/yolov3_net_ref.forward); the flips are counted below
    masks, flips = {}, 0
    taps = {}
    with torch.no_grad():
        NR.forward(q, imgs, True, taps=taps)
    for name, _, _, _, _, act in NR.layer_specs():
        if act:
            a = m.acts[name]
            masks[name] = (a.t[:, :a.C].float().cpu() > 0).view(a.N, a.H, a.W, a.C).permute(0, 3, 1, 2)
            flips += int((masks[name] != (taps[name] > 0)).sum())
    print('leaky sign flips against the free-running oracle:', flips, 'of', sum(v.numel() for v in masks.values()))
    total, data, grads = NR.train_step(q, mom, imgs, gt, 0.01, leaky_masks=masks)
    for got, want in zip(m.preds, preds_ref):
        assert float((got.cpu() - want).abs().max()) < 2e-3 * (float(want.abs().max()) + 1), 'training-mode predictions'
    assert abs(loss - total) < 2e-3 * abs(total), (loss, total)
    assert abs(0.5 * float(m.loss_parts[:, 4].mean()) - 0.5 * data) < 2e-3 * abs(data)
    worst, errs_all, errs_w = ('', 0.), [], []
    for k in NR.trainable_names(p):
        if k.endswith('.b'):
            continue                      # feeds batch norm: exactly 0 here, round-off noise in autograd
        got = m.get_param(k, m.G)
        want = grads[k] - 5e-4 * p[k]     # the kernel's G holds the data gradient; weight decay is applied by the optimizer
        if k in ('c59.beta', 'c67.beta'):
            # the lateral convs have no activation: their beta only shifts the next conv's output by a per-channel constant, which
            # that conv's batch norm removes -- the true gradient is 0 and both sides hold round-off noise
            scale = float(grads[k[:-4] + 'gamma'].norm())
            assert float(got.norm()) < 1e-3 * scale and float(want.norm()) < 1e-3 * scale, k
            continue
        err = float((got - want).norm()) / (float(want.norm()) + 1e-8)
        worst = max(worst, (k, err), key=lambda t: t[1])
        errs_all.append(err)
        if k.endswith('.w'):
            errs_w.append(err)
        assert err < 2e-3, (k, err)
    print('by depth', [(k, round(e, 5)) for k, e in zip([n for n in NR.trainable_names(p) if n.endswith('.w')], errs_w)][::4])
    errs = sorted(errs_all)
    print('relative gradient error: median', errs[len(errs) // 2], 'worst', worst)
    assert errs[len(errs) // 2] < 5e-4
    after = m.export_params()
    for k in q:
        if k.endswith(('.mmean', '.mvar')):
            err = float((after[k] - q[k]).norm()) / (float(q[k].norm()) + 1e-6)
            assert err < 2e-3, (k, err)
        elif not k.endswith('.b') and k not in ('c59.beta', 'c67.beta'):
            step = q[k] - p[k]                    # the optimizer update itself: lr * (gradient + weight decay)
            err = float((after[k] - p[k] - step).norm()) / (float(step.norm()) + 1e-12)
            assert err < 2e-3, (k, err)


def test_bf16_step_direction_and_class_surface(dev, tmp_path):
    torch.set_num_threads(16)
    p = NR.init_params(6)
    batches = [_batch(2, 128, 50), _batch(2, 128, 52)]
    m = _model('train', 'bf16', 2, 128, _provider(batches))
    m.load_oracle_params(p)
    m.set_batch(*batches[0])
    loss = float(m.train_step(0.005).item())
    q = {k: v.clone() for k, v in p.items()}
    mom = {k: torch.zeros_like(v) for k, v in p.items() if k in NR.trainable_names(p)}
    total, data, grads = NR.train_step(q, mom, batches[0][0], batches[0][1], 0.005)
    assert abs(loss - total) < 6e-2 * abs(total), (loss, total)
    cos = []
    for k in ('c74.w', 'c66.w', 'c58.w', 'c51.w', 'c26.w', 'c5.w', 'c0.w', 'c74.gamma', 'c59.w'):
        a, b = m.get_param(k, m.G).reshape(-1), (grads[k] - 5e-4 * p[k]).reshape(-1)
        cos.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-12)))
    print('bf16 gradient cosines', cos)
    # bf16 through 75 batch norms over as few as 32 samples (2 x 4 x 4): the direction survives, the digits do not
    assert min(cos) > 0.35 and cos[0] > 0.8 and cos[7] > 0.99
    l0 = m.train_one_epoch(0.001)
    assert np.isfinite(l0) and m.global_step == 3
    path = str(tmp_path / 'y' / 'yolo')
    m.save_weight('latest', path)
    m2 = _model('test', 'bf16', 1, 128)
    m2.load_weight(path + '-3')
    a, b = m.export_params(), m2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a)
    out = m2.test_one_image(batches[0][0][:1].numpy())
    assert len(out) == 3 and out[1].shape[1] == 4


def test_f32_inference_detections_equal_oracle(dev):
    torch.set_num_threads(16)
    p = NR.init_params(8)
    imgs, _ = _batch(1, 128, 60)
    stats = {}
    with torch.no_grad():                  # moving statistics = batch statistics of a nearby picture: logits in a lively range
        NR.forward(p, imgs + 20 * torch.randn(imgs.shape, generator=torch.Generator().manual_seed(2)), True, stats, subtract_mean=False)
    for k, (mean, var) in stats.items():
        p[k + '.mmean'], p[k + '.mvar'] = mean.clone(), var.clone()
    m = _model('test', 'f32', 1, 128, nms_score_threshold=0.5)
    m.load_oracle_params(p)
    got = m.test_one_image(imgs.numpy())
    want = NR.test_one_image(p, imgs, 0.5, 10, 0.5)
    with torch.no_grad():
        for a, b in zip(m.preds, NR.forward(p, imgs, False, subtract_mean=False)):
            assert float((a.cpu() - b).abs().max()) < 2e-3 * (float(b.abs().max()) + 1)
    assert len(want[0]) > 0 and len(got[0]) == len(want[0])
    assert np.array_equal(got[2], want[2].numpy())
    np.testing.assert_allclose(got[0], want[0].numpy(), atol=2e-3)
    np.testing.assert_allclose(got[1], want[1].numpy(), atol=0.5)


def test_tf_saver_checkpoint_roundtrip(dev, tmp_path):
    """checkpoint_format='tf': the files tf.train.Saver would leave (YOLOv3.py:466-478), every variable of the reference's graph under
    its name and shape (tests/golden/yolov3_variables.json) + momentum slots + global_step, and back; `load_pretraining_weight`
    restores the trainable 'backone' variables only (:377-378, :480-482)."""
    import json
    import os
    from odtk import tf_checkpoint as T
    batches = [_batch(2, 64, 70)]
    m = _model('train', 'bf16', 2, 64, _provider(batches), checkpoint_format='tf', seed=1)
    m.train_one_epoch(0.001)
    path = str(tmp_path / 'ck' / 'yolo.ckpt')
    m.save_weight('latest', path)
    r = T.NewCheckpointReader(path + '-1')
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'yolov3_variables.json')))
    shapes = r.get_variable_to_shape_map()
    for name, info in want.items():
        assert shapes[name] == info['shape'], name
        assert (name + '/Momentum' in shapes) == info['trainable'], name
    assert len(shapes) == len(want) + sum(v['trainable'] for v in want.values())
    m2 = _model('train', 'bf16', 2, 64, _provider(batches), seed=2)
    m2.load_weight(path + '-1')
    a, b = m.export_params(), m2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(m.Mom, m2.Mom) and m2.global_step == 1
    m3 = _model('train', 'bf16', 2, 64, _provider(batches), seed=3)
    before = m3.export_params()
    m3.load_pretraining_weight(path + '-1')
    c = m3.export_params()
    for k in a:
        layer = int(k[1:].split('.')[0])
        restored = layer < 52 and not k.endswith(('.mmean', '.mvar'))
        assert torch.equal(c[k], a[k] if restored else before[k]), k


def test_graph_replay_equals_eager_launches(dev):
    """the HIP-graph replay of forward + loss + backward (default from the third step on) against eager launches of the same
    steps: f32 engine, equality up to the float-atomic order of the filter gradients"""
    batch = _batch(2, 64, 80)
    out = {}
    for use_graph in (False, True):
        m = _model('train', 'f32', 2, 64, _provider([batch]), seed=4, use_graph=use_graph)
        m.set_batch(*batch)
        # two eager steps, then the first replayed one.  Not more: the float-atomic order of the filter gradients differs from run
        # to run, and training at batch 2 amplifies that by ~5x per step (0.6 % of the loss after five steps, measured) whether
        # or not a graph is involved -- the third step still separates "same launches" from "wrong graph" by orders of magnitude
        losses = [float(m.train_step(0.002)) for _ in range(3)]
        assert (m._graph is not None) == use_graph
        out[use_graph] = (losses, m.P.clone())
    for a, b in zip(out[False][0], out[True][0]):
        assert abs(a - b) <= 2e-3 * abs(a), (out[False][0], out[True][0])
    step = out[False][1] - _model('train', 'f32', 2, 64, _provider([batch]), seed=4).P
    assert float((out[True][1] - out[False][1]).norm()) < 5e-2 * float(step.norm())


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_resize_bilinear_feature_maps(dev, dt):
    """tf.image.resize_bilinear on NHWC feature maps (TF-1.x grid), forward, accumulate form, and the gather backward against
    autograd of the CPU restatement (oracle/augment_ref.resize_bilinear_legacy): the pyramid glue of RetinaNet / FCOS"""
    import odtk  # noqa: F401
    from odtk import ops
    from oracle import augment_ref as AR
    tdt = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    for (N, H, W, Ho, Wo, C, ld) in [(2, 5, 7, 10, 14, 16, 16), (1, 4, 4, 7, 9, 8, 24), (2, 3, 5, 3, 5, 8, 8), (1, 13, 13, 25, 25, 32, 32)]:
        x = torch.randn(N * H * W, ld, generator=g).to(tdt)
        xd = x.to(dev)
        y = torch.zeros(N * Ho * Wo, C, dtype=tdt, device=dev)
        ops.resize_bilinear_fwd(xd, ld, y, C, N, H, W, Ho, Wo, C)
        xr = x[:, :C].float().view(N, H, W, C).clone().requires_grad_(True)
        want = torch.stack([AR.resize_bilinear_legacy(xr[n], Ho, Wo) for n in range(N)])
        tol = 2e-2 if dt == 'bf16' else 1e-5
        torch.testing.assert_close(y.float().cpu().view(N, Ho, Wo, C), want.detach(), rtol=tol, atol=tol)
        ops.resize_bilinear_fwd(xd, ld, y, C, N, H, W, Ho, Wo, C, accumulate=True)
        torch.testing.assert_close(y.float().cpu().view(N, Ho, Wo, C), 2 * want.detach(), rtol=2 * tol, atol=2 * tol)
        dy = torch.randn(N * Ho * Wo, C, generator=g).to(tdt)
        want.backward(dy.float().view(N, Ho, Wo, C))
        dx = torch.zeros(N * H * W, ld, dtype=tdt, device=dev)
        ops.resize_bilinear_bwd(dy.to(dev), C, dx, ld, N, H, W, Ho, Wo, C)
        torch.testing.assert_close(dx[:, :C].float().cpu().view(N, H, W, C), xr.grad, rtol=tol, atol=2 * tol)
        assert float(dx[:, C:].abs().sum()) == 0                          # pad columns of the pitched operand untouched
