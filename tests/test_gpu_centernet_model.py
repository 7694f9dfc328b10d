"""GPU parity of the whole CenterNet model (BASELINE config 5) through the C-ABI against oracle/centernet_net_ref.py, which is pinned on two
training steps of the reference's own class (tests/golden/centernet_train.npz).  f32 engine (the class default): the three new
element-wise kernels alone; predictions, loss, EVERY gradient on the GPU's ReLU region, the Adam update, the moving statistics;
inference detections; the class surface (train_one_epoch, save / load, load_pretrained_weight)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import centernet_net_ref as NR     # noqa: E402
from oracle import centernet_ref as CR         # noqa: E402

CONFIG = {'mode': 'train', 'input_size': 128, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'batch_size': 2, 'score_threshold': 0.1, 'top_k_results_output': 100, 'verbose': False, 'compute_dtype': 'f32'}


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _batch(n, seed, size=128):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, size, size, 3, generator=g) * 255).round(), CR.synthetic_gt(n, size, seed + 1, pad=8, max_obj=4)


def _model(mode, batch, provider=None, **kw):
    import odtk
    return odtk.CenterNet(dict(CONFIG, mode=mode, batch_size=batch, **kw), provider)


def _provider(batches):
    return {'num_train': sum(b[0].shape[0] for b in batches), 'num_val': 0, 'train_generator': batches, 'val_generator': None}


def _rel(a, b):
    return float((a - b).norm()) / (float(b.norm()) + 1e-30)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_elementwise_kernels(dt, dev):
    import odtk  # noqa: F401
    from odtk import ops
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    DT = ops.F32 if dt == 'f32' else ops.BF16
    g = torch.Generator().manual_seed(3)
    # (images / 255 - mean) / std, bit for bit in f32
    img = (torch.rand(2, 16, 12, 3, generator=g) * 255).round()
    ld = ops.pad_to(3, ops.chunk(DT))
    x = torch.full((2 * 16 * 12, ld), 7.0, dtype=dtype, device=dev)
    ops.preprocess_norm(img.to(dev), 255., NR.MEAN, NR.STD, ld, DT, x)
    want = ((img / 255. - torch.tensor(NR.MEAN)) / torch.tensor(NR.STD)).reshape(-1, 3)
    torch.cuda.synchronize()
    assert torch.equal(x[:, :3].cpu(), want.to(dtype)) and float(x[:, 3:].float().abs().max()) == 0.0
    # 2x2 average pooling and its gradient
    N, H, W, C = 2, 6, 8, 16
    a = torch.randn(N * H * W, C, generator=g).to(dtype)
    y = torch.zeros(N * (H // 2) * (W // 2), C, dtype=dtype, device=dev)
    ops.avgpool2x2_fwd(a.to(dev), y, N, H, W, C)
    ar = a.float().reshape(N, H, W, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = torch.nn.functional.avg_pool2d(ar, 2, 2)
    torch.cuda.synchronize()
    assert float((y.float().cpu().reshape(N, H // 2, W // 2, C).permute(0, 3, 1, 2) - yr.detach()).abs().max()) <= (1e-6 if dt == 'f32' else 2e-2)
    dy = torch.randn(N * (H // 2) * (W // 2), C, generator=g).to(dtype)
    dx = torch.full((N * H * W, C), 9.0, dtype=dtype, device=dev)
    ops.avgpool2x2_bwd(dy.to(dev), dx, N, H, W, C)
    yr.backward(dy.float().reshape(N, H // 2, W // 2, C).permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    assert float((dx.float().cpu().reshape(N, H, W, C).permute(0, 3, 1, 2) - ar.grad).abs().max()) <= (1e-6 if dt == 'f32' else 1e-2)


def test_adam_kernel_three_steps(dev):
    """odtk_adam against tf.train.AdamOptimizer's update rule (ApplyAdam) for three steps, with the L2 term and its partial sums"""
    import odtk  # noqa: F401
    from odtk import ops
    n = 70001
    g = torch.Generator().manual_seed(9)
    p = torch.randn(n, generator=g); m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    pc = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    part = torch.zeros(ops.sgd_blocks(n), device=dev); tot = torch.zeros(1, device=dev)
    lr, wd = 1e-3, 1e-4
    for t in (1, 2, 3):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))
        lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        l2_want = 0.5 * float((p.double() ** 2).sum())
        ops.adam(pd, md, vd, grad.to(dev), lr_t, 0.9, 0.999, 1e-8, wd, 1.0, part, pc)
        ops.sum_f32(part, tot)
        f = torch.float32
        b1, b2 = torch.tensor(0.9, dtype=f), torch.tensor(0.999, dtype=f)
        gg = grad + torch.tensor(wd, dtype=f) * p                         # ApplyAdam in float32, its own expression order
        m = m + (gg - m) * (1 - b1)
        v = v + (gg * gg - v) * (1 - b2)
        p = p - (m * torch.tensor(lr_t, dtype=f)) / (torch.sqrt(v) + torch.tensor(1e-8, dtype=f))
        torch.cuda.synchronize()
        assert float((pd.cpu() - p).abs().max()) < 1e-6 and _rel(md.cpu(), m) < 1e-6 and _rel(vd.cpu(), v) < 1e-6
        assert abs(float(tot) - l2_want) < 1e-5 * l2_want
        assert torch.equal(pc.cpu(), pd.cpu().to(torch.bfloat16))


def test_f32_model_matches_oracle_forward_loss_gradients_and_step(dev):
    torch.set_num_threads(16)
    p = NR.init_params(17)
    imgs, gt = _batch(2, 140)
    m = _model('train', 2, _provider([(imgs, gt)]))
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.001).item())
    torch.cuda.synchronize()
    with torch.no_grad():
        kp, off, size = NR.forward(p, imgs, True)
    for got, want, tag in ((m.keypoints, kp, 'keypoints'), (m.offset, off, 'offset'), (m.size, size, 'size')):
        assert float((got.cpu() - want).abs().max()) < 2e-3 * (float(want.abs().max()) + 1), tag
    masks = {}
    for name, kind, _, _, _, _, relu, ghost in NR.layer_specs():
        if relu and not ghost:
            a = m.acts[name]
            masks[name] = (a.t[:, :a.C].float().cpu() > 0).view(a.N, a.H, a.W, a.C).permute(0, 3, 1, 2)
    q = {k: v.clone() for k, v in p.items()}
    total, data, grads = NR.train_step(q, {}, imgs, gt, 0.001, relu_masks=masks)
    assert abs(loss - total) < 2e-3 * abs(total), (loss, total)
    errs, worst = [], ('', 0.)
    for k in NR.trainable_names(p):
        if k.endswith('.b'):
            assert float(m.get_param(k, m.G).abs().max()) == 0.0
            continue
        want = grads[k] - 1e-4 * p[k]
        if float(want.norm()) < 1e-9:
            assert float(m.get_param(k, m.G).norm()) < 1e-9, k               # ghost shortcut layers: no data gradient
            continue
        err = _rel(m.get_param(k, m.G), want)
        errs.append(err)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err < 2e-2, (k, err)
    errs.sort()
    print('relative gradient error: median', errs[len(errs) // 2], 'worst', worst)
    after = m.export_params()
    for k in q:
        if k.endswith('.b'):
            continue
        if k in grads:
            sig = grads[k].abs() > 1e-2 * grads[k].abs().max()                # Adam's first step is lr * sign(g): compare where g is significant
            assert float((after[k] - q[k])[sig].abs().max()) < 5e-5, k
            assert float((after[k] - q[k]).abs().max()) <= 2.01e-3, k
        else:
            assert float((after[k] - q[k]).abs().max()) <= 1e-3 * (float(q[k].abs().max()) + 0.05), k       # moving statistics
    assert torch.equal(after['c8.mmean'], torch.zeros(64)) and torch.equal(after['c8.mvar'], torch.ones(64))
    assert float((after['c8.w'] - p['c8.w']).abs().max()) > 5e-4            # ... but its weights move (L2 term through Adam)


def test_inference_and_class_surface(dev, tmp_path):
    torch.set_num_threads(16)
    p = NR.init_params(19)
    p['c63.beta'] = p['c63.beta'] + 1.0                     # lift the keypoint logits so that peaks pass the score threshold
    imgs, _ = _batch(2, 150)
    # calibrate the moving statistics on the oracle (training-mode forward), as a trained checkpoint would hold them
    stats = {}
    with torch.no_grad():
        NR.forward(p, imgs, True, stats_out=stats, normalize=False)
    for name, (mean, unb) in stats.items():
        p[name + '.mmean'], p[name + '.mvar'] = mean.clone(), unb.clone()
    m = _model('test', 1)
    m.load_oracle_params(p)
    got = m.test_one_image(imgs[:1].numpy())
    with torch.no_grad():
        kp, off, size = NR.forward(p, imgs[:1], False, normalize=False)      # the reference's test-mode feed bypasses the normalisation
    want = CR.decode(kp[0], off[0], size[0], 0.1, 100)
    assert len(want[0]) > 0 and len(got[0]) == len(want[0])
    assert np.array_equal(got[2], want[2].numpy())
    np.testing.assert_allclose(got[0], want[0].numpy(), atol=2e-3)
    np.testing.assert_allclose(got[1], want[1].numpy(), atol=0.5, rtol=5e-3)
    batches = [_batch(2, 160), _batch(2, 162)]
    t = _model('train', 2, _provider(batches))
    l0 = t.train_one_epoch(0.001)
    assert np.isfinite(l0) and t.global_step == 2
    path = str(tmp_path / 'c' / 'centernet')
    t.save_weight('latest', path)
    t2 = _model('train', 2, _provider(batches), seed=5)
    t2.load_weight(path + '-2')
    a, b = t.export_params(), t2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(t.M1, t2.M1) and torch.equal(t.M2, t2.M2) and t2.global_step == 2
    t3 = _model('test', 1, seed=7)
    before = t3.export_params()
    t3.load_pretrained_weight(path + '-2')
    c = t3.export_params()
    assert all(torch.equal(c[k], a[k] if (k in t3.pinfo and int(k[1:].split('.')[0]) < 50) else before[k]) for k in a)
    assert len(t3.test_one_image(imgs[:1].numpy())) == 3


def test_bf16_engine_trains(dev):
    """the bf16 engine on the same graph: the loss agrees with f32 at the first step (6 %) and falls over 8 Adam steps"""
    imgs, gt = _batch(4, 180)
    losses = {}
    for dt in ('f32', 'bf16'):
        m = _model('train', 4, _provider([(imgs, gt)]), compute_dtype=dt, seed=3)
        m.set_batch(imgs, gt)
        losses[dt] = [float(m.train_step(0.001).item()) for _ in range(8)]
    assert abs(losses['bf16'][0] - losses['f32'][0]) < 6e-2 * losses['f32'][0], losses
    assert losses['bf16'][-1] < losses['bf16'][0] and losses['f32'][-1] < losses['f32'][0], losses
