"""GPU parity of the whole PFPNetR model through the C-ABI against oracle/pfpnet_net_ref.py, which is pinned on two training
steps of the reference's own class (tests/golden/pfpnet_train.npz).  f32 engine (the class default): predictions, loss, every gradient, the
momentum update and the moving statistics; inference detections; the class surface; a bf16 run."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pfpnet_net_ref as NR        # noqa: E402
from oracle import refinedet_ref as FR         # noqa: E402

CONFIG = {'mode': 'train', 'input_size': 320, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 2,
          'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': 'f32'}


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 320, 320, 3, generator=g) * 255).round(), FR.synthetic_gt(n, 320, seed + 1, pad=8, max_obj=4)


def _model(mode, batch, provider=None, **kw):
    import odtk
    return odtk.PFPNetR(dict(CONFIG, mode=mode, batch_size=batch, **kw), provider)


def _provider(batches):
    return {'data_shape': [320, 320, 3], 'num_train': sum(b[0].shape[0] for b in batches), 'num_val': 0, 'train_generator': batches, 'val_generator': None}


def _rel(a, b):
    return float((a - b).norm()) / (float(b.norm()) + 1e-30)


def test_f32_model_matches_oracle_forward_loss_gradients_and_step(dev):
    torch.set_num_threads(16)
    p = NR.init_params(33)
    imgs, gt = _batch(2, 210)
    m = _model('train', 2, _provider([(imgs, gt)]))
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.001).item())
    torch.cuda.synchronize()
    with torch.no_grad():
        want = NR.forward(p, imgs, True)
    for got, w, tag in zip((m.arm_loc, m.arm_conf, m.odm_loc, m.odm_conf), want, ('arm_loc', 'arm_conf', 'odm_loc', 'odm_conf')):
        assert float((got.cpu() - w).abs().max()) < 2e-3 * (float(w.abs().max()) + 1), tag
    q = {k: v.clone() for k, v in p.items()}
    mom = {k: torch.zeros_like(p[k]) for k in NR.trainable_names(p)}
    total, data, grads = NR.train_step(q, mom, imgs, gt, 0.001)
    assert abs(loss - total) < 2e-3 * abs(total), (loss, total)
    errs, worst = [], ('', 0.)
    for k in NR.trainable_names(p):
        if k.endswith('.b') and (k[:-2] + '.gamma') in p:
            assert float(m.get_param(k, m.G).abs().max()) == 0.0
            continue
        w = grads[k] - 1e-4 * p[k]
        if k.endswith('_l2_norm'):
            # every consumer of the scaled feature map starts with conv + BATCH NORM, which is invariant to the scale of its input (up to its
            # epsilon): the true gradient of the scalar is ~0 and what is computed is the round-off of a sum of 400 K mixed-sign terms
            assert float((m.get_param(k, m.G) - w).abs().max()) <= 0.3 * float(w.abs().max()) + 1e-4, (k, m.get_param(k, m.G), w)
            continue
        if re.fullmatch(r'fl\d_\dd\.beta', k):
            # the offset of an up-path transposed conv's batch norm goes through `+ fl_b`, a 1x1 conv and the NEXT batch norm, which removes any
            # per-channel constant: the true gradient is 0 and both sides hold round-off (compare against the scale of the layer's gamma gradient)
            assert float((m.get_param(k, m.G) - w).abs().max()) <= 1e-2 * float(grads[k[:-5] + '.gamma'].abs().max()) + 1e-5, k
            continue
        err = _rel(m.get_param(k, m.G), w)
        errs.append(err)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err < 6e-2, (k, err)                             # (ReLU flips of ~1e-6 pre-activations in front of 81 batch norms at batch 2)
    errs.sort()
    print('relative gradient error: median', errs[len(errs) // 2], 'worst', worst)
    after = m.export_params()
    for k in q:
        if k.endswith('.b') and (k[:-2] + '.gamma') in q:
            continue
        step = q[k] - p[k]
        if float(step.norm()) > 1e-12 and not k.endswith('_l2_norm'):
            assert _rel(after[k] - p[k], step) < 6e-2, k


def test_inference_class_surface_and_bf16(dev, tmp_path):
    torch.set_num_threads(16)
    p = NR.init_params(39)
    imgs, gt = _batch(2, 220)
    stats = {}
    with torch.no_grad():
        NR.forward(p, imgs, True, stats_out=stats, subtract_mean=False)
    for name, (mean, unb) in stats.items():
        p[name + '.mmean'], p[name + '.mvar'] = mean.clone(), unb.clone()
    m = _model('test', 1)
    m.load_oracle_params(p)
    got = m.test_one_image(imgs[:1].numpy())
    want = NR.test_one_image(p, imgs[:1], 0.1, 20, 0.45)
    assert len(want[0]) > 0 and np.array_equal(got[2], want[2].numpy())
    np.testing.assert_allclose(got[0], want[0].numpy(), atol=2e-3)
    w = want[1].numpy()
    assert ((np.abs(got[1] - w) <= 2.0 + 5e-3 * np.abs(w)).all(axis=1)).mean() >= 0.95
    batches = [_batch(2, 230), _batch(2, 232)]
    t = _model('train', 2, _provider(batches))
    l0 = t.train_one_epoch(0.001)
    assert np.isfinite(l0) and t.global_step == 2
    path = str(tmp_path / 'r' / 'pfpnet')
    t.save_weight('latest', path)
    t2 = _model('train', 2, _provider(batches), seed=5)
    t2.load_weight(path + '-2')
    a, b = t.export_params(), t2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(t.Mom, t2.Mom) and t2.global_step == 2
    losses = {}
    for dt in ('f32', 'bf16'):
        mm = _model('train', 2, _provider(batches), compute_dtype=dt, seed=3)
        mm.set_batch(*batches[0])
        losses[dt] = [float(mm.train_step(0.001).item()) for _ in range(6)]
    assert abs(losses['bf16'][0] - losses['f32'][0]) < 6e-2 * losses['f32'][0], losses
    assert losses['bf16'][-1] < losses['bf16'][0] and losses['f32'][-1] < losses['f32'][0], losses
