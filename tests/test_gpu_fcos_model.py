"""GPU parity of the whole FCOS model (BASELINE config 5) through the C-ABI against oracle/fcos_net_ref.py, which is pinned on two
training steps of the reference's own class (tests/golden/fcos_train.npz).  f32 engine (the class default):
predictions, loss, EVERY gradient on the GPU's ReLU region, the optimizer update; inference detections; the class surface."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detect_common as DC    # noqa: E402
from oracle import fcos_net_ref as NR     # noqa: E402
from oracle import fcos_ref as FR         # noqa: E402

CONFIG = {'mode': 'train', 'data_shape': [128, 160, 3], 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45, 'verbose': False, 'compute_dtype': 'f32'}


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 128, 160, 3, generator=g) * 255).round(), FR.synthetic_gt(n, 128, seed + 1)


def _model(mode, batch, provider=None, **kw):
    import odtk
    return odtk.FCOS(dict(CONFIG, mode=mode, batch_size=batch, **kw), provider)


def _provider(batches):
    return {'num_train': sum(b[0].shape[0] for b in batches), 'num_val': 0, 'train_generator': batches, 'val_generator': None}


def test_f32_model_matches_oracle_forward_loss_gradients_and_step(dev):
    torch.set_num_threads(16)
    p = NR.init_params(17)
    imgs, gt = _batch(2, 140)
    m = _model('train', 2, _provider([(imgs, gt)]))
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.001).item())
    q = {k: v.clone() for k, v in p.items()}
    with torch.no_grad():
        conf, reg, center = NR.forward(q, imgs)
    for l in range(5):
        assert float((m.conf[l].cpu() - conf[l]).abs().max()) < 2e-3 * (float(conf[l].abs().max()) + 1), ('conf', l)
        assert float((m.reg[l].cpu() - reg[l]).abs().max()) < 5e-3 * (float(reg[l].abs().max()) + 1), ('reg', l)
        assert float((m.center[l].cpu() - center[l]).abs().max()) < 2e-3 * (float(center[l].abs().max()) + 1), ('center', l)
    masks = {}
    for name, a in m.acts.items():                   # ReLU outputs: the stem's 'l0', every other layer's '<instance>.y' (heads: l<k>@<level>)
        if name == 'l0' or name.endswith('.y'):
            masks[name[:-2] if name.endswith('.y') else name] = (a.t[:, :a.C].float().cpu() > 0).view(a.N, a.H, a.W, a.C).permute(0, 3, 1, 2)
    mom = {k: torch.zeros_like(v) for k, v in p.items()}
    total, data, grads = NR.train_step(q, mom, imgs, gt, 0.001, relu_masks=masks)
    assert abs(loss - total) < 2e-3 * abs(total), (loss, total)
    errs, worst = [], ('', 0.)
    for k in p:
        want = grads[k] - 1e-4 * p[k]
        if float(want.norm()) < 1e-7:
            continue
        err = float((m.get_param(k, m.G) - want).norm()) / float(want.norm())
        errs.append(err)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err < 1e-2, (k, err)
    errs.sort()
    print('relative gradient error: median', errs[len(errs) // 2], 'worst', worst)
    after = m.export_params()
    for k in q:
        step = q[k] - p[k]
        if float(step.norm()) > 1e-9:
            assert float((after[k] - p[k] - step).norm()) / float(step.norm()) < 1e-2, k


def test_inference_and_class_surface(dev, tmp_path):
    torch.set_num_threads(16)
    p = NR.init_params(19)
    p['l79.b'] = p['l79.b'] + 4.0                      # class outputs (shared by the five levels): lift the pi bias so that detections exist
    p['l80.b'] = p['l80.b'] + 4.0                      # centre-ness
    p['l85.w'] = p['l85.w'] * 0.05                     # distances: keep exp(t) in a sane range
    imgs, _ = _batch(1, 150)
    m = _model('test', 1, nms_score_threshold=0.3)
    m.load_oracle_params(p)
    got = m.test_one_image(imgs.numpy())
    with torch.no_grad():
        conf, reg, center = NR.forward(p, imgs, subtract_mean=False)
    pconf, pbbox = FR.decode_candidates([c[0] for c in conf], [r[0] for r in reg], [z[0] for z in center])
    want = DC.per_class_nms(pconf, pbbox, 19, 0.3, 10, 0.45)
    assert len(want[0]) > 0 and len(got[0]) == len(want[0])
    assert np.array_equal(got[2], want[2].numpy())
    np.testing.assert_allclose(got[0], want[0].numpy(), atol=2e-3)
    w = want[1].numpy()
    assert ((np.abs(got[1] - w) <= 2.0 + 5e-3 * np.abs(w)).all(axis=1)).mean() >= 0.95
    batches = [_batch(2, 160), _batch(2, 162)]
    t = _model('train', 2, _provider(batches))
    l0 = t.train_one_epoch(0.001)
    assert np.isfinite(l0) and t.global_step == 2
    path = str(tmp_path / 'f' / 'fcos')
    t.save_weight('latest', path)
    t2 = _model('test', 1)
    t2.load_weight(path + '-2')
    a, b = t.export_params(), t2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert len(t2.test_one_image(imgs.numpy())) == 3
