"""GPU end-to-end parity of the SSD300 class surface against the CPU oracle (through libodtk)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ssd300_ref as R  # noqa: E402

CONFIG = {
    'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
    'keep_prob': 0.5, 'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20,
    'nms_iou_threshold': 0.5, 'pretraining_weight': './vgg_16.ckpt', 'verbose': False,
}


def _model(mode, dtype, batch, provider=None):
    import odtk
    cfg = dict(CONFIG, mode=mode, compute_dtype=dtype, batch_size=batch)
    return odtk.SSD300(cfg, provider)


def _provider(batch, n_batches=2, seed=0):
    data = [R.synthetic_batch(batch, seed + i) for i in range(n_batches)]
    return {'data_shape': [300, 300, 3], 'num_train': batch * n_batches, 'num_val': 0,
            'train_generator': data, 'val_generator': None}, data


@pytest.mark.parametrize("engine", ["f32", "f32x3"])
def test_inference_parity_f32(engine, dev):
    torch.set_num_threads(16)
    p = R.init_params(3)
    imgs, _ = R.synthetic_batch(2, 7)
    R.calibrate_bn(p, imgs, subtract_mean=False)        # test mode feeds raw pixels (reference quirk)
    m = _model('test', engine, 1)
    m.load_oracle_params(p)
    # head logits
    m.images.copy_(imgs[:1]); m._forward(False, subtract_mean=False)
    with torch.no_grad():
        taps = {}
        pred_ref = R.forward(p, imgs[:1], False, taps=taps, subtract_mean=False)
    torch.cuda.synchronize()
    pred = m.pred.cpu()
    for name in ['conv1_1', 'conv2_2', 'conv4_3', 'conv5_3', 'pool5', 'conv7', 'conv9_2', 'conv11_2', 'feat1']:
        got = m.acts[name].t.float().cpu()[:, : m.acts[name].C].reshape(taps[name].shape)
        err = float((got - taps[name]).abs().max()) / (float(taps[name].abs().max()) + 1e-9)
        assert err < (2e-4 if engine == 'f32' else 1e-3), (name, err)      # (f32x3: 2^-17 per product, measured 1e-4 .. 4e-4 through 20 layers)
    assert float((pred - pred_ref).abs().max()) < (1e-3 if engine == 'f32' else 3e-3) * max(1.0, float(pred_ref.abs().max()))
    # detections: lower the score threshold so that something is reported for random weights
    for thr in (0.5, 0.2):
        m.nms_score_threshold = thr
        s, b, c = m.test_one_image(imgs[:1].numpy())
        s_ref, b_ref, c_ref = R.test_one_image(p, imgs[:1], thr, 20, 0.5)
        assert c.tolist() == c_ref.tolist()
        if len(s_ref):
            # north_star: "boxes/scores within 1e-3 of the TF1.13 reference".  Scores are probabilities: 1e-3 ABSOLUTE (held here: 5e-4; measured 2e-5 on the
            # exact engine, 4e-5 on f32x3).  Boxes are pixel coordinates of a 300-pixel image; two readings, both asserted, measured values printed:
            #   * absolute, against 1e-3 of the IMAGE size = 0.3 px: exact f32 engine bound 0.1 px (measured 0.04 px); f32x3 bound 0.6 px (measured 0.40 px on a
            #     437-pixel box that extends far beyond the image, 0.05 px on boxes inside it) -- f32x3 is 1.3e-3 of the image on that one box;
            #   * relative to the BOX's own size sqrt(h w), boxes of >= 16 px (random weights also emit degenerate 0..2-pixel boxes): measured 2.0e-3 (f32) /
            #     1.8e-3 (f32x3), bound 3e-3 -- ABOVE 1e-3, and not a kernel property: h = prior_h * exp(z), so a box is as accurate as its logit, and two f32
            #     computations (this engine's MFMA order, the oracle's MKL order) differ by up to 1e-3 of the largest logit after 20 layers + batch norm
            #     (asserted above).  A bit-compatible summation order with TF-1.13's Eigen kernels is not available to any re-implementation.
            size = np.sqrt(np.maximum((b_ref[:, 2] - b_ref[:, 0]) * (b_ref[:, 3] - b_ref[:, 1]), 0.0))
            big = size >= 16.0
            rel = float((np.abs(b - b_ref).max(axis=1)[big] / size[big]).max()) if big.any() else 0.0
            print(f'{engine} thr {thr}: {len(s_ref)} detections, scores {float(np.abs(s - s_ref).max()):.2e}, boxes {float(np.abs(b - b_ref).max()):.3f} px, '
                  f'{rel:.2e} of the box size over the {int(big.sum())} boxes of >= 16 px')
            assert float(np.abs(s - s_ref).max()) < 5e-4, float(np.abs(s - s_ref).max())
            assert float(np.abs(b - b_ref).max()) < (0.1 if engine == 'f32' else 0.6), float(np.abs(b - b_ref).max())
            assert rel < 3e-3, rel
    assert s.dtype == np.float32 and b.shape[1] == 4 and c.dtype == np.int32


def test_smoke_bounds_hold_and_catch_a_dropped_tap(dev):
    """__graft_entry__.smoke(): the bf16 engine's gradients against what bf16 storage alone costs in the CPU mock, parameter by parameter (none 0.08 below, at
    most three 0.03 below: smoke_check's docstring has the measurements) -- and a defect of the size the round-4 review named (one of the nine taps of an
    input-gradient filter dropped: conv4_2, and a layer above and below it) must FAIL those bounds, in both dispatch modes of the halo kernel's small tiles."""
    import __graft_entry__ as G
    from odtk import ops
    for bits in (0, 16384):
        ops.debug_set(6, bits)
        try:
            print(G.smoke_check(G.smoke_metrics()))
            for layer in ('conv4_2',) if bits else ('conv4_2', 'conv2_2', 'conv5_2'):
                broken = G.smoke_metrics(break_layer=layer)
                with pytest.raises(AssertionError) as e:
                    G.smoke_check(broken)
                print(f'bits {bits}: a dropped tap in {layer} dgrad is reported as:', str(e.value)[:200])
        finally:
            ops.debug_set(6, 0)


def test_bf16_train_step_tracks_f32_within_mock_bounds(dev):
    """The end-to-end bound of the benchmarked engine at a small batch (round 5; it replaces 'head error < 0.4, trunk cosine > 0.4'): bf16 engine against the f32
    engine on identical weights / images at batch 8, EVERY gradient's cosine within 0.04 of the CPU mock of bf16 storage
    (tests/golden/ssd300_bf16_mock_small.json :: test_train_step_parity, tools/calib_bf16_mock_small.py), norm ratios within 0.12, loss within 2 %."""
    import json
    import os
    import odtk
    mock = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ssd300_bf16_mock_small.json')))['test_train_step_parity']
    B = mock['batch']
    p = R.init_params(mock['param_seed'])
    imgs, gt = R.synthetic_batch(B, mock['data_seed'])
    prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None}
    g, loss = {}, {}
    for dt in ('f32', 'bf16'):
        m = odtk.SSD300(dict(CONFIG, compute_dtype=dt, batch_size=B, use_graph=False), prov)
        m.load_oracle_params(p)
        m.set_batch(imgs, gt)
        m._step_front(); m._backward()
        torch.cuda.synchronize()
        loss[dt] = float(m.loss_parts[:, 3].sum().item()) / B
        g[dt] = {n: (m.param(n, m.G)[..., : m.convs[n[:-2]].cin] if n.endswith('.w') else m.param(n, m.G)).float().cpu().clone() for n in m.pinfo}
    assert abs(loss['bf16'] - loss['f32']) <= 2e-2 * abs(loss['f32']), loss
    assert abs(loss['f32'] - mock['loss'][0]) <= 2e-3 * abs(mock['loss'][0]), (loss, mock['loss'])       # the mock's f32 arm IS the oracle's arithmetic
    gap = 0.0
    for k, (mc, mr) in mock['gradient'].items():
        a, b = g['bf16'][k], g['f32'][k]
        c = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)); r = float(a.norm() / (b.norm() + 1e-30))
        gap = max(gap, mc - c)
        assert c >= mc - 0.04, (k, c, mc)
        assert abs(r - mr) < 0.12, (k, r, mr)
    print('largest cosine shortfall against the mock', gap)


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-3), ("bf16", 6e-2)])
def test_train_step_parity(dtype, tol, dev):
    torch.set_num_threads(16)
    B = 2
    p = R.init_params(5)
    prov, data = _provider(B, 1, seed=40)
    m = _model('train', dtype, B, prov)
    m.load_oracle_params(p)
    mom = {k: torch.zeros_like(v) for k, v in p.items()}
    imgs, gt = data[0]
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.01).item())
    torch.cuda.synchronize()
    g_gpu = m.export_params()
    loss_ref, data_ref = R.train_step(p, mom, imgs, gt, 0.01, 1e-4)
    assert abs(loss - loss_ref) <= tol * abs(loss_ref), (loss, loss_ref)
    # one SGD step from identical weights: the update equals -lr * (grad + wd*w) -> compares all
    # gradients.  Frobenius norm: a single ReLU flip of a near-zero pre-activation (values differ by
    # ~1e-5 between summation orders) moves one channel's BN gradient by a few %, so max-abs is not a
    # meaningful bound for a discontinuous graph.
    init = R.init_params(5)
    worst = 0.0
    for k in R.trainable_names(p):
        bn_bias = k.endswith('.b') and (k[:-2] + '.gamma') in p        # exactly-zero gradient (BN removes it)
        if bn_bias:
            continue
        upd_ref = p[k] - init[k]
        upd = g_gpu[k] - init[k]
        err = float((upd - upd_ref).norm()) / (float(upd_ref.norm()) + 1e-20)
        worst = max(worst, err)
        if dtype == 'f32':
            assert err < 3e-2, (k, err)
        else:
            # bf16 at batch 2: BatchNorm over 18..722 samples + ReLU flips make the extra layers chaotic
            # (forward error grows 1.8% -> 13% from conv6 to conv11_2), so only the head gradients are
            # bounded tightly; the trunk must still point the same way.  (This batch-2 case checks the ORACLE side of the bf16 step -- loss, moving
            # statistics, direction; the tight per-parameter bounds are test_bf16_train_step_tracks_f32_within_mock_bounds above, at batch 8 against the mock.)
            cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm() + 1e-20))
            if k.startswith('pred') and k.endswith('.w'):
                assert err < 0.4, (k, err)
            elif k.endswith('.w'):
                assert cos > 0.4, (k, cos)
    print('worst relative (Frobenius) update error', dtype, worst)
    for k in p:
        if k.endswith('.mmean') or k.endswith('.mvar'):
            assert float((g_gpu[k] - p[k]).abs().max()) <= (1e-3 if dtype == 'f32' else 3e-2) * (float(p[k].abs().max()) + 0.05)


def test_train_one_epoch_and_checkpoint(dev, tmp_path):
    B = 2
    prov, data = _provider(B, 2, seed=50)
    m = _model('train', 'bf16', B, prov)
    l0 = m.train_one_epoch(0.001)
    l1 = m.train_one_epoch(0.001)
    assert np.isfinite(l0) and np.isfinite(l1)
    assert m.global_step == 4
    path = str(tmp_path / 'ssd' / 'test')
    m.save_weight('latest', path)
    m2 = _model('test', 'bf16', 1)
    m2.load_weight(path + '-4')
    a, b = m.export_params(), m2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a)
    out = m2.test_one_image(data[0][0][:1].numpy())
    assert len(out) == 3


@pytest.mark.parametrize("source", ["numpy", "pinned", "reused_buffer"])
def test_train_one_epoch_prefetch_equals_the_plain_loop(source, dev):
    """Round 6: train_one_epoch copies the NEXT batch's pixels to the device under the running step (side stream, behind the preprocess launch; ssd300.py
    _prefetch_issue).  Same batches, same order, deterministic filter gradients (the default): mean loss and every parameter after the epoch equal those of the
    hand-written loop set_batch -> train_step -> float(loss), bit for bit -- from pageable numpy arrays, from pinned tensors, and from an iterator that hands
    out ONE host buffer refilled in place."""
    B, n = 2, 5
    data = [R.synthetic_batch(B, 60 + i) for i in range(n)]

    class Refill:                                             # yields the same numpy array object every time, refilled in place
        def __iter__(self):
            buf_i, buf_g = np.empty((B, 300, 300, 3), np.float32), np.empty(tuple(data[0][1].shape), np.float32)
            for im, gt in data:
                buf_i[...] = im.numpy()
                buf_g[...] = gt.numpy()
                yield buf_i, buf_g
    if source == 'numpy':
        gen = [(im.numpy(), gt.numpy()) for im, gt in data]
    elif source == 'pinned':
        gen = [(im.pin_memory(), gt.pin_memory()) for im, gt in data]
    else:
        gen = Refill()
    prov = {'data_shape': [300, 300, 3], 'num_train': B * n, 'num_val': 0, 'train_generator': gen, 'val_generator': None}
    m = _model('train', 'bf16', B, prov)
    p0 = m.export_params()
    mean = m.train_one_epoch(0.002)
    torch.cuda.synchronize()
    assert m.global_step == n and m._prefetch_next is None
    ref = _model('train', 'bf16', B, dict(prov, train_generator=[]))
    ref.load_oracle_params(p0)
    losses = []
    for im, gt in data:
        ref.set_batch(im, gt)
        losses.append(float(ref.train_step(0.002).item()))
    assert float(mean) == float(np.mean(losses)), (float(mean), losses)
    assert torch.equal(m.P, ref.P)
    # ... and with the overlap switched off the public method gives the same again
    m2 = _model('train', 'bf16', B, prov)
    m2.config['prefetch_images'] = False
    m2.load_oracle_params(p0)
    assert float(m2.train_one_epoch(0.002)) == float(mean) and torch.equal(m2.P, m.P)


def test_default_engine_warms_up_on_an_f32x3_twin_and_hands_over(dev):
    """Round 6 (the bf16 gate's consequence for SSD300, tests/test_gpu_bf16_gate.py): with NO engine named a training instance is the bf16 engine behind an f32x3
    twin for its first `f32_warmup_steps` steps (300 by default; 2 here).  The warm-up steps ARE f32x3 steps (losses bit-identical to an explicit f32x3 model's:
    deterministic filter gradients), the state moves over bit for bit, and step 3 runs on the bf16 engine."""
    import odtk
    B = 2
    imgs, gt = R.synthetic_batch(B, 91)
    prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None}
    cfg = dict(CONFIG, mode='train', batch_size=B, use_graph=False)
    cfg.pop('compute_dtype', None)
    assert odtk.SSD300(dict(cfg), prov).f32_warmup_steps == 300
    m = odtk.SSD300(dict(cfg, f32_warmup_steps=2), prov)
    ref = odtk.SSD300(dict(cfg, compute_dtype='f32x3'), prov)
    ref.load_oracle_params(m.export_params())
    assert m.DT == odtk.ops.BF16 and m.f32_warmup_steps == 2 and ref.f32_warmup_steps == 0
    m.set_batch(imgs, gt); ref.set_batch(imgs, gt)
    l_ref = [float(ref.train_step(0.002)) for _ in range(2)]
    l0 = float(m.train_step(0.002))
    assert m._twin is not None and m._twin.DT == odtk.ops.F32 and m._twin.CDT == odtk.ops.F32X3 and m.global_step == 1
    l1 = float(m.train_step(0.002))
    assert [l0, l1] == l_ref, ([l0, l1], l_ref)
    assert m._twin is None and m.global_step == 2
    a, b = m.export_params(), ref.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a)
    l2, l2_ref = float(m.train_step(0.002)), float(ref.train_step(0.002))        # bf16 engine against f32x3 from identical weights
    assert np.isfinite(l2) and abs(l2 - l2_ref) <= 5e-2 * abs(l2_ref) and m.global_step == 3


def test_tf_saver_checkpoint_roundtrip_and_pretraining(dev, tmp_path):
    """checkpoint_format='tf': the files tf.train.Saver would leave (SSD300.py:490-504) -- every variable of the reference's
    graph under its name and shape (tests/golden/ssd300_variables.json, collected from the reference's own class), momentum
    slots, global_step -- and back; then slim-style `pretraining_weight` from a V1 checkpoint file (SSD300.py:31)."""
    import json
    import os
    import odtk
    from odtk import tf_checkpoint as T
    from test_tf_checkpoint_cpu import _v1_file
    B = 2
    prov, data = _provider(B, 2, seed=60)
    m = odtk.SSD300(dict(CONFIG, mode='train', compute_dtype='bf16', batch_size=B, checkpoint_format='tf'), prov)
    m.train_one_epoch(0.001)
    path = str(tmp_path / 'tfckpt' / 'model.ckpt')
    m.save_weight('latest', path)
    assert T.latest_checkpoint(str(tmp_path / 'tfckpt')) == path + '-2'
    r = T.NewCheckpointReader(path + '-2')
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ssd300_variables.json')))
    shapes = r.get_variable_to_shape_map()
    for name, info in want.items():
        assert shapes[name] == info['shape'], name
        assert (f'inference/{name}/Momentum' in shapes) == info['trainable'], name
    assert len(shapes) == len(want) + sum(v['trainable'] for v in want.values())
    assert int(r.get_tensor('global_step')) == 2 and r.get_variable_to_dtype_map()['global_step'] == np.int32
    m2 = odtk.SSD300(dict(CONFIG, mode='train', compute_dtype='bf16', batch_size=B, seed=99), prov)
    m2.load_weight(path + '-2')
    a, b = m.export_params(), m2.export_params()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(m.Mom, m2.Mom) and float(m.Mom.abs().max()) > 0 and m2.global_step == 2
    # slim-style initialisation: a V1 file holding vgg_16/convX/convX_Y/{weights,biases}
    g = np.random.default_rng(5)
    vgg = {}
    for n, cin, cout in [('conv1_1', 3, 64), ('conv4_3', 512, 512)]:
        vgg[f'vgg_16/{n.split("_")[0]}/{n}/weights'] = (g.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        vgg[f'vgg_16/{n.split("_")[0]}/{n}/biases'] = g.standard_normal(cout).astype(np.float32)
    fn = str(tmp_path / 'vgg_16.ckpt')
    _v1_file(fn, vgg)
    m3 = odtk.SSD300(dict(CONFIG, mode='test', compute_dtype='bf16', batch_size=1, pretraining_weight=fn), None)
    w = m3.get_param('conv4_3.w').permute(1, 2, 3, 0).numpy()
    assert np.array_equal(w, vgg['vgg_16/conv4/conv4_3/weights']) and np.array_equal(m3.get_param('conv1_1.b').numpy(), vgg['vgg_16/conv1/conv1_1/biases'])


def test_graph_replay_equals_eager_launches(dev):
    """HIP-graph replay of the step (default after two eager steps) == the eagerly launched step.
    f32 path; wgrad accumulates with float atomics, so equality is to rounding, not bitwise."""
    imgs, gt = R.synthetic_batch(2, 31)
    losses = {}
    for use_graph in (False, True):
        import odtk
        cfg = dict(CONFIG, compute_dtype='f32', batch_size=2, use_graph=use_graph, seed=4)
        m = odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': 2, 'num_val': 0, 'train_generator': [],
                              'val_generator': None})
        m.set_batch(imgs, gt)
        ls = [float(m.train_step(0.005).item()) for _ in range(5)]
        assert (m._g_front is not None) == use_graph and (m._g_back is not None) == use_graph
        losses[use_graph] = ls
    # Batch-2 training is chaotic (float-atomic wgrad order + BatchNorm over 18 samples): two EAGER runs already
    # differ by 4e-4 at step 3 and 4e-3 at step 4 (tools/debug_graph.py), so only the first replayed steps are
    # compared tightly.
    a, b = losses[False], losses[True]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(a[1])      # eager in both
    assert abs(a[2] - b[2]) <= 3e-3 * abs(a[2]), (a, b)                                       # first replay
    assert abs(a[3] - b[3]) <= 3e-2 * abs(a[3]), (a, b)
    assert b[-1] < b[0]                                  # it trains


@pytest.mark.parametrize("side_stream", [False, True, "dp"], ids=["one-stream", "wgrad-stream", "dp-bucket-graphs"])
def test_graph_replay_gradients_match_eager_under_allocation_churn(side_stream, dev):
    """lr = 0 keeps the weights fixed, so every step must reproduce the gradient buffer of the first (eager) step
    -- also when it is replayed from HIP graphs while the caller keeps allocating device memory between steps.
    Regression: a hipMemsetAsync captured inside odtk_ssd_loss replayed with a garbage fill value on ROCm 7.2 (the
    library now zero-fills with its own kernel), which scaled the whole gradient by 1e10..1e30 on some host heap layouts."""
    import odtk
    B = 8
    cfg = dict(CONFIG, compute_dtype='bf16', batch_size=B, wgrad_stream=side_stream is True, use_graph=True, seed=0)
    imgs, gt = R.synthetic_batch(B, 5)
    m = odtk.SSD300(cfg, None if False else {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [],
                                              'val_generator': None})
    if side_stream == "dp":
        m.attach_data_parallel(bucket_mb=8)       # world size 1: no collective, but the backward graph is cut per bucket
    m.set_batch(imgs, gt)
    keep = []
    ref = None
    for step in range(7):
        m.train_step(0.0)
        torch.cuda.synchronize()
        g = m.G.clone()
        keep.append(g)                                        # allocation churn: 105 MB per step stays alive
        keep.append(torch.empty(3_000_001, device=dev))
        if ref is None:
            ref = g
            continue
        assert torch.isfinite(g).all(), step
        for name, (off, shape) in m.pinfo.items():
            n = int(np.prod(shape))
            a, b = g[off:off + n], ref[off:off + n]
            assert float((a - b).abs().max()) <= 2e-2 * (float(b.abs().max()) + 1e-12), (step, name)
    assert m._g_front is not None                              # steps 2.. were graph replays
    if side_stream == "dp":
        assert m._g_back is None and len(m._g_back_segs) >= 4 and sum(len(n) for _, n in m._g_back_segs) >= 25


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_head_stream_equals_single_stream(use_graph, dev):
    """config 'tail_stream' (default on): the six heads run on a second stream beside the extra-layer chain, forward and backward,
    with their own batch-norm workspace and split-K scratch slot.  The result must be what the one-stream order computes: same
    loss, same predictions (bit for bit: no atomics in the forward pass), same gradients up to the float-atomic order of wgrad --
    for several steps, so that a missing cross-stream dependency (a head reading a feature map too early, an extras dgrad
    accumulating into a gradient buffer the head has not written yet) has a chance to show."""
    import odtk
    B = 8
    imgs, gt = R.synthetic_batch(B, 17)
    res = {}
    for tail in (False, True):
        cfg = dict(CONFIG, compute_dtype='bf16', batch_size=B, use_graph=use_graph, tail_stream=tail, seed=2)
        m = odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None})
        assert (m._tail is not None) == tail
        m.set_batch(imgs, gt)
        out = []
        for step in range(5):
            loss = float(m.train_step(0.0).item())           # lr 0: identical weights in every step
            torch.cuda.synchronize()
            out.append((loss, m.pred.clone(), m.G.clone()))
        res[tail] = out
        assert (m._g_front is not None) == use_graph
    for (l0, p0, g0), (l1, p1, g1) in zip(res[False], res[True]):
        assert l0 == l1 or abs(l0 - l1) <= 1e-6 * abs(l0)
        assert torch.equal(p0, p1)
        assert float((g0 - g1).abs().max()) <= 2e-2 * float(g0.abs().max())
        for name, (off, shape) in m.pinfo.items():
            n = int(np.prod(shape))
            a, b = g0[off:off + n], g1[off:off + n]
            assert float((a - b).abs().max()) <= 2e-2 * (float(a.abs().max()) + 1e-12), name


@pytest.mark.parametrize("tail", [True, False], ids=["head-stream", "one-stream"])
def test_recorded_launch_list_equals_eager_launches(tail, dev):
    """`use_graph='list'`: from the third step on the step's C-ABI calls (and the stream forks / joins / events between them) are replayed from a recorded
    list of pre-bound argument tuples.  lr = 0 keeps the weights fixed, so every replayed step must reproduce the gradient buffer of the eager steps (to
    the float-atomic order of the filter gradients); with lr > 0 the losses follow the eager run; a new ground-truth shape drops the list."""
    import odtk
    B = 8
    imgs, gt = R.synthetic_batch(B, 5)
    prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    m = odtk.SSD300(dict(CONFIG, compute_dtype='bf16', batch_size=B, use_graph='list', tail_stream=tail, seed=0), prov)
    m.set_batch(imgs, gt)
    ref = None
    for i in range(6):
        junk = torch.empty(1 << (18 + i), device='cuda')         # allocation churn between steps: the list holds raw pointers of persistent buffers only
        m.train_step(0.0)
        torch.cuda.synchronize()
        g = m.G.clone()
        if ref is None:
            ref = g
        else:
            assert float((g - ref).norm() / ref.norm()) < 2e-3, i
        assert (m._cmds is not None) == (i >= 2)
        del junk
    assert len(m._cmds) > 150
    losses = {}
    for mode in (False, 'list'):
        mm = odtk.SSD300(dict(CONFIG, compute_dtype='f32', batch_size=2, use_graph=mode, tail_stream=tail, seed=4), dict(prov, num_train=2))
        mm.set_batch(*R.synthetic_batch(2, 31))
        losses[mode] = [float(mm.train_step(0.005).item()) for _ in range(5)]
    a, b = losses[False], losses['list']
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and abs(a[2] - b[2]) <= 3e-3 * abs(a[2]) and abs(a[3] - b[3]) <= 3e-2 * abs(a[3]), (a, b)
    gt2 = torch.cat([gt, torch.full((B, 4, 5), -1.0)], 1)         # a different pad length: new device buffer -> the list is dropped and re-recorded
    m.set_batch(imgs, gt2)
    assert m._cmds is None
    for _ in range(4):
        m.train_step(0.0)
    torch.cuda.synchronize()
    assert m._cmds is not None and float((m.G - ref).norm() / ref.norm()) < 2e-3


def test_model_instances_share_one_set_of_side_streams(dev):
    """The HIP runtime deals streams onto four hardware queues; a second instance with its OWN side streams got streams that alias the main stream's queue and ran
    2-7 % slower (profiles/r03s_ab_clean_and_queue_aliasing.md).  Every instance of a process uses the same four streams."""
    from odtk import ssd300 as S
    cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 2,
           'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False, 'seed': 0}
    prov = {'data_shape': [300, 300, 3], 'num_train': 2, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    a, b = S.SSD300(cfg, prov), S.SSD300(dict(cfg, compute_dtype='f32'), prov)
    for attr in ('_side', '_tail', '_twg'):
        assert getattr(a, attr) is not None and getattr(a, attr) is getattr(b, attr), attr
    assert len({a._side.cuda_stream, a._tail.cuda_stream, a._twg.cuda_stream, torch.cuda.current_stream().cuda_stream}) == 4
