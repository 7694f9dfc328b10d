"""GPU parity of Light-Head R-CNN (csrc/lhrcnn.hip, odtk.LHRCNN) through the C-ABI:
  * depthwise convolution forward / input gradient / filter gradient against torch's grouped convolution (3x3, 1x15, 15x1; f32 and bf16 storage);
  * crop_and_resize forward / image gradient against oracle/lhrcnn_ref.crop_and_resize (boxes partly outside the picture, empty rows);
  * the RPN loss chain (match -> NMS x 2 -> loss) against oracle.rpn_one_image: index lists bit-exact, loss 1e-5, gradients 1e-5, the R-CNN slots;
  * the R-CNN loss against torch;
  * the whole class: two training steps against oracle.train_step and the detections of the reference's own class (tests/golden/lhrcnn_*.npz).
Tolerances: f32 kernels 1e-5 relative (sums in a different order), bf16 storage 2^-8 relative; whole model: losses 2e-4, every variable's update in direction (cosine > 0.999) and length (1 %), entries to 0.25 of the largest (ReLU flips of the dense layer; 0.058 measured)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import lhrcnn_ref as LR  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEV = 'cuda:0'


def _ops():
    import odtk  # noqa: F401
    from odtk import ops
    return ops


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max() / (b.float().abs().max() + 1e-30))


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('kh,kw,C,H,W', [(3, 3, 144, 20, 26), (1, 15, 576, 10, 13), (15, 1, 256, 10, 13), (3, 3, 24, 7, 5), (3, 3, 23, 9, 6), (5, 3, 1028, 6, 7)])
def test_depthwise_kernels(kh, kw, C, H, W, dt):
    ops = _ops()
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(kh * 100 + kw + C)
    N, ld = 2, ops.pad_to(C, 8) + 8
    x = torch.zeros(N * H * W, ld, dtype=dtype)
    x[:, :C] = torch.randn(N * H * W, C, generator=g).to(dtype)
    f = torch.randn(kh, kw, C, generator=g) * 0.3
    dy = torch.zeros(N * H * W, ld, dtype=dtype)
    dy[:, :C] = torch.randn(N * H * W, C, generator=g).to(dtype)
    xin = x[:, :C].float().reshape(N, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
    w = f.permute(2, 0, 1).unsqueeze(1).clone().requires_grad_(True)
    ref = F.conv2d(F.pad(xin, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2)), w, None, groups=C)
    gx, gw = torch.autograd.grad(ref, [xin, w], dy[:, :C].float().reshape(N, H, W, C).permute(0, 3, 1, 2))
    xd, fd, dyd = x.to(DEV), f.to(DEV), dy.to(DEV)
    y = torch.full((N * H * W, ld), 7.0, dtype=dtype, device=DEV)
    ops.depthwise_conv(xd, ld, fd, y, ld, N, H, W, C, kh, kw)
    dx = torch.zeros(N * H * W, ld, dtype=dtype, device=DEV)
    ops.depthwise_conv(dyd, ld, fd, dx, ld, N, H, W, C, kh, kw, True, False)
    dx2 = dx.clone()
    ops.depthwise_conv(dyd, ld, fd, dx2, ld, N, H, W, C, kh, kw, True, True)          # accumulate
    df = torch.zeros(kh, kw, C, device=DEV)
    ops.depthwise_wgrad(xd, ld, dyd, ld, df, N, H, W, C, kh, kw)
    torch.cuda.synchronize()
    tol = 1e-5 if dt == 'f32' else 1e-2
    assert _rel(y[:, :C], ref.detach().permute(0, 2, 3, 1).reshape(-1, C)) < tol
    assert float((y[:, C:].float() - 7.0).abs().max()) == 0.0                          # pad columns untouched
    assert _rel(dx[:, :C], gx.permute(0, 2, 3, 1).reshape(-1, C)) < tol
    assert _rel(dx2[:, :C], 2 * gx.permute(0, 2, 3, 1).reshape(-1, C)) < 2 * tol
    assert _rel(df, gw.squeeze(1).permute(1, 2, 0)) < (1e-4 if dt == 'f32' else 1e-2)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_crop_and_resize_kernels(dt):
    ops = _ops()
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    N, H, W, C, R, crop = 2, 22, 35, 490, 40, 7
    ld = ops.pad_to(C, 8)
    feat = torch.zeros(N * H * W, ld, dtype=dtype)
    feat[:, :C] = torch.randn(N * H * W, C, generator=g).to(dtype)
    y1, x1 = torch.rand(R, generator=g) * 0.7 - 0.1, torch.rand(R, generator=g) * 0.7 - 0.1        # some boxes start / end outside [0, 1]
    boxes = torch.stack([y1, x1, y1 + 0.1 + torch.rand(R, generator=g) * 0.6, x1 + 0.1 + torch.rand(R, generator=g) * 0.6], 1)
    boxes[3] = torch.tensor([0., 0., 1., 1.]); boxes[4] = torch.tensor([0.25, 0.5, 0.25, 0.5])       # the whole picture; a degenerate box
    # proposals clamped to the picture end exactly at 0.0 / 1.0: their last sample row / column sits ON the border, where one ulp decides inside / outside
    edge = torch.rand(12, 4, generator=g)
    edge[:, 2:] = 1.0; edge[:6, :2] *= 0.9; edge[6:, :2] = 0.0; edge[6:, 2:] = 0.3 + 0.7 * torch.rand(6, 2, generator=g)
    boxes[8:20] = edge
    img = torch.randint(0, N, (R,), generator=g).to(torch.int32)
    img[7] = -1; img[R - 1] = -1
    ldo = ops.pad_to(crop * crop * C, 8)
    dout = torch.zeros(R, ldo, dtype=dtype)
    dout[:, : crop * crop * C] = torch.randn(R, crop * crop * C, generator=g).to(dtype)
    f4 = feat[:, :C].float().reshape(N, H, W, C).requires_grad_(True)
    live = (img >= 0).view(-1, 1)
    ref = LR.crop_and_resize(f4, boxes, img.clamp(min=0), crop).reshape(R, -1) * live
    gref, = torch.autograd.grad(ref, f4, dout[:, : crop * crop * C].float())
    out = torch.full((R, ldo), 3.0, dtype=dtype, device=DEV)
    ops.crop_and_resize_fwd(feat.to(DEV), ld, N, H, W, C, boxes.to(DEV), img.to(DEV), crop, out, ldo)
    dfeat = torch.full((N * H * W, ld), 9.0, device=DEV)
    ops.crop_and_resize_bwd(dout.to(DEV), ldo, N, H, W, C, boxes.to(DEV), img.to(DEV), crop, dfeat, ld)
    torch.cuda.synchronize()
    tol = 1e-5 if dt == 'f32' else 1e-2
    assert _rel(out[:, : crop * crop * C], ref.detach()) < tol
    assert float(out[7, : crop * crop * C].float().abs().max()) == 0.0
    assert _rel(dfeat[:, :C], gref.reshape(-1, C)) < 1e-4
    assert float(dfeat[:, C:].abs().max()) == 0.0


def _rpn_inputs(seed, N=2, H=320, W=416, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    fh, fw = -(-H // 32), -(-W // 32)
    A_full = fh * fw * LR.NA
    conf = torch.randn(N, A_full, 2, generator=g) * scale
    bbox = torch.randn(N, A_full, 4, generator=g) * 0.3
    gt = LR.synthetic_gt(N, H, W, seed + 1, pad=7, max_obj=4)
    return conf, bbox, gt, LR.anchors(fh, fw, H, W), A_full


def _anchors_dev(anc, A_full):
    return dict(y1x1=anc['y1x1'].to(DEV), y2x2=anc['y2x2'].to(DEV), yx=anc['yx'].to(DEV), hw=anc['hw'].to(DEV),
                row=torch.nonzero(anc['keep']).flatten().to(torch.int32).to(DEV), A_full=A_full)


@pytest.mark.parametrize('seed,shape', [(11, (320, 416)), (12, (320, 416)), (13, (448, 608))])
def test_rpn_loss_chain_vs_oracle(seed, shape):
    ops = _ops()
    H, W = shape
    N = 2
    conf, bbox, gt, anc, A_full = _rpn_inputs(seed, N, H, W)
    keep = anc['keep']
    ad = _anchors_dev(anc, A_full)
    A = anc['yx'].shape[0]
    ws = ops.lhrcnn_workspace(N, A, gt.shape[1], DEV)
    cd, bd, gd = conf.to(DEV), bbox.to(DEV), gt.to(DEV)
    ops.lhrcnn_match(ad, cd, gd, ws)
    cap = ws['cap']
    counts = ws['counts'].view(-1)
    ops.nms_batched(ws['pos_box'], cap * 4, ws['pos_score'], cap, 1, ws['pos_valid'], cap, 1, 1, cap, N, counts[3:], 8, 0, 0.7, ws['sel_pos'], 128, ws['cnt_pos'])
    ops.nms_batched(ws['neg_box'], cap * 4, ws['neg_score'], cap, 1, ws['neg_valid'], cap, 1, 1, cap, N, counts[4:], 8, 0, 0.7, ws['sel_neg'], 256, ws['cnt_neg'])
    d_conf = torch.full((N, A_full, 2), 5.0, device=DEV)
    d_bbox = torch.full((N, A_full, 4), 5.0, device=DEV)
    ops.lhrcnn_rpn_loss(ad, cd, bd, gd, ws, 21, 1.0 / N, H, W, d_conf, d_bbox)
    torch.cuda.synchronize()
    cf = conf.clone().requires_grad_(True)
    bb = bbox.clone().requires_grad_(True)
    tot = 0.
    lim = torch.tensor([H - 1., W - 1., H - 1., W - 1.])
    for i in range(N):
        loss, pos_prop, pos_lab, truth, neg_prop, d = LR.rpn_one_image(bb[i, keep, :2], bb[i, keep, 2:], cf[i, keep], anc, gt[i], detail=True)
        tot = tot + loss
        c = ws['counts'][i].cpu().tolist()
        n_pos, n_neg = d['pos_a'].shape[0], d['neg_o'].shape[0]
        assert c[:5] == [d['G'], n_pos, n_neg, min(n_pos, 128), min(n_neg, 256 - min(n_pos, 128))], (c, d['G'], n_pos, n_neg)
        assert torch.equal(ws['pos_anchor'][i, :n_pos].cpu().long(), d['pos_a']) and torch.equal(ws['pos_gt'][i, :n_pos].cpu().long(), d['pos_gi'])
        assert torch.equal(ws['neg_anchor'][i, :n_neg].cpu().long(), d['neg_o'])
        kp, kn = d['sel_p'].shape[0], d['sel_n'].shape[0]
        assert int(ws['cnt_pos'][i]) == kp and int(ws['cnt_neg'][i]) == kn
        assert torch.equal(ws['sel_pos'][i, :kp].cpu().long(), d['sel_p']) and torch.equal(ws['sel_neg'][i, :kn].cpu().long(), d['sel_n'])
        assert abs(float(ws['rpn_parts'][i, 3]) - float(loss)) < 1e-5 * abs(float(loss))
        s = i * 256
        assert ws['roi_counts'][i].cpu().tolist() == [kp, kn]
        prop = torch.minimum(torch.clamp(torch.cat([pos_prop, neg_prop]).detach(), min=0.), lim)
        np.testing.assert_allclose(ws['roi_prop'][s: s + kp + kn].cpu().numpy(), prop.numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(ws['roi_box'][s: s + kp + kn].cpu().numpy(), (prop / lim).numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ws['roi_truth'][s: s + kp].cpu().numpy(), truth.detach().numpy(), rtol=1e-4, atol=1e-5)
        assert torch.equal(ws['roi_label'][s: s + kp].cpu().long(), pos_lab.long()) and bool((ws['roi_label'][s + kp: s + kp + kn] == 20).all())
        assert bool((ws['roi_img'][s: s + kp + kn] == i).all()) and bool((ws['roi_img'][s + kp + kn: s + 256] == -1).all())
        assert ws['roi_kind'][s: s + 256].cpu().tolist() == [1] * kp + [2] * kn + [0] * (256 - kp - kn)
    g1, g2 = torch.autograd.grad(tot / N, [cf, bb])
    assert _rel(d_conf, g1) < 1e-5 and _rel(d_bbox, g2) < 1e-5


def test_rcnn_loss_kernel():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    N, C = 3, 21
    ldl, ldb = 24, 4
    ws = ops.lhrcnn_workspace(N, 100, 4, DEV)
    kind = torch.zeros(N * 256, dtype=torch.int32)
    label = torch.full((N * 256,), -1, dtype=torch.int32)
    counts = [(5, 40), (128, 128), (1, 0)]
    for i, (kp, kn) in enumerate(counts):
        kind[i * 256: i * 256 + kp] = 1; kind[i * 256 + kp: i * 256 + kp + kn] = 2
        label[i * 256: i * 256 + kp] = torch.randint(0, 20, (kp,), generator=g).to(torch.int32)
        label[i * 256 + kp: i * 256 + kp + kn] = 20
    truth = torch.randn(N * 256, 4, generator=g)
    logits = torch.zeros(N * 256, ldl); logits[:, :C] = torch.randn(N * 256, C, generator=g) * 2
    pbbox = torch.randn(N * 256, ldb, generator=g) * 1.5
    ws['roi_kind'].copy_(kind); ws['roi_label'].copy_(label); ws['roi_truth'].copy_(truth)
    ws['roi_counts'].copy_(torch.tensor(counts, dtype=torch.int32))
    dl = torch.full((N * 256, ldl), 4.0, device=DEV); db = torch.full((N * 256, ldb), 4.0, device=DEV)
    ops.lhrcnn_rcnn_loss(logits.to(DEV), ldl, pbbox.to(DEV), ldb, N, C, ws, 1.0, dl, db)
    torch.cuda.synchronize()
    z = logits[:, :C].clone().requires_grad_(True); b = pbbox.clone().requires_grad_(True)
    live, pos = kind != 0, kind == 1
    ce = (torch.logsumexp(z, 1) - z.gather(1, label.clamp(min=0).long().view(-1, 1)).squeeze(1))[live].mean()
    box = LR.smooth_l1(b[pos] - truth[pos]).sum(-1).mean()
    g1, g2 = torch.autograd.grad(ce + box, [z, b])
    assert abs(float(ws['rcnn_parts'][:, 0].sum()) - float(ce)) < 1e-5 * float(ce) and abs(float(ws['rcnn_parts'][:, 1].sum()) - float(box)) < 1e-5 * float(box)
    assert _rel(dl[:, :C], g1) < 1e-5 and _rel(db, g2) < 1e-5 and float(dl[:, C:].abs().max()) == 0.0


def _cfg(mode, batch, **kw):
    cfg = {'data_shape': [320, 416, 3], 'mode': mode, 'is_pretraining': False, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
           'keep_prob': 0.5, 'batch_size': batch, 'rpn_first_step': 60000, 'rcnn_first_step': 100000, 'rpn_second_step': 160000, 'nms_score_threshold': 0.5,
           'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'post_nms_proposal': 500, 'verbose': False}
    cfg.update(kw)
    return cfg


@pytest.mark.parametrize('engine', ['f32', 'f32x3'])
def test_training_steps_vs_oracle(engine):
    """two steps from the parameters of the golden fixture: both losses against oracle/lhrcnn_ref.train_step (which the reference's own class pins,
    tests/golden/lhrcnn_train.npz -- checked here as well), every parameter and moving statistic after the first step"""
    import odtk
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(901)
    imgs = (torch.rand(2, 320, 416, 3, generator=g) * 255).round()
    gt = LR.synthetic_gt(2, 320, 416, 911)
    p = LR.init_params(71)
    gold = np.load(os.path.join(GOLD, 'lhrcnn_train.npz'))
    m = odtk.LHRCNN(_cfg('train', 2, rpn_first_step=1, compute_dtype=engine), {'data_shape': [320, 416, 3], 'num_train': 2, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    q = {k: v.clone() for k, v in p.items()}
    mom = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(2):
        loss = float(m.train_step(0.003))
        rpn, rcnn = LR.train_step(q, mom, imgs, gt, 0.003)
        got_rpn, got_rcnn = float(m.last_losses[0]), float(m.last_losses[1])
        # ('f32x3': the FIRST step meets the exact engine's bounds -- losses 2e-4, every update's cosine > 0.999; the second step starts from an update of lr 0.003 that
        #  takes the losses from 9 to 32, and the 2^-17 products move that overshoot by 6-9 %: measured 32.8 / 36.7 against 31.1 / 33.7)
        tol = 2e-4 if step == 0 else (0.15 if engine == 'f32x3' else 5e-2)       # step 2 starts from parameters that agree to ~1e-6: an NMS pick or a 0.5 / 0.3 IoU decision on a near-tie may flip (the
        #                                          reference-vs-oracle fixture saw 1.3 % from one such flip), measured 3 runs: within 5e-3
        # the R-CNN loss gets a looser first-step bound than the RPN loss: crops of proposals clamped to the picture put their last sample row ON the border, where
        # the last bits of the proposal decide inside / extrapolated (tests/test_hip_cpu.py has the measurement); measured here: 2e-4 on both
        assert abs(got_rpn - rpn) < tol * abs(rpn) and abs(got_rcnn - rcnn) < max(tol, 5e-3) * abs(rcnn), (step, got_rpn, rpn, got_rcnn, rcnn)
        assert loss == (got_rpn if step == 0 else got_rcnn)
        if step == 0:
            assert abs(got_rpn - gold['rpn_losses'][0]) < 2e-4 * gold['rpn_losses'][0] and abs(got_rcnn - gold['rcnn_losses'][0]) < 5e-3 * gold['rcnn_losses'][0]
            # the UPDATE of every variable (lr * momentum-accumulated gradient; moving statistics: their 1 % move) against the oracle's, relative to the
            # largest entry of that update -- the bound of the other classes' whole-model tests (batch-norm backward over 2 x 10 x 13 samples amplifies the
            # summation-order differences of the f32 convolutions)
            after = m.export_params()
            errs = {}
            for k in q:
                du, dref = after[k] - p[k], q[k] - p[k]
                if k.endswith('.b') and k[:-2] not in ('roi_feat_dense', 'rcnn_pconf', 'rcnn_pbbox'):
                    # a bias in front of a batch norm: its gradient is round-off noise in TensorFlow / the oracle (< 1e-5), exactly zero here
                    assert float(du.abs().max()) == 0.0 and float(dref.abs().max()) < 1e-5, k
                    continue
                du64, dref64 = du.double(), dref.double()
                cos = float((du64 * dref64).sum() / (du64.norm() * dref64.norm() + 1e-30))
                errs[k] = (_rel(du, dref), cos, float(du64.norm() / (dref64.norm() + 1e-30)))
            report = sorted(errs.items(), key=lambda t: -t[1][0])
            if os.path.isdir('gpurun_out'):
                with open('gpurun_out/lhrcnn_update_errors.txt', 'w') as f:
                    f.write('variable  max|du - du_ref| / max|du_ref|  cosine  norm ratio\n' + '\n'.join(f'{k} {e:.3e} {c:.6f} {r:.5f}' for k, (e, c, r) in report) + '\n')
            # measured on MI355X (profiles/r03zzzz_lhrcnn_update_errors.txt): worst entry 5.8e-2 of the largest (roi_feat_dense.w: a handful of the 2 x 256 x 2048
            # ReLU inputs of the first dense layer sit within the f32 summation noise of zero and flip, each flip moves one row of the filter gradient),
            # direction and length of every update to 1e-3
            assert report[0][1][0] < 0.25, report[:8]            # which inputs flip depends on the atomics' order of that run: bounded loosely, the cosines are the check
            assert min(c for _, (_, c, _) in report) > 0.999 and max(abs(r - 1.) for _, (_, _, r) in report) < 1e-2, sorted(errs.items(), key=lambda t: t[1][1])[:8]
    assert m.global_step == 2


def test_first_step_in_the_second_pinned_configuration():
    """256 x 480, batch 3, 5 classes, up to five objects, weight decay 5e-4, lr 0.002 (tests/golden/lhrcnn_train_b.npz, produced by the reference's own class): both
    losses of the first step against the reference's numbers (2e-4; R-CNN 5e-3, as above), the sub-sampled variables after it to 1e-3 of their largest entry"""
    import json
    import odtk
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lhrcnn_train_b.npz'))
    c = json.loads(str(g['case']))
    gen = torch.Generator().manual_seed(c['seeds'][0])
    gt = LR.synthetic_gt(c['batch'], c['H'], c['W'], c['seeds'][0] + 10, pad=c['pad'], max_obj=c['max_obj'])
    gt[..., 4] = torch.where(gt[..., 4] >= 0, gt[..., 4] % c['num_classes'], gt[..., 4])
    imgs = (torch.rand(c['batch'], c['H'], c['W'], 3, generator=gen) * 255).round()
    p = LR.init_params(c['seed_params'], num_classes=c['num_classes'] + 1)
    shape = [c['H'], c['W'], 3]
    m = odtk.LHRCNN(_cfg('train', c['batch'], data_shape=shape, num_classes=c['num_classes'], weight_decay=c['weight_decay']),
                    {'data_shape': shape, 'num_train': c['batch'], 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    m.train_step(c['lr'])
    got_rpn, got_rcnn = float(m.last_losses[0]), float(m.last_losses[1])
    assert abs(got_rpn - g['rpn_losses'][0]) < 2e-4 * g['rpn_losses'][0] and abs(got_rcnn - g['rcnn_losses'][0]) < 5e-3 * g['rcnn_losses'][0], (got_rpn, got_rcnn)
    after = m.export_params()
    for key in [k for k in g.files if '__' in k]:
        name = key.replace('__', '.')
        flat = after[name].contiguous().reshape(-1).cpu()
        got = flat[::max(1, flat.numel() // 1024)].numpy()
        want = g[key]
        before = p[name].contiguous().reshape(-1)[::max(1, flat.numel() // 1024)].numpy()
        upd, ref_upd = got - before, want - before           # the step's update: compared relative to its own largest entry (ReLU flips of the dense layer: see above)
        assert float(np.abs(upd - ref_upd).max()) <= 0.25 * float(np.abs(ref_upd).max()) + 1e-7, name
    del m
    torch.cuda.empty_cache()


def test_detections_vs_reference_class():
    import odtk
    g = np.load(os.path.join(GOLD, 'lhrcnn_detect.npz'))
    p = LR.init_params(71)
    for k in g.files:
        if k.startswith('stat__'):
            p[k[6:].replace('__', '.')] = torch.from_numpy(g[k])
    m = odtk.LHRCNN(_cfg('test', 1, nms_score_threshold=float(g['score_threshold']), post_nms_proposal=int(g['post_nms_proposal'])), None)
    m.load_oracle_params(p)
    img = torch.from_numpy(g['image']).float() / 127.5 - 1.
    scores, bbox, cid = m.test_one_image(img.numpy())
    assert np.array_equal(cid, g['class_id'])                       # same detections in the same order (135 of them, 13 classes)
    es = float(np.abs(scores - g['scores']).max())
    size = np.maximum(1.0, np.maximum(g['bbox'][:, 2] - g['bbox'][:, 0], g['bbox'][:, 3] - g['bbox'][:, 1]))[:, None]
    eb = float((np.abs(bbox - g['bbox']) / size).max())
    if os.path.isdir('gpurun_out'):
        with open('gpurun_out/lhrcnn_detect_errors.txt', 'w') as f:
            f.write(f'detections {len(scores)}  max |score error| {es:.3e}  max box-coordinate error relative to the longer side of the box {eb:.3e}\n')
    # north_star: boxes / scores within 1e-3 (scores absolute; box coordinates relative to the box's longer side -- this head's random-weight boxes reach 2 000 px;
    # measured: scores 1.2e-4, profiles/r03zzzz_lhrcnn_gpu_tests.md)
    assert es < 1e-3 and eb < 1e-3, (es, eb)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f32x3'])
def test_every_launch_in_situ_at_the_driver_shape(dtype):
    """testlhrcnn.py's shape -- 700 x 1100, batch 32 -- on the GPU: every launch of a whole training step is re-executed in plain f32 PyTorch from the engine's
    own stored inputs of that launch and compared (tests/insitu.py, as for the other classes in tests/test_gpu_insitu_configs.py): 27 convolutions / dense
    layers forward, filter and input gradients, 24 batch norms, 32 + 17 depthwise launches, the crop and its gradient over 8 192 rows, the RPN loss against
    the oracle on the engine's own predictions (CPU), the R-CNN loss, both momentum launches.  The toy-shape cases above do not see the tile counts, pixel
    splits and 32-bit offsets of this size."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import insitu
    import odtk
    torch.set_num_threads(16)
    H, W, B = 700, 1100, 32
    g = torch.Generator().manual_seed(7)
    imgs = (torch.rand(B, H, W, 3, generator=g) * 255).round()
    gt = LR.synthetic_gt(B, H, W, 8, pad=60, max_obj=6)
    sh = insitu.Shadow()
    with sh.installed():
        m = odtk.LHRCNN(_cfg('train', B, data_shape=[H, W, 3], compute_dtype=dtype), {'data_shape': [H, W, 3], 'num_train': B, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
        m.set_batch(imgs, gt)
        m.train_step(0.003)                      # un-shadowed first step: lazily grown scratch exists, momentum / moving statistics are non-trivial
        sh.recording = True
        loss = m.train_step(0.003)
        sh.recording = False
        torch.cuda.synchronize()
    assert bool(torch.isfinite(loss).all())
    base_tol = insitu.default_tol(dtype)

    def tol(row):
        # the oracle recomputes the NMS scores from the engine's logits with another exp / log: two overlapping candidates whose scores differ in the last bit
        # may swap, which moves two of a picture's 256 gradient rows (relative error sqrt(2 / (32 * 256)) = 1.6e-2 per swap); measured: no swap, 6.6e-8
        if row['op'] == 'lhrcnn_rpn_loss':
            return 5e-2
        # bf16 engine: x / 127.5 - 1 of an integer pixel value is rounded to bf16 by the kernel and by the restatement from f32 values that may differ in the last
        # f32 bit (division + subtraction vs torch's own order): where such a value sits on a bf16 rounding tie the two stores differ by ONE bf16 ulp, 2^-7 at
        # |x| ~ 1 relative to the largest element (measured on MI355X, round 4: exactly 7.8125e-3 on 24.6 M pixels); a wrong pixel would be O(1)
        if row['op'] == 'preprocess_norm' and dtype == 'bf16':
            return 8e-3
        return base_tol(row)
    rows = sh.check(tol, verbose=True, label=f'lhrcnn {dtype} {H}x{W} batch {B}')
    seen = {x['op'] for x in rows}
    assert {'conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad', 'depthwise_conv', 'depthwise_wgrad', 'crop_and_resize_fwd', 'crop_and_resize_bwd', 'lhrcnn_rpn_loss',
            'lhrcnn_rcnn_loss', 'bn_fwd', 'bn_bwd', 'sgd_momentum'} <= seen
    assert sum(1 for x in rows if x['op'] == 'depthwise_conv') == 32 and sum(1 for x in rows if x['op'] == 'conv2d_wgrad' and x['out'] == 'dw') == 27
    if os.path.isdir('gpurun_out'):
        per = {}
        for r_ in rows:
            a = per.setdefault(r_['op'] + ':' + r_['out'].split('[')[0], [0, 0.0])
            a[0] += 1; a[1] = max(a[1], r_['rel'])
        with open(f'gpurun_out/lhrcnn_insitu_{dtype}.txt', 'w') as f:
            f.write(f'lhrcnn {dtype} {H}x{W} batch {B}: {len(rows)} outputs of {sh.seq} launches\n' + '\n'.join(f'{k} x{c} worst {w:.3e}' for k, (c, w) in sorted(per.items(), key=lambda kv: -kv[1][1])) + '\n')
    del m
    torch.cuda.empty_cache()
