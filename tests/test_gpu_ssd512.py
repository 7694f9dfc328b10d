"""GPU parity of SSD512 (the SSD300 class with the 512 x 512 variant's tables; 24 912 priors) through the C-ABI against oracle/ssd512_ref.py,
which is pinned on the reference's own SSD512.py (tests/golden/ssd512.npz): priors bit for bit, prior matching bit for bit above the old
16 384-prior limit, NMS on 24 912 boxes incl. the global-memory whole-problem engine, one f32 training step, bf16 step, inference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ssd300_ref as R     # noqa: E402
from oracle import ssd512_ref as R5    # noqa: E402

CONFIG = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 2,
          'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': './vgg_16.ckpt', 'verbose': False}


def _model(mode, dtype, batch, **kw):
    import odtk
    prov = {'data_shape': [512, 512, 3], 'num_train': batch, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    return odtk.SSD512(dict(CONFIG, mode=mode, compute_dtype=dtype, batch_size=batch, **kw), prov if mode == 'train' else None)


def test_priors_and_matching_bit_exact(dev):
    m = _model('train', 'f32', 2)
    g = np.load('tests/golden/ssd512.npz')
    for got, key in zip(m.pri[:4], ('y1x1', 'y2x2', 'yx', 'hw')):
        assert np.array_equal(got.cpu().numpy(), g[key]), key                     # the reference's own _get_abbox
    imgs, gt = R5.synthetic_batch(2, 41)
    m.set_batch(imgs, gt)
    m.m_best = torch.zeros(2, gt.shape[1], dtype=torch.int32, device=dev)
    m._match()
    torch.cuda.synchronize()
    anchors = R5.priors()
    for i in range(2):
        mt = R.match(anchors, gt[i])
        G = mt['G']
        assert int(m.m_ngt[i]) == G and m.m_best[i, :G].cpu().tolist() == mt['best'].tolist()
        st = torch.zeros(24912, dtype=torch.uint8)
        other = torch.nonzero(mt['othermask']).squeeze(1)
        st[other] = torch.where(mt['pos'], torch.tensor(1, dtype=torch.uint8), torch.tensor(2, dtype=torch.uint8))
        assert torch.equal(m.m_status[i].cpu(), st)
        assert torch.equal(m.m_rg[i].cpu().long()[other], mt['rgindex'])
        num_pos = G + int(mt['pos'].sum()); num_neg = int((~mt['pos']).sum())
        assert m.m_counts[i].cpu().tolist()[:3] == [num_pos, num_neg, min(3 * num_pos, num_neg)]


@pytest.mark.parametrize("engine", ["split", "single"])
def test_nms_24912_boxes(engine, dev):
    """odtk_nms_batched above 16 384 boxes per problem: the split path, and the whole-problem engine whose sort runs through global memory"""
    import odtk  # noqa: F401
    from odtk import ops
    ops.debug_set(3, 1 if engine == "single" else 0)
    try:
        n, B = 24912, 2
        g = torch.Generator().manual_seed(12)
        yx = torch.rand(B, n, 2, generator=g) * 512
        hw = torch.rand(B, n, 2, generator=g) * 80 + 5
        boxes = torch.cat([yx - hw / 2, yx + hw / 2], -1).contiguous()
        scores = torch.stack([(torch.randperm(n, generator=g).float() + 1) / n for _ in range(B)])
        valid = (torch.rand(B, n, generator=g) > 0.1).to(torch.uint8) * 2
        for mo_list in ([60, 300], [24912, 5000]):             # the second: the candidate margin runs out -> whole-problem fallback
            cap = max(mo_list)
            out_idx = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
            mo = torch.tensor(mo_list, dtype=torch.int32, device=dev)
            ops.nms_batched(boxes.to(dev), n * 4, scores.to(dev), n, 1, valid.to(dev), n, 1, 2, n, B, mo, 1, 0, 0.7, out_idx, cap, out_cnt)
            torch.cuda.synchronize()
            for b in range(B):
                msk = valid[b] == 2
                ids = torch.nonzero(msk).squeeze(1)
                ref = R.nms(boxes[b][msk].numpy(), scores[b][msk].numpy(), mo_list[b], 0.7)
                ref = ids[torch.from_numpy(ref.astype(np.int64))].tolist()
                assert out_idx[b, : int(out_cnt[b])].cpu().tolist() == ref
    finally:
        ops.debug_set(3, 0)


def test_f32_train_step_matches_oracle(dev):
    torch.set_num_threads(16)
    p = R5.init_params(6)
    init = {k: v.clone() for k, v in p.items()}
    imgs, gt = R5.synthetic_batch(2, 43)
    m = _model('train', 'f32', 2, use_graph=False)
    m.load_oracle_params(p)
    m.set_batch(imgs, gt)
    loss = float(m.train_step(0.01).item())
    torch.cuda.synchronize()
    got = m.export_params()
    mom = {k: torch.zeros_like(p[k]) for k in R5.trainable_names(p)}
    total, _ = R5.train_step(p, mom, imgs, gt, 0.01)
    assert abs(loss - total) <= 2e-3 * abs(total), (loss, total)
    for k in R5.trainable_names(p):
        if k.endswith('.b') and (k[:-2] + '.gamma') in p:
            continue
        step = p[k] - init[k]
        err = float((got[k] - init[k] - step).norm()) / (float(step.norm()) + 1e-20)
        assert err < 3e-2, (k, err)                            # the SSD300 f32 bound (batch-2 batch norm over 8 ... 8192 samples)


def test_bf16_steps_with_graph_replay_and_inference(dev):
    imgs, gt = R5.synthetic_batch(4, 45)
    m = _model('train', 'bf16', 4, use_graph=True)          # (eager launches are the default since round 3; the replay path stays tested)
    m.set_batch(imgs, gt)
    ls = [float(m.train_step(0.003).item()) for _ in range(6)]
    assert all(np.isfinite(ls)) and ls[-1] < ls[0] and m._g_front is not None
    p = R5.init_params(3)
    R5.calibrate_bn(p, imgs[:2], subtract_mean=False)
    t = _model('test', 'f32', 1)
    t.load_oracle_params(p)
    for thr in (0.5, 0.2):
        t.nms_score_threshold = thr
        s, b, c = t.test_one_image(imgs[:1].numpy())
        s_ref, b_ref, c_ref = R5.test_one_image(p, imgs[:1], thr, 20, 0.5)
        assert c.tolist() == c_ref.tolist()
        if len(s_ref):
            assert float(np.abs(s - s_ref).max()) < 1e-3 and float(np.abs(b - b_ref).max()) < 1e-3 * 512
