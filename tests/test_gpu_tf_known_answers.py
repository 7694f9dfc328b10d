"""GPU: TensorFlow's own known answers (tests/tf_known_answers.py) against the HIP kernels, through the C-ABI:
NonMaxSuppression op-test vectors on both NMS engines, the ResizeImagesTest tables on odtk_resize_bilinear_fwd (TF-1.x grid) and on
the augmentor's align_corners=True resize, the fused-batch-norm statistics on odtk_bn_fwd."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tf_known_answers as K  # noqa: E402


def _ops():
    import odtk  # noqa: F401
    from odtk import ops
    return ops


@pytest.mark.parametrize("engine", ["split", "single"])
@pytest.mark.parametrize("case", [c for c in K.NMS_CASES if len(c[2])], ids=[c[0] for c in K.NMS_CASES if len(c[2])])
def test_nms_op_vectors(case, engine, dev):
    name, boxes, scores, max_out, iou, score_thr, want = case
    ops = _ops()
    ops.debug_set(3, 1 if engine == "single" else 0)
    try:
        n = len(scores)
        b = torch.from_numpy(boxes).to(dev).contiguous()
        s = torch.from_numpy(scores).to(dev).contiguous()
        valid = torch.from_numpy((scores > (-np.inf if score_thr is None else score_thr)).astype(np.uint8)).to(dev)   # V3's score threshold
        cap = max(max_out, 1)
        out_idx = torch.full((1, cap), -1, dtype=torch.int32, device=dev)
        out_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.nms_batched(b, 0, s, n, 1, valid, n, 1, 1, n, 1, None, 0, max_out, iou, out_idx, cap, out_cnt)
        torch.cuda.synchronize()
        assert out_idx[0, : int(out_cnt[0])].cpu().tolist() == want
    finally:
        ops.debug_set(3, 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_resize_bilinear_legacy_grid(dt, dev):
    ops = _ops()
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    C = 8
    x = torch.from_numpy(K.RESIZE_LEGACY_IN).repeat(1, 1, C).reshape(6, C).to(dtype).to(dev).contiguous()        # rows = 3 x 2 pixels
    y = torch.zeros(24, C, dtype=dtype, device=dev)
    ops.resize_bilinear_fwd(x, C, y, C, 1, 3, 2, 6, 4, C)
    torch.cuda.synchronize()
    want = torch.from_numpy(K.RESIZE_LEGACY_OUT).reshape(24, 1).repeat(1, C)
    assert torch.equal(y.float().cpu(), want)                     # every table value is exact in bf16


def test_augmentor_resize_align_corners(dev):
    from odtk import augment as A
    img = torch.from_numpy(K.RESIZE_ALIGN_IN).repeat(1, 1, 3).to(dev).contiguous()         # HWC f32, 3 x 2 x 3
    out = A.image_augmentor(img, [3, 2, 3], 'channels_last', output_shape=[5, 4], fill_mode='BILINEAR')
    torch.cuda.synchronize()
    want = torch.from_numpy(K.RESIZE_ALIGN_OUT).repeat(1, 1, 3)
    assert torch.allclose(out.float().cpu().reshape(5, 4, 3), want, atol=1e-5)


def test_augmentor_resize_nearest_align_corners(dev):
    from odtk import augment as A
    img = torch.from_numpy(K.RESIZE_ALIGN_IN).repeat(1, 1, 3).to(dev).contiguous()
    out = A.image_augmentor(img, [3, 2, 3], 'channels_last', output_shape=[5, 4], fill_mode='NEAREST_NEIGHBOR')
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu().reshape(5, 4, 3), torch.from_numpy(K.RESIZE_ALIGN_NEAREST_OUT).repeat(1, 1, 3))


def test_augmentor_hue_contrast_rotate_tables(dev):
    """TensorFlow's AdjustHueTest / AdjustContrastTest / test_rotate_even|odd tables through the augmentor kernels (scripted draws: only
    the wanted colour op fires; the picture is not resized: output_shape = its size)"""
    from odtk import augment as A
    x = torch.from_numpy(K.HUE_IN).to(dev).contiguous()
    for delta, want in K.HUE_CASES:                 # draws: bcs[3] (only hue < 0.5), hue delta
        out = A.image_augmentor(x, [2, 2, 3], 'channels_last', output_shape=[2, 2], color_jitter_prob=0.5, draws=[0.9, 0.9, 0.1, delta])
        torch.cuda.synchronize()
        assert float((out.cpu() - torch.from_numpy(want)).abs().max()) <= 0.5 + 1e-3       # the table is truncated to integers
    out = A.image_augmentor(x, [2, 2, 3], 'channels_last', output_shape=[2, 2], color_jitter_prob=0.5, draws=[0.9, 0.1, 0.9, K.CONTRAST_FACTOR])
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), torch.from_numpy(K.CONTRAST_OUT), atol=1e-3)
    for n, want in ((6, K.ROTATE_EVEN_OUT), (5, K.ROTATE_ODD_OUT)):
        img = torch.arange(n * n, dtype=torch.float32).view(n, n, 1).repeat(1, 1, 3).to(dev).contiguous()
        out = A.image_augmentor(img, [n, n, 3], 'channels_last', output_shape=[n, n], rotate=[1.0, 90., 90.], draws=[0.0, 90.0])
        torch.cuda.synchronize()
        # the reference's pi (3.1415926, image_augmentor.py:236) is 5e-8 short of a quarter turn: source coordinates within 2e-7 px of the integers
        assert torch.allclose(out.cpu(), torch.from_numpy(want).unsqueeze(-1).repeat(1, 1, 3), atol=1e-3)


def test_crop_and_resize_tables(dev):
    """TensorFlow's crop_and_resize_op_test.cc tables through odtk_crop_and_resize_fwd (f32 and bf16 storage: every table value is exact in bf16)"""
    ops = _ops()
    for dtype in (torch.float32, torch.bfloat16):
        feat = torch.zeros(4, 8, dtype=dtype, device=dev)
        feat[:, 0] = torch.from_numpy(K.CROP_IN).reshape(4).to(dtype).to(dev)
        for box, crop, want in K.CROP_CASES:
            out = torch.full((1, 16), 9.0, dtype=dtype, device=dev)
            ops.crop_and_resize_fwd(feat, 8, 1, 2, 2, 1, torch.tensor([box], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), crop, out, 16)
            torch.cuda.synchronize()
            assert out[0, : crop * crop].float().cpu().tolist() == [float(v) for v in want], (box, dtype)


def test_pooling_same_padding_tables(dev):
    """pooling_ops_test.py's SAME tables on odtk_maxpool_fwd (the window that only covers the last column) and odtk_avgpool2x2_fwd"""
    ops = _ops()
    for dtype in (torch.float32, torch.bfloat16):
        ld = 8
        x = torch.zeros(6, ld, dtype=dtype, device=dev)
        x[:, :3] = torch.from_numpy(K.MAXPOOL_SAME_IN).reshape(6, 3).to(dtype).to(dev)
        y = torch.zeros(2, ld, dtype=dtype, device=dev)
        ops.maxpool_fwd(x, y, 1, 2, 3, ld, ld, 1, 2, 2, 2, 0, 0)          # whole 16-byte chunks of channels: the three table channels + zero padding
        xa = torch.zeros(8, ld, dtype=dtype, device=dev)
        xa[:, :3] = torch.from_numpy(K.AVGPOOL_SAME_IN).reshape(8, 3).to(dtype).to(dev)
        ya = torch.zeros(2, ld, dtype=dtype, device=dev)
        ops.avgpool2x2_fwd(xa, ya, 1, 2, 4, ld)
        torch.cuda.synchronize()
        assert torch.equal(y[:, :3].float().cpu(), torch.from_numpy(K.MAXPOOL_SAME_OUT).reshape(2, 3))
        assert torch.equal(ya[:, :3].float().cpu(), torch.from_numpy(K.AVGPOOL_SAME_OUT).reshape(2, 3))       # every table value is exact in bf16


def test_conv2d_orientation_and_filter_layout(dev):
    """conv_ops_test.py's testConv2D2x2Filter table on odtk_conv2d_fwd (f32 engine), the 2 x 2 filter embedded in the centre / bottom-right taps of a 3 x 3
    one (SAME padding of a 3 x 3 filter pads one cell on every side: taps (1..2, 1..2) read x[h + 0..1, w + 0..1])"""
    ops = _ops()
    x, f, want = torch.from_numpy(K.CONV_IN), torch.from_numpy(K.CONV_FILTER_HWIO), torch.from_numpy(K.CONV_VALID_OUT)
    d = ops.conv_desc(1, 2, 3, 4, 4, 3, 4, 3, 1, 1, ops.F32, ops.F32)
    rows = torch.zeros(6, 4); rows[:, :3] = x.view(6, 3)
    w = torch.zeros(3, 3, 3, 4)                       # [K][R][S][C padded to one chunk]
    w[:, 1:, 1:, :3] = f.permute(3, 0, 1, 2)
    y = torch.zeros(6, 4, device=dev)
    ops.conv2d_fwd(d, rows.to(dev), w.reshape(-1).to(dev), None, y, False)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().view(2, 3, 4)[:1, :2, :3], want[0])          # integers below 2^24: exact in f32 whatever the summation order


def test_momentum_optimizer_update_rule(dev):
    """momentum_test.py's doBasic on odtk_sgd_momentum (weight decay 0, gradient scale 1; the buffers padded to one 64-element segment)"""
    ops = _ops()
    p_ = torch.zeros(64, device=dev); m_ = torch.zeros(64, device=dev); g_ = torch.zeros(64, device=dev)
    p_[:2] = torch.from_numpy(K.MOMENTUM_VAR0).to(dev); g_[:2] = torch.from_numpy(K.MOMENTUM_GRAD).to(dev)
    part = torch.zeros(ops.sgd_blocks(64), device=dev)
    for step in range(2):
        ops.sgd_momentum(p_, m_, g_, K.MOMENTUM_LR, K.MOMENTUM_M, 0.0, 1.0, part, None)
        torch.cuda.synchronize()
        np.testing.assert_allclose(p_[:2].cpu().numpy(), K.MOMENTUM_AFTER[step], rtol=1e-6)
        np.testing.assert_allclose(m_[:2].cpu().numpy(), K.MOMENTUM_ACCUM[step], rtol=1e-6)
        assert float(p_[2:].abs().max()) == 0.0


def test_fused_batch_norm_training_statistics(dev):
    ops = _ops()
    e = K.BN_EXPECT
    C, M = 4, 2                                                     # one f32 chunk of identical channels
    z = torch.from_numpy(K.BN_X).reshape(2, 1).repeat(1, C).to(dev).contiguous()
    mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
    sm = torch.empty(C, device=dev); si = torch.empty(C, device=dev)
    ws = torch.zeros(ops.bn_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    y = torch.zeros(M, C, device=dev)
    ops.bn_fwd(z, M, C, C, torch.ones(C, device=dev), torch.zeros(C, device=dev), mm, mv, sm, si, True, False, y, C, M, M * C, ws)
    torch.cuda.synchronize()
    assert torch.allclose(y[:, 0].cpu(), torch.tensor(e['y'], dtype=torch.float32), atol=1e-6)
    assert abs(float(sm[0]) - e['mean']) < 1e-6
    assert abs(float(mm[0]) - e['moving_mean']) < 1e-6 and abs(float(mv[0]) - e['moving_var']) < 1e-6
