"""CPU: TensorFlow's own known answers (tests/tf_known_answers.py) against every restatement of a TF kernel in this repository:
oracle/nms_ref.cpp, the oracle's brute-force NMS, the shim's independent NMS; SAME padding in the oracle, the shim and odtk.ops;
bilinear resizing in the augmentor oracle, the shim (both grids) and the CPU mock; fused batch norm in the oracle and the shim."""
import numpy as np
import pytest
import torch

import tf_known_answers as K
from oracle import ssd300_ref as R
from oracle import augment_ref as AR
from oracle import tf_shim


@pytest.mark.parametrize("case", K.NMS_CASES, ids=[c[0] for c in K.NMS_CASES])
def test_nms_op_vectors(case):
    name, boxes, scores, max_out, iou, score_thr, want = case
    thr = float('-inf') if score_thr is None else score_thr
    assert R.nms(boxes, scores, max_out, iou, thr).tolist() == want                      # oracle/nms_ref.cpp
    got = tf_shim.image.non_max_suppression(torch.from_numpy(boxes), torch.from_numpy(scores), max_out, iou, thr)
    assert got.tolist() == want                                                          # the shim's own implementation
    if score_thr is None:
        assert R.nms_python(boxes, scores, max_out, iou).tolist() == want               # brute force


def test_nms_three_implementations_agree_on_random_boxes():
    """distinct scores: the heap-based restatement, the brute-force loop and the shim's matrix form must pick the same boxes"""
    g = np.random.default_rng(0)
    for n in (1, 7, 64, 500):
        yx = g.uniform(0, 100, (n, 2)).astype(np.float32); hw = g.uniform(1, 40, (n, 2)).astype(np.float32)
        boxes = np.concatenate([yx - hw / 2, yx + hw / 2], 1)
        scores = (g.permutation(n).astype(np.float32) + 1) / n
        for thr in (0.3, 0.5, 0.7):
            a = R.nms(boxes, scores, n, thr).tolist()
            assert a == R.nms_python(boxes, scores, n, thr).tolist()
            assert a == tf_shim.image.non_max_suppression(torch.from_numpy(boxes), torch.from_numpy(scores), n, thr).tolist()


@pytest.mark.parametrize("args,want", K.SAME_PAD_CASES)
def test_same_padding(args, want):
    import odtk  # noqa: F401
    from odtk import ops
    assert R.same_pad(*args) == want
    assert ops.same_pad(*args) == want
    i, k, s, d = args
    assert tf_shim._same_pad(i, k, s, d) == want[1:]


def test_resize_bilinear_legacy_grid():
    x, want = torch.from_numpy(K.RESIZE_LEGACY_IN), torch.from_numpy(K.RESIZE_LEGACY_OUT)
    assert torch.equal(AR.resize_bilinear_legacy(x, 6, 4), want)
    assert torch.equal(tf_shim.image.resize_bilinear(x[None], [6, 4])[0], want)
    import mock_ops
    assert torch.equal(mock_ops._bilinear(x[None], 6, 4)[0], want)
    from oracle import retinanet_net_ref, fcos_net_ref
    for mod in (retinanet_net_ref, fcos_net_ref):
        assert torch.equal(mod._resize(x.permute(2, 0, 1)[None], 6, 4)[0].permute(1, 2, 0), want)


def test_resize_bilinear_align_corners():
    x, want = torch.from_numpy(K.RESIZE_ALIGN_IN), torch.from_numpy(K.RESIZE_ALIGN_OUT)
    assert torch.allclose(AR.resize_bilinear_align(x, 5, 4), want, atol=1e-6)
    assert torch.allclose(tf_shim.image.resize_images(x, [5, 4], 'bilinear', align_corners=True), want, atol=1e-6)


def test_resize_nearest_align_corners():
    x, want = torch.from_numpy(K.RESIZE_ALIGN_IN), torch.from_numpy(K.RESIZE_ALIGN_NEAREST_OUT)
    assert torch.equal(AR.resize_nearest_align(x, 5, 4), want)
    assert torch.equal(tf_shim.image.resize_images(x, [5, 4], 'nearest', align_corners=True), want)


def test_resize_bicubic_table_kernel():
    """TF's testResizeUpBicubic (align_corners=False): all 64 table values are met to the rounding of the table (0.5), by the oracle's
    gather form and by the shim's matrix form; the two forms agree to 1e-4 with each other, also with align_corners=True"""
    x, want = torch.from_numpy(K.RESIZE_BICUBIC_IN), torch.from_numpy(K.RESIZE_BICUBIC_OUT)
    a = AR.resize_bicubic_align(x, 8, 8, align_corners=False)
    b = tf_shim.image.resize_images(x, [8, 8], 'bicubic', align_corners=False)
    assert float((a - want).abs().max()) <= 0.5 and float((b - want).abs().max()) <= 0.5
    assert float((a - b).abs().max()) <= 1e-4
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(37, 50, 3, generator=g) * 255).round()
    for (oh, ow) in ((44, 61), (20, 33), (37, 50), (1, 7)):
        d = (AR.resize_bicubic_align(img, oh, ow) - tf_shim.image.resize_images(img, [oh, ow], 'bicubic', align_corners=True)).abs().max()
        assert float(d) <= 2e-4, (oh, ow, float(d))
        assert torch.equal(AR.resize_nearest_align(img, oh, ow), tf_shim.image.resize_images(img, [oh, ow], 'nearest', align_corners=True))
    assert torch.equal(AR.resize_bicubic_align(img, 37, 50), img)            # same size: delta 0 -> weights (0, 1, 0, 0)


def _to_u8(y):
    """convert_image_dtype(float -> uint8, saturate=True): trunc(clip(y) * 255.5), y in 0..1"""
    return torch.floor(y.clamp(0., 1.) * 255.5).clamp(max=255.)


def test_adjust_hue_tables():
    x = torch.from_numpy(K.HUE_IN)
    for delta, want in K.HUE_CASES:
        for fn in (AR.adjust_hue, tf_shim.image.adjust_hue):
            assert torch.equal(_to_u8(fn(x / 255., delta)), torch.from_numpy(want)), (delta, fn)
            # scale-free in the value range (the augmentor works on 0..255 pictures): same table within the truncation of the conversion
            assert float((fn(x, delta) - torch.from_numpy(want)).abs().max()) <= 0.5 + 1e-3


def test_adjust_contrast_tables():
    x, want = torch.from_numpy(K.CONTRAST_IN), torch.from_numpy(K.CONTRAST_OUT)
    got_shim = tf_shim.image.adjust_contrast(x, K.CONTRAST_FACTOR)
    mean = x.mean(dim=(0, 1), keepdim=True)                               # oracle/augment_ref.augment_image's expression
    got_oracle = (x - mean) * K.CONTRAST_FACTOR + mean
    assert torch.allclose(got_shim, want, atol=1e-4) and torch.allclose(got_oracle, want, atol=1e-4)
    assert torch.equal(_to_u8(got_oracle / 255.), torch.from_numpy(K.CONTRAST_OUT_U8))


def test_rotate_quarter_turn_tables():
    import math
    for n, want in ((6, K.ROTATE_EVEN_OUT), (5, K.ROTATE_ODD_OUT)):
        img = torch.arange(n * n, dtype=torch.float32).view(n, n, 1)
        assert torch.allclose(AR.rotate_bilinear(img, math.pi / 2)[..., 0], torch.from_numpy(want), atol=1e-4)
        assert torch.allclose(tf_shim.contrib.image.rotate(img, math.pi / 2, 'BILINEAR')[..., 0], torch.from_numpy(want), atol=1e-4)


def test_crop_and_resize_tables():
    from oracle import lhrcnn_ref as LR
    img = torch.from_numpy(K.CROP_IN)
    for box, crop, want in K.CROP_CASES:
        b = torch.tensor([box], dtype=torch.float32)
        want = torch.tensor(want, dtype=torch.float32)
        assert torch.equal(LR.crop_and_resize(img, b, torch.zeros(1), crop).flatten(), want), box
        assert torch.equal(tf_shim.image.crop_and_resize(img, b, torch.zeros(1, dtype=torch.int32), [crop, crop]).flatten(), want), box


def test_conv2d_orientation_and_filter_layout():
    """TensorFlow's conv2d is a cross-correlation over an HWIO filter: conv_ops_test.py's 2 x 2 table on the oracle, the shim and the mocked launch (the golden
    fixtures cannot pin this: the shim that produced them defines its convolution through the same torch call)"""
    import mock_ops
    import odtk  # noqa: F401
    from odtk import ops
    x, f, want = torch.from_numpy(K.CONV_IN), torch.from_numpy(K.CONV_FILTER_HWIO), torch.from_numpy(K.CONV_VALID_OUT)
    w_krsc = f.permute(3, 0, 1, 2).contiguous()
    assert torch.equal(R.conv2d_same(x.permute(0, 3, 1, 2), w_krsc, None)[:, :, :1, :2].permute(0, 2, 3, 1), want)
    assert torch.equal(tf_shim.nn.conv2d(x, f, [1, 1, 1, 1], 'SAME')[:, :1, :2], want)
    d = ops.conv_desc(1, 2, 3, 4, 4, 3, 4, 2, 1, 1, ops.F32, ops.F32)
    rows = torch.zeros(6, 4); rows[:, :3] = x.view(6, 3)
    wp = torch.zeros(3, 2, 2, 4); wp[..., :3] = w_krsc
    y = torch.zeros(6, 4)
    mock_ops.conv2d_fwd(d, rows, wp.reshape(-1), None, y, False)
    assert torch.equal(y.view(2, 3, 4)[:1, :2, :3], want[0])


def test_momentum_optimizer_update_rule():
    """momentum_test.py's doBasic on the mocked fused optimizer launch (weight decay 0, gradient scale 1) and on the shim's optimizer (through a one-variable model)"""
    import mock_ops
    p_, m_, g_ = torch.from_numpy(K.MOMENTUM_VAR0).clone(), torch.zeros(2), torch.from_numpy(K.MOMENTUM_GRAD).clone()
    for step in range(2):
        mock_ops.sgd_momentum(p_, m_, g_, K.MOMENTUM_LR, K.MOMENTUM_M, 0.0, 1.0, torch.zeros(4), None)
        np.testing.assert_allclose(p_.numpy(), K.MOMENTUM_AFTER[step], rtol=1e-6)
        np.testing.assert_allclose(m_.numpy(), K.MOMENTUM_ACCUM[step], rtol=1e-6)
    tf_shim.reset()
    v = tf_shim.get_variable('v0', initializer=torch.from_numpy(K.MOMENTUM_VAR0).clone())
    opt = tf_shim.train.MomentumOptimizer(K.MOMENTUM_LR, K.MOMENTUM_M)
    for step in range(2):
        tf_shim.S.pending = []
        opt.minimize((v * torch.from_numpy(K.MOMENTUM_GRAD)).sum())
        tf_shim._flush(True)
        np.testing.assert_allclose(v.detach().numpy(), K.MOMENTUM_AFTER[step], rtol=1e-6)
    tf_shim.reset()


def test_pooling_same_padding_tables():
    import mock_ops
    x, want = torch.from_numpy(K.MAXPOOL_SAME_IN), torch.from_numpy(K.MAXPOOL_SAME_OUT)
    assert torch.equal(R.maxpool_same(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1), want)
    assert torch.equal(tf_shim.layers.max_pooling2d(x, 2, 2, 'same', 'channels_last'), want)
    rows, y = torch.zeros(6, 4), torch.zeros(2, 4)
    rows[:, :3] = x.view(6, 3)
    mock_ops.maxpool_fwd(rows, y, 1, 2, 3, 3, 4, 1, 2, 2, 2, 0, 0)
    assert torch.equal(y[:, :3], want.view(2, 3))
    xa, wa = torch.from_numpy(K.AVGPOOL_SAME_IN), torch.from_numpy(K.AVGPOOL_SAME_OUT)
    assert torch.equal(tf_shim.layers.average_pooling2d(xa, 2, 2, 'same', 'channels_last'), wa)
    rows, y = xa.view(8, 3).clone(), torch.zeros(2, 3)
    mock_ops.avgpool2x2_fwd(rows, y, 1, 2, 4, 3)
    assert torch.equal(y, wa.view(2, 3))


def test_fused_batch_norm_training_statistics():
    e = K.BN_EXPECT
    x = torch.from_numpy(K.BN_X)                                    # NHWC [2,1,1,1]
    # oracle: NCHW
    p = {'l.gamma': torch.ones(1), 'l.beta': torch.zeros(1)}
    stats = {}
    y = R.batch_norm(x.permute(0, 3, 1, 2), p, 'l', True, stats)
    assert torch.allclose(y.reshape(-1), torch.tensor(e['y'], dtype=torch.float32), atol=1e-6)
    assert abs(float(stats['l'][0]) - e['mean']) < 1e-6 and abs(float(stats['l'][1]) - e['var_unbiased']) < 1e-6
    assert R.BN_EPS == 1e-3 and R.BN_MOMENTUM == 0.99
    # shim: tf.layers.batch_normalization(training=True) + the update ops
    tf_shim.reset(); tf_shim._SCOPE_COUNT.clear()
    tf_shim.S.pending = []
    ys = tf_shim.layers.batch_normalization(tf_shim.wrap(x), axis=3, training=True)
    assert torch.allclose(ys.reshape(-1), torch.tensor(e['y'], dtype=torch.float32), atol=1e-6)
    upd = {id(t): v for kind, t, v in tf_shim.S.pending}
    V = tf_shim.S.variables
    assert abs(float(upd[id(V['batch_normalization/moving_mean'])]) - e['moving_mean']) < 1e-6
    assert abs(float(upd[id(V['batch_normalization/moving_variance'])]) - e['moving_var']) < 1e-6
    tf_shim.reset()
