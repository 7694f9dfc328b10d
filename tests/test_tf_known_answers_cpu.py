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
