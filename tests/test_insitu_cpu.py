"""The in-situ shadow harness (tests/insitu.py) on the CPU: with tests/mock_ops.py standing in for libodtk, every launch of a training step of each model
class is seen, bound to the restatement's signature, cloned, re-executed and compared (all differences exactly 0 here: both sides are the same code) --
and a launch that writes one wrong tile is caught.  The GPU runs of the same harness at the BASELINE shapes are tests/test_gpu_insitu_configs.py."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import insitu      # noqa: E402
import mock_ops    # noqa: E402


def _yolov3():
    import odtk
    from oracle import yolov3_ref as YR
    cfg = {'mode': 'train', 'data_shape': [64, 64, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
           'batch_size': 2, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3, 'nms_score_threshold': 0.5,
           'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'priors': YR.PRIORS_PX, 'verbose': False, 'compute_dtype': 'f32', 'device': 'cpu', 'use_graph': False}
    g = torch.Generator().manual_seed(40)
    imgs = (torch.rand(2, 64, 64, 3, generator=g) * 255).round()
    gt = YR.synthetic_gt(2, 64, 41, max_obj=3)
    return lambda: odtk.YOLOv3(cfg, {'num_train': 2, 'train_generator': [(imgs, gt)], 'val_generator': None, 'num_val': 0}), imgs, gt, 0.01


def _ssd300():
    import odtk
    from oracle import ssd300_ref as R
    cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
           'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False,
           'compute_dtype': 'f32', 'seed': 0, 'use_graph': False, 'device': 'cpu'}
    imgs, gt = R.synthetic_batch(1, 33)
    return (lambda: odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})), imgs, gt, 0.01


def _retinanet():
    import odtk
    from oracle import retinanet_ref as RR
    cfg = {'is_bottleneck': True, 'residual_block_list': [3, 4, 6, 3], 'init_conv_filters': 16, 'mode': 'train', 'is_pretraining': False,
           'data_shape': [128, 128, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'data_format': 'channels_last', 'batch_size': 2,
           'gamma': 2.0, 'alpha': 0.25, 'nms_score_threshold': 0.8, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45, 'verbose': False,
           'compute_dtype': 'f32', 'device': 'cpu'}
    g = torch.Generator().manual_seed(90)
    imgs = (torch.rand(2, 128, 128, 3, generator=g) * 255).round()
    gt = RR.synthetic_gt(2, 128, 91)
    return (lambda: odtk.RetinaNet(cfg, {'num_train': 2, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})), imgs, gt, 0.01


def _fcos():
    import odtk
    from oracle import fcos_ref as FR
    cfg = {'mode': 'train', 'data_shape': [64, 64, 3], 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
           'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45, 'verbose': False, 'device': 'cpu'}
    g = torch.Generator().manual_seed(7)
    imgs = (torch.rand(2, 64, 64, 3, generator=g) * 255).round()
    gt = FR.synthetic_gt(2, 64, 64, 8)
    return (lambda: odtk.FCOS(cfg, {'num_train': 2, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})), imgs, gt, 0.01


def _centernet():
    import odtk
    from oracle import centernet_ref as CR
    cfg = {'mode': 'train', 'input_size': 128, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 2,
           'score_threshold': 0.1, 'top_k_results_output': 10, 'verbose': False, 'compute_dtype': 'f32', 'device': 'cpu'}
    g = torch.Generator().manual_seed(8)
    imgs = (torch.rand(2, 128, 128, 3, generator=g) * 255).round()
    gt = CR.synthetic_gt(2, 128, 9, pad=8, max_obj=4)
    return (lambda: odtk.CenterNet(cfg, {'data_shape': [128, 128, 3], 'num_train': 2, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})), imgs, gt, 1e-3


MODELS = {'yolov3': _yolov3, 'ssd300': _ssd300, 'retinanet': _retinanet, 'fcos': _fcos, 'centernet': _centernet}


@pytest.mark.parametrize('kind', sorted(MODELS))
def test_shadow_sees_every_launch_of_a_training_step(kind):
    torch.set_num_threads(8)
    make, imgs, gt, lr = MODELS[kind]()
    sh = insitu.Shadow()
    with mock_ops.installed(), sh.installed():
        m = make()
        if kind == 'retinanet':
            mock_ops.retina_loss.anchors = m.anc
        m.set_batch(imgs, gt)
        sh.recording = True
        m.train_step(lr)
        sh.recording = False
    rows = sh.check(lambda r: 0.0, verbose=True, label=kind + ' (mock vs mock)')
    ops_seen = {r['op'] for r in rows}
    assert {'conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'} <= ops_seen and sh.seq > 50
    assert any(o.endswith('_loss') for o in ops_seen) and ({'sgd_momentum', 'adam'} & ops_seen)
    assert all(r.get('stray', 0) == 0 for r in rows)


def test_shadow_catches_one_wrong_tile():
    torch.set_num_threads(8)
    make, imgs, gt, lr = _yolov3()
    sh = insitu.Shadow()
    with mock_ops.installed():
        from odtk import ops
        good = ops.conv2d_fwd
        calls = {'n': 0}

        def broken(d, x, w, bias, y, relu):
            good(d, x, w, bias, y, relu)
            calls['n'] += 1
            if calls['n'] == 7:                              # one launch leaves a 16-row tile of its output stale
                y[32:48, : d.K] = 0
        ops.conv2d_fwd = broken
        try:
            with sh.installed():
                m = make()
                m.set_batch(imgs, gt)
                sh.recording = True
                m._step_front() if hasattr(m, '_step_front') else m.train_step(lr)
                sh.recording = False
        finally:
            ops.conv2d_fwd = good
    with pytest.raises(AssertionError, match='out of bound'):
        sh.check(insitu.default_tol('f32'), verbose=False)
    rows, _ = sh.summary()
    worst = rows[0]
    assert worst['op'] == 'conv2d_fwd' and worst['rel'] > 1e-2
    assert sum(1 for r in rows if r['rel'] > 2e-4 and not r['exact']) == 1     # ... and only that launch: its consumers start from the stored (wrong) input
