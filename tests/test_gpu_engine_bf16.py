"""bf16 engine of the models on refinedet.py's graph engine (RefineDet320, PFPNetR, YOLOv2) against their f32 engine -- which the model tests tie to the
oracles -- on the SAME weights and the SAME batch of 8: loss, and the direction and size of every layer's filter gradient.

What bf16 can and cannot do here (measured, and reproduced WITHOUT the GPU by tests/mock_ops.py in bf16-storage mode, so it is arithmetic, not a kernel
property): these networks are identity-free stacks of conv + batch norm (Darknet-19: 23 in a row; RefineDet: 4-6 per head on top of the VGG trunk), and at random
initialisation every batch norm re-amplifies the 2^-9 rounding of its input.  The loss agrees to 0.1-2 %, the LAST layers' gradients to cosine 0.95-0.99, and the
agreement decays by ~3 % per layer towards the input: YOLOv2 b1 0.52 (mock 0.55), RefineDet conv1_1 0.61 (mock 0.58) with the trunk's gradient norm at 0.75-0.8
of f32.  That is why `compute_dtype` defaults to 'f32' for these classes and why DESIGN.md labels their bf16 throughput as such.  The bounds below are the measured
values with margin: a kernel fault (a zero or mis-scaled gradient, a layer feeding garbage) breaks them, rounding noise does not."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 8


def _build(kind, dtype):
    import odtk
    g = torch.Generator().manual_seed(77)
    if kind == 'yolov2':
        from oracle import yolov2_ref as YR
        cfg = {'mode': 'train', 'is_pretraining': False, 'data_shape': [320, 320, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
               'data_format': 'channels_last', 'batch_size': B, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'nms_score_threshold': 0.5,
               'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'rescore_confidence': False, 'priors': YR.PRIORS, 'verbose': False, 'compute_dtype': dtype, 'seed': 11}
        batch = ((torch.rand(B, 320, 320, 3, generator=g) * 255).round(), YR.synthetic_gt(B, 320, 78, pad=8, max_obj=4))
        m = odtk.YOLOv2(cfg, {'data_shape': [320, 320, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    else:
        from oracle import refinedet_ref as FR
        cfg = {'mode': 'train', 'input_size': 320, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': B,
               'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': dtype, 'seed': 11}
        batch = ((torch.rand(B, 320, 320, 3, generator=g) * 255).round(), FR.synthetic_gt(B, 320, 78, pad=8, max_obj=4))
        cls = odtk.RefineDet320 if kind == 'refinedet' else odtk.PFPNetR
        m = cls(cfg, {'data_shape': [320, 320, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    return m, batch


@pytest.mark.parametrize("kind", ["refinedet", "pfpnet", "yolov2"])
def test_bf16_engine_tracks_f32_engine(kind):
    res = {}
    for dt in ('f32', 'bf16'):
        m, batch = _build(kind, dt)
        m.set_batch(*batch)
        m._step_body()
        torch.cuda.synchronize()
        res[dt] = (float(m._data_loss), {k: m.get_param(k, m.G).double() for k in m.pinfo if k.endswith('.w')})
        del m
    lf, lb = res['f32'][0], res['bf16'][0]
    # (round 4: 6e-2, was 2e-2.  RefineDet320's bf16 loss moves by 2-4 % when numerically EQUIVALENT kernels are swapped -- raster-run halo kernel against the 8-wave kernel
    #  on conv2_x / conv3_x: 1e-4 of the outputs differ by one bf16 ulp, error against f64 identical to 9 digits -- 21.12 / 20.74 / 20.30 against the f32 engine's 21.14
    #  with 0 / 2 / 5 layers on the halo kernel; the split-f32 engine on the very same halo kernels reads 21.141.  The bound has to sit above that spread.)
    # YOLOv2 (one trunk, no VGG front on the halo kernels) keeps the round-3 bounds: 2e-2 on the loss, 1.5 on the norm ratio (round-4 advisory).
    loss_tol, ratio_max = (2e-2, 1.5) if kind == 'yolov2' else (6e-2, 1.8)
    assert abs(lb - lf) <= loss_tol * abs(lf), (lb, lf)
    worst_c, worst_r, worst_last = ('', 1.0), ('', 1.0), ('', 1.0)
    last = ('pred.w',) if kind == 'yolov2' else tuple(f'{h}{l}.{o}.w' for h in ('arm', 'odm') for l in range(1, 5) for o in ('loc', 'conf'))
    for k, gf in res['f32'][1].items():
        gb = res['bf16'][1][k]
        if float(gf.norm()) == 0.0:                           # a head level without a positive anchor in this batch: no gradient in either engine
            assert float(gb.norm()) == 0.0, k
            continue
        c = float((gf * gb).sum() / (gf.norm() * gb.norm() + 1e-30))
        r = float(gb.norm() / gf.norm())
        worst_c = min(worst_c, (k, c), key=lambda t: t[1])
        worst_r = max(worst_r, (k, max(r, 1.0 / r)), key=lambda t: t[1])
        if k in last:
            worst_last = min(worst_last, (k, c), key=lambda t: t[1])
    print(kind, 'loss f32 / bf16', lf, lb, 'worst cosine', worst_c, 'of the output layers', worst_last, 'worst norm ratio', worst_r)
    assert worst_c[1] > 0.4, worst_c                           # measured 0.51 (RefineDet conv3_3), 0.52 (YOLOv2 b1)
    assert worst_last[1] > 0.75, worst_last                    # measured 0.80 (RefineDet odm4.loc), 0.95 (YOLOv2 pred)
    assert worst_r[1] < ratio_max, worst_r                           # measured 1.33 -- 1.53 (RefineDet trunk: bf16 norm 0.65-0.8 of f32; moves with the loss, see above)
