#!/usr/bin/env python
"""Generates tests/golden/lhrcnn_train.npz, lhrcnn_detect.npz, lhrcnn_variables.json and lhrcnn_names.json by constructing the REFERENCE's own LHRCNN
class (/root/reference/LH_RCNN.py, configuration keys of testlhrcnn.py:20-37 at 320 x 416, batch 2) on the eager TF-1.x shim:
  * train mode, two steps through its session, twice from the same start: once with the driver's schedule (the reported loss is rpn_loss) and once with
    rpn_first_step = 0 (the reported loss is rcnn_loss).  The updates are the same in both runs -- a TF-1.x graph runs both optimizer ops on every step
    whatever the tf.case selects (oracle/lhrcnn_ref.py header, item 1; the shim is eager and does the same) -- which the generator asserts;
  * a subsample of every parameter kind after the FIRST step, moving statistics, global_step;
  * test mode: detections of one (already normalised: the class's feed bypasses `/ 127.5 - 1`) picture from calibrated moving statistics.
tf.gather runs with the GPU kernel's out-of-range behaviour (tf_shim.GATHER_OOB_ZERO): with the CPU kernel LH_RCNN.py:337 aborts the first step.
The parameters of oracle/lhrcnn_ref.init_params(71) are pushed into the shim's variables in creation order.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_lhrcnn.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lhrcnn_ref as LR           # noqa: E402
from oracle import tf_shim                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
H, W = 320, 416
CONFIG = {'data_shape': [H, W, 3], 'mode': 'train', 'is_pretraining': False, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
          'keep_prob': 0.5, 'batch_size': 2, 'rpn_first_step': 60000, 'rcnn_first_step': 100000, 'rpn_second_step': 160000,
          'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'post_nms_proposal': 500}
KEEP = ['conv1.w', 'conv1.gamma', 'stage2_sconv1.w', 'stage2_sconv2.dw', 'stage2_sconv2.w', 'stage3_sconv5.dw', 'stage3_sconv8.beta', 'stage4_sconv1.b',
        'stage4_sconv4.w', 'rpn_conv.w', 'rpn_conf.w', 'rpn_conf.gamma', 'rpn_pbbox.w', 'rpn_pbbox.beta', 'state5_conv1_1.dw', 'state5_conv1_1.w',
        'state5_conv1_2.dw', 'state5_conv2_2.w', 'state5_conv2_2.gamma', 'roi_feat_dense.w', 'roi_feat_dense.b', 'rcnn_pconf.w', 'rcnn_pconf.b', 'rcnn_pbbox.w',
        'conv1.mmean', 'stage3_sconv2.mvar', 'rpn_conf.mmean', 'state5_conv1_2.mvar']
LR_STEP = 0.003
SEED_PARAMS = 71


def batches():
    out = []
    for s in (901, 900):
        g = torch.Generator().manual_seed(s)
        out.append(((torch.rand(2, H, W, 3, generator=g) * 255).round(), LR.synthetic_gt(2, H, W, s + 10)))
    return out


def to_tf(key, v):
    """our layout -> the TensorFlow variable's"""
    if key.endswith('.dw'):
        return v.unsqueeze(-1)                                      # [kh,kw,C] -> [kh,kw,C,1]
    if key.endswith('.w'):
        return v.permute(1, 2, 3, 0) if v.dim() == 4 else v.t()     # [K,R,S,C] -> [R,S,C,K]; dense [units,in] -> [in,units]
    return v


def from_tf(key, v):
    if key.endswith('.dw'):
        return v.squeeze(-1)
    if key.endswith('.w'):
        return v.permute(3, 0, 1, 2) if v.dim() == 4 else v.t()
    return v


def name_map(V):
    """our parameter name -> TensorFlow variable name, by creation order"""
    specs = LR.layer_specs()
    names = list(V)
    kernels = [k for k in names if k.endswith(('/kernel', '/depthwise_kernel'))]            # one per layer, in creation order
    bns = [k[:-len('/gamma')] for k in names if k.endswith('/gamma')]
    assert len(kernels) == len(specs) == 27 and len(bns) == 24, (len(kernels), len(bns))
    tfname = {}
    bi = 0
    for s, kn in zip(specs, kernels):
        name, kind = s[0], s[1]
        scope = kn.rsplit('/', 1)[0]
        assert scope.endswith('/' + name), (name, kn)
        if kind == 'sep':
            tfname[name + '.dw'], tfname[name + '.w'] = scope + '/depthwise_kernel', scope + '/pointwise_kernel'
            assert scope + '/bias' not in V
        else:
            tfname[name + '.w'], tfname[name + '.b'] = scope + '/kernel', scope + '/bias'
        if kind != 'dense':
            for a, b in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
                tfname[f'{name}.{a}'] = f'{bns[bi]}/{b}'
            bi += 1
    return tfname


def push(V, tfname, p):
    with torch.no_grad():
        for k, n in tfname.items():
            V[n].copy_(to_tf(k, p[k]))


def train_run(ref, p0, data, first_step, config=None):
    tf_shim.reset()
    state = {'i': 0}
    config = config or CONFIG

    class It:
        def get_next(self):
            im, g = data[state['i'] % 2]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'data_shape': list(config['data_shape']), 'num_train': 2 * config['batch_size'], 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    m = ref.LHRCNN(dict(config, rpn_first_step=first_step), prov)
    V = tf_shim.S.variables
    tfname = name_map(V)
    push(V, tfname, p0)
    losses, steps, after1 = [], [], None
    for step in range(2):
        state['i'] = step
        _, loss, gs = m.sess.run([m.train_op, m.loss, m.global_step], feed_dict={m.lr: LR_STEP, m.is_training: True})
        losses.append(float(loss)); steps.append(int(gs))
        if step == 0:
            after1 = {k: from_tf(k, V[n].detach().clone()) for k, n in tfname.items()}
    return m, V, tfname, losses, steps, after1


# A second training fixture away from the first one's settings: another picture shape (8 x 15 feature map), three pictures, 5 classes, up to five objects, a
# larger weight decay, other seeds -> lhrcnn_train_b.npz (`python tests/golden/make_golden_lhrcnn.py b` writes only this one).
CASE_B = dict(H=256, W=480, batch=3, num_classes=5, weight_decay=5e-4, seeds=(501, 502), seed_params=73, lr=0.002, max_obj=5, pad=8)


def batches_b():
    c, out = CASE_B, []
    for s in c['seeds']:
        g = torch.Generator().manual_seed(s)
        gt = LR.synthetic_gt(c['batch'], c['H'], c['W'], s + 10, pad=c['pad'], max_obj=c['max_obj'])
        gt[..., 4] = torch.where(gt[..., 4] >= 0, gt[..., 4] % c['num_classes'], gt[..., 4])
        out.append(((torch.rand(c['batch'], c['H'], c['W'], 3, generator=g) * 255).round(), gt))
    return out


def main_b():
    global LR_STEP
    c = CASE_B
    tf_shim.install()
    tf_shim.GATHER_OOB_ZERO = True
    ref = tf_shim.load_reference_module('/root/reference/LH_RCNN.py', 'reference_LHRCNN')
    config = dict(CONFIG, data_shape=[c['H'], c['W'], 3], batch_size=c['batch'], num_classes=c['num_classes'], weight_decay=c['weight_decay'])
    p0 = LR.init_params(c['seed_params'], num_classes=c['num_classes'] + 1)
    data = batches_b()
    keep_lr, LR_STEP = LR_STEP, c['lr']
    try:
        _, V, _, rpn_losses, steps, after1 = train_run(ref, p0, data, 60000, config)
        assert int(V['global_step']) == 2 and steps == [0, 1]
        _, _, _, rcnn_losses, _, _ = train_run(ref, p0, data, 0, config)
    finally:
        LR_STEP = keep_lr
    out = dict(rpn_losses=np.asarray(rpn_losses, np.float64), rcnn_losses=np.asarray(rcnn_losses, np.float64), global_steps=np.asarray(steps),
               case=np.asarray(json.dumps(c)))
    for key in KEEP:
        flat = after1[key].contiguous().reshape(-1)
        out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'lhrcnn_train_b.npz'), **out)
    print('case b: rpn', rpn_losses, 'rcnn', rcnn_losses)
    tf_shim.GATHER_OOB_ZERO = False
    tf_shim.uninstall()


def main():
    tf_shim.install()
    tf_shim.GATHER_OOB_ZERO = True
    ref = tf_shim.load_reference_module('/root/reference/LH_RCNN.py', 'reference_LHRCNN')
    p0 = LR.init_params(SEED_PARAMS)
    data = batches()
    m, V, tfname, rpn_losses, steps, after1 = train_run(ref, p0, data, 60000)
    variables = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable) for n, v in V.items()}
    json.dump(variables, open(os.path.join(OUT, 'lhrcnn_variables.json'), 'w'), indent=0, sort_keys=True)
    json.dump(tfname, open(os.path.join(OUT, 'lhrcnn_names.json'), 'w'), indent=0, sort_keys=True)
    assert int(V['global_step']) == 2 and steps == [0, 1], (int(V['global_step']), steps)          # train_rcnn_op ran on both steps of the "RPN only" phase
    _, V2, _, rcnn_losses, _, after1b = train_run(ref, p0, data, 0)
    worst = max(float((after1[k] - after1b[k]).abs().max() / (after1[k].abs().max() + 1e-30)) for k in after1)
    print('two schedules, same updates: worst relative difference', worst)
    assert worst < 1e-5                       # the schedule only selects the REPORTED loss (not bit-equal: threaded scatter-adds in the crop gradient)
    out = dict(rpn_losses=np.asarray(rpn_losses, np.float64), rcnn_losses=np.asarray(rcnn_losses, np.float64), global_steps=np.asarray(steps))
    for key in KEEP:
        flat = after1[key].contiguous().reshape(-1)
        out[key.replace('.', '__')] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'lhrcnn_train.npz'), **out)
    print('variables', len(variables), 'trainable', sum(v['trainable'] for v in variables.values()), 'rpn', rpn_losses, 'rcnn', rcnn_losses)

    # ---- test mode: calibrated moving statistics (batch statistics of one training-mode pass of the oracle), low score threshold
    tf_shim.reset()
    p = LR.init_params(SEED_PARAMS)
    stats = {}
    with torch.no_grad():
        LR.forward(p, data[0][0], True, stats)
    for name, (mean, unb) in stats.items():
        p[name + '.mmean'], p[name + '.mvar'] = mean.clone(), unb.clone()
    cfg = dict(CONFIG, mode='test', nms_score_threshold=0.06, post_nms_proposal=60)
    mt = ref.LHRCNN(cfg, None)
    Vt = tf_shim.S.variables
    push(Vt, name_map(Vt), p)
    img = data[1][0][:1]
    # test_one_image feeds the tensor that `self.images` names AFTER `/ 127.5 - 1` (LH_RCNN.py:68-69, :467): the caller hands over normalised pictures
    scores, bbox, cid = mt.test_one_image((img / 127.5 - 1.).numpy())
    print('detections', scores.shape, bbox.shape, np.bincount(cid, minlength=20))
    assert scores.shape[0] >= 8
    np.savez_compressed(os.path.join(OUT, 'lhrcnn_detect.npz'), image=img.numpy().astype(np.uint8), scores=scores, bbox=bbox, class_id=cid,
                        score_threshold=0.06, post_nms_proposal=60,
                        **{'stat__' + k.replace('.', '__'): p[k].numpy() for k in p if k.endswith(('.mmean', '.mvar'))})
    tf_shim.GATHER_OOB_ZERO = False
    tf_shim.uninstall()


if __name__ == '__main__':
    if sys.argv[1:] == ['b']:
        main_b()
    else:
        main()
        main_b()
