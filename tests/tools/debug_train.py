"""Debug helper (GPU): per-tensor comparison of one f32/bf16 training step against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ssd300_ref as R
import odtk

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
B = 2
torch.set_num_threads(16)
cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
       'batch_size': B, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5,
       'pretraining_weight': '', 'verbose': False, 'compute_dtype': dtype}
imgs, gt = R.synthetic_batch(B, 40)
prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None}
m = odtk.SSD300(cfg, prov)
p = R.init_params(5)
m.load_oracle_params(p)
m.set_batch(imgs, gt)
m.G.zero_()
m._forward(True)
m._loss(1.0 / B)
m._backward()
torch.cuda.synchronize()

names = R.trainable_names(p)
for k in names:
    p[k].requires_grad_(True)
taps = {'_retain': True}
stats = {}
pred = R.forward(p, imgs, True, stats, taps)
pred.retain_grad()
anchors = R.priors()
loss = R.batch_loss(pred, anchors, gt)
loss.backward()


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20), float((a - b).norm()) / (float(b.norm()) + 1e-20)


print('loss gpu', float(m.data_loss.item()) / B, 'ref', float(loss))
print('pred', rel(m.pred.cpu(), pred.detach()))
print('dpred', rel(m.dpred.cpu(), pred.grad))
for i in range(B):
    d = R.one_image_loss(pred[i, :, 21:23].detach(), pred[i, :, 23:].detach(), pred[i, :, :21].detach(), anchors, gt[i], detail=True)
    got = set(m.sel_idx[i, : int(m.sel_cnt[i])].cpu().tolist()); exp = set(d['sel_anchor'].tolist())
    print('image', i, 'sel', len(got), len(exp), 'symdiff', len(got ^ exp), 'parts', m.loss_parts[i].cpu().tolist(),
          [float(d[k]) for k in ('neg_loss', 'pos_conf_loss', 'coord', 'total')])
for name in ['conv11_2', 'conv10_2', 'conv9_2', 'conv8_2', 'conv7', 'conv6', 'pool5', 'conv5_3', 'conv5_1', 'pool4', 'conv4_3',
             'conv4_2', 'conv4_1', 'pool3', 'conv3_3', 'conv2_2', 'conv1_2', 'conv1_1']:
    raw = taps.get(name + '.raw')
    if raw is None or raw.grad is None:
        continue
    a = m.acts[name]
    g = a.g.float().cpu()[:, : a.C].reshape(a.N, a.H, a.W, a.C)
    print('dact', name, rel(g, raw.grad.permute(0, 2, 3, 1)))
for name in ['conv11_2', 'conv11_1', 'conv10_2', 'conv10_1', 'conv9_2', 'conv9_1', 'conv8_2', 'conv8_1', 'conv7', 'conv6']:
    a = m.acts[name]; z = m.zbuf[name]
    raw = taps[name + '.raw']; zr = taps[name + '.z.raw']
    g = a.g.float().cpu()[:, : a.C].reshape(a.N, a.H, a.W, a.C)
    gz = z.g.float().cpu()[:, : z.C].reshape(z.N, z.H, z.W, z.C)
    yv = a.t.float().cpu()[:, : a.C].reshape(a.N, a.H, a.W, a.C)
    print('extra', name, 'y', rel(yv, raw.detach().permute(0, 2, 3, 1)), 'dy', rel(g, raw.grad.permute(0, 2, 3, 1)), 'dz', rel(gz, zr.grad.permute(0, 2, 3, 1)),
          'mask mismatch', int(((yv > 0) != (raw.detach().permute(0, 2, 3, 1) > 0)).sum()),
          'dbeta check', rel((a.g.float() * (a.t.float() > 0)).sum(0).cpu(), m.param(name + '.beta', m.G).cpu()))
for k in names:
    g = m.param(k, m.G).cpu()
    if k.endswith('.w'):
        g = g[..., : p[k].shape[-1]]
    print('dparam', k, rel(g.reshape(p[k].shape), p[k].grad), float(p[k].grad.abs().max()))
