"""where does the f32 YOLOv3 gradient error start?  usage (GPU box): python tests/tools/debug_yolo_grad.py [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import yolov3_net_ref as NR, yolov3_ref as YR
import test_gpu_yolov3 as T

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_num_threads(16)
p = NR.init_params(5)
imgs, gt = T._batch(2, size, 40)
m = T._model('train', 'f32', 2, size, T._provider([(imgs, gt)]))
m.load_oracle_params(p)
m.set_batch(imgs, gt)
m.G.zero_(); m._forward(True); m._loss(0.5 / 2)
for _ in m._backward_iter():
    pass
torch.cuda.synchronize()
q = {k: v.clone().requires_grad_(not k.endswith(('.mmean', '.mvar'))) for k, v in p.items()}
taps = {}
preds = NR.forward(q, imgs, True, taps=taps)
for t in taps.values():
    t.retain_grad()
for t in preds:
    t.retain_grad()
C = 20
data = YR.batch_loss(preds, gt, num_classes=C, coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.)
(.5 * data).backward()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
for l in range(3):
    print('pred', l, rel(m.preds[l].cpu(), preds[l].detach()), 'dpred', rel(m.dpreds[l].cpu(), preds[l].grad),
          'max|d|', float((m.dpreds[l].cpu() - preds[l].grad).abs().max()), float(preds[l].grad.abs().max()))
for name in ('c74', 'c73', 'c72', 'c66', 'c65', 'c58', 'c57', 'c56', 'c52', 'c51', 'c50', 'c26', 'c1', 'c0'):
    a = m.acts[name]
    ref = taps[name].detach().permute(0, 2, 3, 1).reshape(a.M, -1)
    line = f'{name}: act {rel(a.t[:, :a.C].float().cpu(), ref):.2e} sign flips {int(((a.t[:, :a.C].float().cpu() > 0) != (ref > 0)).sum())}/{ref.numel()}'
    for s in ('.w', '.gamma', '.beta'):
        line += f'  d{s} {rel(m.get_param(name + s, m.G), q[name + s].grad):.2e}'
    gid = m.find(a.gid)
    if gid in m.g and name not in ('c74', 'c66', 'c58'):
        line += f'  dact(group) {rel(m.g[gid][:, :a.C].float().cpu(), taps[name].grad.permute(0, 2, 3, 1).reshape(a.M, -1)):.2e}'
    print(line)
