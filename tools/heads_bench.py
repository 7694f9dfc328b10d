"""Timing of the CenterNet / FCOS / YOLOv3 box-side kernels at the BASELINE.json config sizes (per-GPU batch).
usage: python tools/heads_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


g = torch.Generator().manual_seed(0)
# CenterNet, config 5: 512 x 512, 16 images / GPU -> 128 x 128 x 20
N, H, W, C = 16, 128, 128, 20
kp = (torch.randn(N, H, W, C, generator=g) - 2).to(dev); off = torch.rand(N, H, W, 2, generator=g).to(dev); size = (torch.rand(N, H, W, 2, generator=g) * 40).to(dev)
gt = S.synthetic_gt(N, 512, 1, hi=0.7).to(dev)
parts = torch.zeros(N, 4, device=dev); dk, do, dz = torch.empty_like(kp), torch.empty_like(off), torch.empty_like(size)
ws = ops.centernet_workspace(N, H, W, C, dev)
t = timeit(lambda: ops.centernet_loss(kp, off, size, gt, 4.0, 1.0 / N, parts, dk, do, dz, ws))
b = (kp.numel() * 2 + off.numel() * 4) * 4
print(f'centernet_loss   N={N} {H}x{W}x{C}: {t:8.1f} us   {b / t / 1e6:7.2f} TB/s (read logits + write grads)')
t = timeit(lambda: ops.centernet_decode(kp[0], off[0], size[0], 4.0, 0.1, 100, ws))
print(f'centernet_decode 1 image (incl. count read-back): {t:8.1f} us')
# FCOS, config 5: 512 x 512, 16 images / GPU, 21 classes
shapes = S.pyramid_shapes(512, 512)
conf = [(torch.randn(N, h, w, 21, generator=g) - 2).to(dev) for h, w in shapes]
reg = [torch.exp(torch.randn(N, h, w, 4, generator=g)).to(dev) for h, w in shapes]
cen = [torch.randn(N, h, w, 1, generator=g).to(dev) for h, w in shapes]
gt = S.synthetic_gt(N, 512, 2, lo=0.05, hi=0.9).to(dev)
loss = torch.zeros(N, device=dev)
dc, dr, dz = [torch.empty_like(t_) for t_ in conf], [torch.empty_like(t_) for t_ in reg], [torch.empty_like(t_) for t_ in cen]
ws = ops.fcos_workspace(conf, N, dev)
t = timeit(lambda: ops.fcos_loss(conf, reg, cen, gt, 1.0 / N, loss, dc, dr, dz, ws))
print(f'fcos_loss        N={N} 5456 locations x 21: {t:8.1f} us  (3 launches)')
# YOLOv3, config 4: 416 x 416, 8 images / GPU
N = 8
preds = [torch.randn(N, h, h, 3, 25, generator=g).to(dev) for h in (13, 26, 52)]
gt = S.synthetic_gt(N, 416, 3, lo=0.05, hi=0.8).to(dev)
parts = torch.zeros(N, 5, device=dev); dp = [torch.empty_like(p) for p in preds]
ws = ops.yolov3_workspace(preds, N, dev)
pri = S.yolo_priors_flat()
t = timeit(lambda: ops.yolov3_loss(preds, pri, [32., 16., 8.], gt, (1., 1., 5., 1.), 1.0 / N, parts, dp, ws))
print(f'yolov3_loss      N={N} 10647 predictions x 25: {t:8.1f} us  (3 memsets + 2 launches)')
