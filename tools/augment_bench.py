"""Timing of the GPU augmentor at the SSD300 driver configuration (testSSD300.py:34-46): 32 VOC-sized u8 pictures
(375 x 500) -> 300 x 300 f32 with flips, colour jitter and rotate, boxes padded to 60.
usage: python tools/augment_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import augment

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
N = 32
imgs = [torch.randint(0, 256, (375, 500, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(N)]
gts = [torch.tensor([[50., 200., 80., 300., 3.], [120., 330., 200., 460., 7.]]).to(dev) for _ in range(N)]
for name, cfg in (('geometry only', dict(flip_prob=[0., 0.5])),
                  ('driver config', dict(flip_prob=[0., 0.5], color_jitter_prob=0.5, rotate=[0.5, -5., -5.])),
                  ('everything on', dict(flip_prob=[1., 1.], color_jitter_prob=1.0, rotate=[1.0, -5., 5.]))):
    aug = augment.Augmentor('channels_last', [300, 300], crop_method='random', fill_mode='BILINEAR', pad_truth_to=60, seed=1, **cfg)
    for _ in range(3):
        aug(imgs, gts)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    s.record()
    for _ in range(reps):
        aug(imgs, gts)
    e.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    dt = s.elapsed_time(e) / reps * 1e-3
    print(f'{name}: {dt * 1e6:8.1f} us / batch of {N} on the stream ({N / dt:9.0f} img/s), host-inclusive {wall * 1e6:8.1f} us ({N / wall:9.0f} img/s)')
