"""FCOS training throughput at BASELINE config 5 (512 x 512, 16 images per GPU of the 128 over 8 GPUs): synthetic VOC-shaped batch, random-init weights,
full step (forward, match + loss, backward, optimizer).  usage: python tools/fcos_bench.py [dtype=f32] [batch=16] [steps=5] [size=512]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
size = int(sys.argv[4]) if len(sys.argv) > 4 else 512
cfg = {'mode': 'train', 'data_shape': [size, size, 3], 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
       'batch_size': batch, 'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45, 'verbose': False, 'compute_dtype': dtype}
g = torch.Generator().manual_seed(0)
imgs = (torch.rand(batch, size, size, 3, generator=g) * 255).round()
gt = S.synthetic_gt(batch, size, 1, lo=0.05, hi=0.6)
m = odtk.FCOS(cfg, {'num_train': batch, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
m.set_batch(imgs, gt)
for _ in range(2):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
spec = {s[0]: s for s in m.specs}
flops = 3 * sum(2 * batch * d.Ho * d.Wo * spec[n.split('@')[0]][2] * spec[n.split('@')[0]][1] * spec[n.split('@')[0]][3] ** 2 for n, d in m.desc.items())     # head layers: one launch per level
print(f'FCOS {size}x{size} batch {batch} {dtype}: {dt * 1e3:8.2f} ms/step  {batch / dt:8.1f} images/s   conv {flops / dt / 1e12:6.1f} TFLOP/s   '
      f'loss {float(loss):.3f}')
