"""SSD512 training throughput (testSSD512.py's configuration: 512 x 512, VGG-16, 24 564 priors): synthetic VOC-shaped batch, random-init weights, full step.
usage: python tools/ssd512_bench.py [dtype=bf16] [batch=32] [steps=10]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': batch,
       'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': dtype, 'seed': 0}
g = torch.Generator().manual_seed(0)
imgs = (torch.rand(batch, 512, 512, 3, generator=g) * 255).round()
gt = S.synthetic_gt(batch, 512, 1, lo=0.1, hi=0.7)
m = odtk.SSD512(cfg, {'data_shape': [512, 512, 3], 'num_train': batch, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
m.set_batch(imgs, gt)
for _ in range(8):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'SSD512 batch {batch} {dtype}: {dt * 1e3:8.2f} ms/step  {batch / dt:8.1f} images/s   loss {float(loss):.3f}   launch mode {getattr(m, "launch_mode", None)}')
