"""PFPNetR training throughput (testrefinedet.py's configuration: 320 x 320, batch 32): synthetic VOC-shaped batch, random-init weights, full
step (forward, matching + mining + two-stage loss, backward, momentum).  usage: python tools/refinedet_bench.py [dtype=bf16] [batch=32] [steps=5] [graph]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = {'mode': 'train', 'input_size': 320, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': batch,
       'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': dtype, 'use_graph': len(sys.argv) > 4 and sys.argv[4] == 'graph'}
g = torch.Generator().manual_seed(0)
imgs = (torch.rand(batch, 320, 320, 3, generator=g) * 255).round()
gt = S.synthetic_gt(batch, 320, 1, lo=0.1, hi=0.7)
m = odtk.PFPNetR(cfg, {'data_shape': [320, 320, 3], 'num_train': batch, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
m.set_batch(imgs, gt)
for _ in range(2):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flops = sum(3 * 2 * batch * m.desc[n].Ho * m.desc[n].Wo * co * ci * k * k for n, kind, ci, co, k, *_ in m.specs)
print(f'PFPNetR batch {batch} {dtype}: {dt * 1e3:8.2f} ms/step  {batch / dt:8.1f} images/s   conv {flops / dt / 1e12:6.1f} TFLOP/s   loss {float(loss):.3f}')
