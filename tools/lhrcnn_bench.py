"""Light-Head R-CNN training throughput (testlhrcnn.py's configuration: 700 x 1100, batch 32; f32 engine -- the only one this class has): synthetic VOC-shaped
batch, random-init weights, the full step (forward, RPN loss with its two NMS, crop + dense head, both backward passes, both momentum updates).
usage: python tools/lhrcnn_bench.py [batch=32] [steps=3] [H=700] [W=1100] [dtype=f32]   (bf16: opt-in engine, not yet run on the GPU as a whole)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else 700
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1100
dtype = sys.argv[5] if len(sys.argv) > 5 else 'f32'
cfg = {'data_shape': [H, W, 3], 'mode': 'train', 'is_pretraining': False, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
       'batch_size': batch, 'rpn_first_step': 60000, 'rcnn_first_step': 100000, 'rpn_second_step': 160000, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20,
       'nms_iou_threshold': 0.45, 'post_nms_proposal': 500, 'verbose': False, 'compute_dtype': dtype}
g = torch.Generator().manual_seed(0)
imgs = (torch.rand(batch, H, W, 3, generator=g) * 255).round()
gt = torch.full((batch, 60, 5), -1.0)
for i in range(batch):
    n = int(torch.randint(1, 7, (1,), generator=g))
    h = (0.15 + 0.5 * torch.rand(n, generator=g)) * H
    w = (0.15 + 0.5 * torch.rand(n, generator=g)) * W
    gt[i, :n] = torch.stack([h / 2 + torch.rand(n, generator=g) * (H - 1 - h), w / 2 + torch.rand(n, generator=g) * (W - 1 - w), h, w,
                             torch.randint(0, 20, (n,), generator=g).float()], 1)
m = odtk.LHRCNN(cfg, {'data_shape': [H, W, 3], 'num_train': batch, 'num_val': 0, 'train_generator': [(imgs, gt)], 'val_generator': None})
m.set_batch(imgs, gt)
for _ in range(2):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ws = m.loss.ws
print(f'LHRCNN {H}x{W} batch {batch} {dtype}: {dt * 1e3:8.2f} ms/step  {batch / dt:8.1f} images/s   anchors inside the picture {m.anc["yx"].shape[0]} of {m.A}   '
      f'crop rows {int(ws["roi_counts"].sum())} of {256 * batch}   losses rpn {float(m.last_losses[0]):.3f} rcnn {float(m.last_losses[1]):.3f}   '
      f'peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
