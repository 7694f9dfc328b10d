"""What does an IDLE RCCL communicator cost the single-device step?  (round 5: `bench.py --dp-world1 --collective odtk` ran 0.5 ms/step slower than the
torch.distributed path even with the collectives switched off.)  SSD300 b32 bf16 plain step timed (a) alone, (b) with an idle odtk communicator in the
process, (c) after destroying it, (d) with an idle torch.distributed nccl group, (e) both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from bench import synthetic_batch

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
B = 32
cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': B,
       'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': 'bf16', 'seed': 0}
prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
m = odtk.SSD300(cfg, prov)
images, gt = synthetic_batch(B, 1000, dev)
m.set_batch(images, gt)


def timed(tag, steps=30):
    for _ in range(5):
        m.train_step(0.001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step(0.001)
    torch.cuda.synchronize()
    print(f'{tag:60s} {(time.perf_counter() - t0) / steps * 1e3:7.3f} ms/step', flush=True)


timed('plain')
timed('plain again')
from odtk.dist import OdtkCollective
c = OdtkCollective(None, dev)
timed('idle odtk communicator (never used)')
x = torch.ones(1 << 20, device=dev)
c.all_reduce(x).wait()
torch.cuda.synchronize()
timed('idle odtk communicator (used once)')
c.close()
timed('odtk communicator destroyed')
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
timed('idle torch.distributed nccl group (used once)')
c = OdtkCollective(None, dev)
c.all_reduce(x).wait()
torch.cuda.synchronize()
timed('torch group + odtk communicator, both idle')
c.close()
dist.destroy_process_group()
timed('both destroyed')
