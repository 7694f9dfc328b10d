"""Where do the 0.5 ms/step of `bench.py --dp-world1 --collective odtk` come from?  SSD300 b32 bf16, one process:
plain -> attached (collectives off) -> collectives on the side stream -> on the launching stream -> detached."""
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
    print(f'{tag:70s} {(time.perf_counter() - t0) / steps * 1e3:7.3f} ms/step', flush=True)


timed('plain')
which = sys.argv[1] if len(sys.argv) > 1 else 'odtk'
if which == 'torch':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
red = m.attach_data_parallel(force_collectives=True, collective=which)
red.red.enabled = False
timed(f'{which}: attached, collectives OFF')
red.red.enabled = True
timed(f'{which}: collectives ON (odtk: on the side stream)')
if which == 'odtk':
    red.red.collective.stream = None
    timed('odtk: collectives ON, on the launching (tail filter-gradient) stream')
    red.red.collective.stream = torch.cuda.Stream()
    timed('odtk: collectives ON, on a private stream')
    red.red.collective.stream = m._side
red.red.enabled = False
timed(f'{which}: collectives OFF again')
m.dist = None
m.loss_divisor_batch = B
timed('detached (plain again)')
