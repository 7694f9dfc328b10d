"""Is the data-parallel SSD300 step (world of one rank, deterministic filter gradients) bit-reproducible, and equal to the plain step?
Plain x3, then DP with the C-ABI collective x4, then DP with torch.distributed x2; for every run the first parameter segments that differ from run 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from oracle import ssd300_ref as R

dev = torch.device('cuda', 0)
B = 4
cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
       'batch_size': B, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '',
       'verbose': False, 'compute_dtype': 'bf16', 'seed': 0, 'deterministic_wgrad': True}
prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
imgs, gt = R.synthetic_batch(B, 7)
if len(sys.argv) > 1 and sys.argv[1] == 'warm':          # grow the library arenas first, as a long test session has
    big = odtk.SSD300(dict(cfg, batch_size=32), dict(prov, num_train=32))
    i2, g2 = R.synthetic_batch(32, 9)
    big.set_batch(i2, g2)
    for _ in range(3):
        big.train_step(0.002)
    torch.cuda.synchronize()
    del big


def run(mode, steps=3):
    m = odtk.SSD300(cfg, prov)
    if mode != 'plain':
        m.attach_data_parallel(bucket_mb=8, force_collectives=True, collective=mode.split(':')[0])
        if mode == 'odtk:caller':
            m.dist.red.collective.stream = None              # on the launching (tail filter-gradient) stream instead of the side stream
    m.set_batch(imgs, gt)
    losses, grads = [], []
    for _ in range(steps):
        losses.append(float(m.train_step(0.002)))
        grads.append(m.G.clone())
    torch.cuda.synchronize()
    if mode.startswith('odtk'):
        m.dist.red.collective.close()
    return m, losses, grads, m.P.clone()


ref = None
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for mode in ['plain'] * 2 + ['odtk'] * NR + ['odtk:caller'] * NR + ['torch'] * NR:
    if mode == 'torch':
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    m, losses, grads, P = run(mode)
    if ref is None:
        ref = (m, losses, grads, P)
        print('run 0 (plain) losses', losses)
        continue
    msg = []
    for s, (g, g0) in enumerate(zip(grads, ref[2])):
        if not torch.equal(g, g0):
            bad = []
            for name, (off, shape) in m.pinfo.items():
                n = 1
                for d in shape:
                    n *= d
                a, b = g[off:off + n], g0[off:off + n]
                if not torch.equal(a, b):
                    bad.append((name, float((a - b).abs().max() / (b.abs().max() + 1e-30))))
            msg.append(f'step {s + 1}: {len(bad)} segments differ, first {bad[:4]}')
    msg = [x[:60] for x in msg]
    print(f'{mode:12s} losses equal {losses == ref[1]}  P max diff {float((P - ref[3]).abs().max()):.3e}  ' + ('; '.join(msg) if msg else 'all gradients bit-identical'), flush=True)
