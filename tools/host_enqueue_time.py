"""How long does the HOST need to enqueue one SSD300 training step (eager launches), against how long the GPU needs to run it?
After a device sync the host enqueues `k` steps without waiting: the wall time of the enqueue calls alone is the host cost; the sync after them gives the GPU time.
    python tools/host_enqueue_time.py [--list]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC

mode = 'list' if '--list' in sys.argv else False
name = next((a for a in sys.argv[1:] if not a.startswith('--')), 'ssd300')
r = BC.make(name, use_graph=mode)
m = r['model']
m.set_batch(r['images'], r['gt'])
for _ in range(10):
    m.train_step(r['lr'])
torch.cuda.synchronize()
from odtk import _lib
calls = [0]
orig = _lib.call
rows = []
for k in (1, 2, 4, 8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        m.train_step(r['lr'])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((k, (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3))
print('| steps enqueued back to back | host ms/step (enqueue only) | wall ms/step (until the GPU is done) |\n|---|---|---|')
for k, h, w in rows:
    print(f'| {k} | {h:.2f} | {w:.2f} |')
if name == 'ssd300':
    # per-phase host cost of one step: forward+loss | backward (enqueue only, GPU idle at the start)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m._step_front(); t1 = time.perf_counter(); m._backward(); t2 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'\nhost enqueue of forward + loss {1e3 * (t1 - t0):.2f} ms, backward {1e3 * (t2 - t1):.2f} ms (GPU idle at the start, eager)')
