"""Does a configuration's step time drift over a run (a data-dependent slow path, like the hard-negative NMS fallback found in round 3)?
Times consecutive blocks of steps of one process.    python tools/step_drift.py <config> [blocks=8] [steps_per_block=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC

name = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
per = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = BC.make(name, use_graph=False)
m = r['model']
m.set_batch(r['images'], r['gt'])
for _ in range(3):
    m.train_step(r['lr'])
torch.cuda.synchronize()
out = []
for b in range(blocks):
    t0 = time.perf_counter()
    for _ in range(per):
        loss = m.train_step(r['lr'])
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / per * 1e3)
print(name, 'ms/step per block of', per, ':', ' '.join(f'{x:.2f}' for x in out), '| loss', float(loss))
