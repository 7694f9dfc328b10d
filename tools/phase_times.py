"""Where the SSD300 training step spends its time WITHOUT a profiler attached: HIP events on the MAIN stream at the phase boundaries of the step
(ssd300.py `_phase`: a dozen event records per step, nothing on the side streams), median over the timed steps.  rocprofv3's kernel trace inflates the
launch-bound small-map regions (the traced step is 0.6 ms longer than the timed one); this is the undisturbed picture.

    python tools/phase_times.py [steps=30] [batch=32]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402

import bench          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    import odtk
    dev = torch.device('cuda', 0)
    cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': B,
           'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'compute_dtype': 'bf16',
           'verbose': False, 'seed': 0, 'use_graph': False}
    m = odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    images, gt = bench.synthetic_batch(B, 1000, dev)
    m.set_batch(images, gt)
    for _ in range(10):
        m.train_step(0.01)
    torch.cuda.synchronize()
    rows = []
    for _ in range(steps):
        m._phases = []
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        m.train_step(0.01)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        rows.append([('step start', e0)] + m._phases + [('optimizer + loss scalar done', e1)])
    torch.cuda.synchronize()
    m._phases = None
    names = [n for n, _ in rows[0]]
    print(f'| phase boundary (event on the main stream) | ms since step start (median of {steps}) | ms since the previous boundary |\n|---|---|---|')
    prev = 0.0
    for i, n in enumerate(names):
        t = statistics.median(r[0][1].elapsed_time(r[i][1]) for r in rows)
        print(f'| {n} | {t:.3f} | {t - prev:.3f} |')
        prev = t


if __name__ == '__main__':
    main()
