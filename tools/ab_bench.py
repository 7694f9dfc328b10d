"""Same-process, interleaved A/B of the SSD300 training step (batch 32, bf16, eager launches): every variant is a set of libodtk debug switches
(`odtk_debug_set key:value`, applied before its block of steps) and / or model config overrides (a second model instance on the same weights).
Blocks of `--block` steps are run round-robin for `--rounds` rounds, so clock / thermal drift hits every variant alike; the table gives each
variant's median ms/step and the median of its PER-ROUND ratio to the first variant -- run-to-run noise of separate processes (+-1 %) hides
the 50-150 us effects this is for.

    python tools/ab_bench.py base= wide=4:-3 rows4096=4:4096 nofuse=cfg:fuse_pool=0 [--rounds 8] [--block 25]
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402

import bench          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+', help='name=KEY:VALUE[,KEY:VALUE...] | name=cfg:KEY=VALUE[,...] | name= (nothing)')
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--block', type=int, default=25)
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    import odtk
    from odtk import ops
    dev = torch.device('cuda', 0)
    B = args.batch
    base_cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': B,
                'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'compute_dtype': 'bf16',
                'verbose': False, 'seed': 0, 'use_graph': False}
    prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    images, gt = bench.synthetic_batch(B, 1000, dev)
    variants, models = [], {}
    for v in args.variants:
        name, _, spec = v.partition('=')
        sets, cfg = [], {}
        for item in filter(None, spec.split(',')):
            if item.startswith('cfg:'):
                k, _, val = item[4:].partition('=')
                cfg[k] = {'0': False, '1': True}.get(val, int(val) if val.isdigit() else val)
            else:
                k, val = item.split(':')
                sets.append((int(k), int(val)))
        key = tuple(sorted(cfg.items()))
        if key not in models:
            m = odtk.SSD300(dict(base_cfg, **cfg), prov)
            m.set_batch(images, gt)
            models[key] = m
        variants.append((name, sets, models[key]))
    resets = sorted({(k, {2: 0, 3: 0, 4: 1024, 5: 0}.get(k, 0)) for _, sets, _ in variants for k, _ in sets})
    lr = 0.01

    def apply(sets):
        for k, v in resets:
            ops.debug_set(k, v)
        ops.debug_set(4, -1); ops.debug_set(4, -4); ops.debug_set(4, -6)          # batch-norm launch shapes back to their defaults
        for k, v in sets:
            ops.debug_set(k, v)

    for _, sets, m in variants:                                  # warm-up: every variant once (lazily grown scratch, clocks)
        apply(sets)
        for _ in range(10):
            m.train_step(lr)
    torch.cuda.synchronize()
    times = {name: [] for name, _, _ in variants}
    for r in range(args.rounds):
        order = variants if r % 2 == 0 else variants[::-1]
        for name, sets, m in order:
            apply(sets)
            for _ in range(3):
                m.train_step(lr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.block):
                m.train_step(lr)
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / args.block * 1e3)
    ref = variants[0][0]
    print(f'| variant | median ms/step | min | max | median per-round ratio to `{ref}` |\n|---|---|---|---|---|')
    for name, _, _ in variants:
        t = times[name]
        ratio = statistics.median(a / b for a, b in zip(t, times[ref]))
        print(f'| {name} | {statistics.median(t):.3f} | {min(t):.3f} | {max(t):.3f} | {ratio:.4f} |')


if __name__ == '__main__':
    main()
