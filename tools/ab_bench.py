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
    ap.add_argument('--own-streams', action='store_true', help='every model instance keeps the side streams it created (shows the hardware-queue aliasing)')
    ap.add_argument('--clocks', action='store_true', help='sample the shader clock and socket power of this card during every block')
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
                cfg[k] = ({'0': False, '1': True} if k != 'seed' else {}).get(val, int(val) if val.isdigit() else (float(val) if val.replace('.', '', 1).isdigit() else val))
            else:
                k, val = item.split(':')
                sets.append((int(k), int(val)))
        key = tuple(sorted(cfg.items()))
        if key not in models:
            m = odtk.SSD300(dict(base_cfg, **cfg), prov)
            # The model class shares one set of side streams per device (ssd300._side_stream).  --own-streams gives every further instance fresh ones, as before
            # round 3: the runtime deals HIP streams onto FOUR hardware queues, a later instance's side stream can alias the main stream's queue -- the 2nd
            # instance ran 1.4-2 % slower and the 4th, 6th, 8th 7 % slower whatever their configuration, which is what the cfg: variants of rounds 2-3 measured.
            if models and args.own_streams:
                for attr in ('_side', '_tail', '_twg', 'wgrad_stream'):
                    if getattr(m, attr, None) is not None:
                        setattr(m, attr, torch.cuda.Stream(device=dev))
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
    # shader clock / socket power of THIS card while a block runs (hwmon sysfs, ~5 ms period): is a slower variant slower at the same clock?
    sampler = None
    if args.clocks:
        import threading
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import clock_trace as CT
        p_ = torch.cuda.get_device_properties(0)
        pci = '%04x:%02x:%02x.0' % (p_.pci_domain_id, p_.pci_bus_id, p_.pci_device_id)
        src = [c for c in CT.sysfs_sources() if os.path.realpath(f"/sys/class/drm/{c['card']}/device").endswith(pci)]
        if src:
            src = src[0]
            state = {'on': False, 'stop': False, 'buf': []}

            def run():
                while not state['stop']:
                    if state['on']:
                        state['buf'].append((CT.read_int(src['sclk_hz']), CT.read_int(src['power_uw'])))
                    time.sleep(0.005)
            sampler = (threading.Thread(target=run, daemon=True), state)
            sampler[0].start()
    clocks = {name: [] for name, _, _ in variants}
    for r in range(args.rounds):
        order = variants if r % 2 == 0 else variants[::-1]
        for name, sets, m in order:
            apply(sets)
            for _ in range(3):
                m.train_step(lr)
            torch.cuda.synchronize()
            if sampler:
                sampler[1]['buf'] = []; sampler[1]['on'] = True
            t0 = time.perf_counter()
            for _ in range(args.block):
                m.train_step(lr)
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / args.block * 1e3)
            if sampler:
                sampler[1]['on'] = False
                clocks[name] += [b for b in sampler[1]['buf'] if b[0] and b[1]]
    ref = variants[0][0]
    print(f'| variant | median ms/step | min | max | median per-round ratio to `{ref}` | shader clock MHz (median) | socket power W (median) |\n|---|---|---|---|---|---|---|')
    for name, _, _ in variants:
        t = times[name]
        ratio = statistics.median(a / b for a, b in zip(t, times[ref]))
        ck = clocks[name]
        cs = f"{statistics.median(c[0] for c in ck) / 1e6:.0f} | {statistics.median(c[1] for c in ck) / 1e6:.0f}" if ck else '- | -'
        print(f'| {name} | {statistics.median(t):.3f} | {min(t):.3f} | {max(t):.3f} | {ratio:.4f} | {cs} |')
    if sampler:
        sampler[1]['stop'] = True


if __name__ == '__main__':
    main()
