#!/usr/bin/env python
"""Benchmark of the SSD300 VGG-16 training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without a launcher it starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one full training step (preprocess + forward + loss incl. hard-negative-mining NMS +
backward + RCCL gradient all-reduce + fused SGD-momentum) on a synthetic VOC-shaped batch of 32
images/GPU already resident in HBM.  Weak scaling: every rank keeps batch 32 (each replica is
exactly the reference computation, local BatchNorm; SURVEY.md 8e option A).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the DOMINANT device kernel of the step (largest summed duration; the raster-run
                  halo implicit-GEMM kernel that serves conv forward and dgrad), HIP-event timed
                  right after the timed region on the launch stream: algorithmic FLOPs per launch / average
                  launch duration vs the dense bf16 MFMA peak (2.5 PFLOP/s).  `traffic` = HBM bytes
                  per launch of that kernel from two `rocprofv3 --pmc` child passes run NOW (round 5:
                  FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE; null when rocprofv3 is
                  missing -- never a committed constant).  `family` = the same ratio over every conv
                  kernel (fwd + dgrad + wgrad, all layers).
  cpu_baseline -- the CPU oracle (PyTorch-CPU restatement of the reference graph; TF 1.13 cannot
                  run here) timed on this box's host cores on a bounded sample (N=1 only).
After the headline fields are final (N = 1, default configuration; `--no-extras` skips them):
  sustained    -- the same step for >= 250 steps (~2 s) between two synchronisations
  inference    -- SSD300.test_one_image latency per engine (f32 = the test-mode default, f32x3, bf16), CPU oracle beside it
  configs      -- compact records of BASELINE.json's configurations 3-5 (yolov3, fcos, centernet, retinanet), one child run each
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2.5e15        # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_F32 = 157.3e12


def synthetic_batch(batch, seed, device):
    """VOC-shaped synthetic batch: U[0,255) images, 1-6 boxes, GT padded to 60 rows."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(batch, 300, 300, 3, generator=g) * 255.
    gt = torch.full((batch, 60, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, 7, (1,), generator=g))
        h = torch.rand(n, generator=g) * 240. + 30.
        w = torch.rand(n, generator=g) * 240. + 30.
        yc = h / 2 + torch.rand(n, generator=g) * (300 - h)
        xc = w / 2 + torch.rand(n, generator=g) * (300 - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return images.to(device), gt.to(device)


class ConvTimer:
    """Wraps the three conv entry points with HIP events on the launch stream."""

    def __init__(self, ops, alg=None):
        self.ops = ops
        self.records = []            # (kind, kernel, flops, start_evt, end_evt, algorithmic bytes)
        self.enabled = False
        self._orig = {}
        self.alg = alg or {}         # id(ConvDesc) -> (cin, cout): ALGORITHMIC channel counts (descriptors carry chunk-padded ones: 7 -> 8, 3 -> 8)

    def install(self):
        ops = self.ops
        for kind in ('conv2d_fwd', 'conv2d_fwd_pool2x2', 'conv2d_dgrad', 'conv2d_wgrad'):
            self._orig[kind] = getattr(ops, kind)
            setattr(ops, kind, self._wrap(kind))

    def _wrap(self, kind):
        orig = self._orig[kind]

        def f(d, *args):
            if not self.enabled:
                return orig(d, *args)
            if id(d) in self.alg:
                c_alg, k_alg = self.alg[id(d)]
            else:
                c_alg, k_alg = (3 if (d.H == 300 and d.C <= 8) else d.C), d.K    # conv1_1: 3 real input channels (padded to one chunk)
            flops = 2.0 * d.N * d.Ho * d.Wo * k_alg * d.R * d.S * c_alg  # algorithmic: 2*M*Cout*R*S*Cin for each pass
            esz = 2 if d.dtype == 0 else 4
            # algorithmic HBM bytes: every operand once (x, y / dy, dx as bf16; filter as bf16, dW as f32)
            abytes = (d.N * d.H * d.W * d.C + d.N * d.Ho * d.Wo * d.K) * esz + d.K * d.R * d.S * d.C * (4 if kind == 'conv2d_wgrad' else esz)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(d, *args)
            e.record()
            rkind = 'conv2d_fwd' if kind == 'conv2d_fwd_pool2x2' else kind      # conv + bias + ReLU + pool in one launch: the conv's FLOPs, its own time
            self.records.append((rkind, self.ops.conv_last_kernel(), flops, s, e, abytes,
                                 (d.N, d.H, d.W, d.C, d.K, d.R, d.stride, d.dil)))
            return r
        return f

    def per_step(self, steps, trimmed=False):
        """One step's launches with the mean duration of each launch over the recorded steps (the launch sequence of a
        step is fixed, so record i and record i + launches_per_step are the same launch).  Default: the PLAIN average over
        every recorded step but the first (the first eager step after graph replay is a warm-up of the eager path) --
        what `rocprofv3 --kernel-trace --stats` reports for the kernel.  trimmed=True additionally drops the slowest sample
        of every launch; it is reported next to the plain number, never instead of it."""
        n = len(self.records)
        per = n // steps if steps > 0 and n % steps == 0 else n
        self._per = per
        reps = n // per
        out = []
        for i in range(per):
            ts = [self.records[i + r * per][3].elapsed_time(self.records[i + r * per][4]) * 1e-3 for r in range(reps)]
            if reps >= 3:
                ts = ts[1:]
            if trimmed:
                ts.sort()
                if len(ts) >= 4:
                    ts = ts[:-1]
            kind, kern, fl, _, _, ab, shp = self.records[i]
            out.append((kind, kern, fl, sum(ts) / len(ts), ab, shp))
        return out

    def per_step_totals(self, kernel):
        """summed duration of `kernel`'s launches in every recorded eager step (ms) -- shows whether a slow average is one slow step"""
        per = self._per
        out = []
        for r in range(len(self.records) // per):
            out.append(round(sum(self.records[i + r * per][3].elapsed_time(self.records[i + r * per][4])
                                 for i in range(per) if self.records[i][1] == kernel), 3))
        return out

    def summary(self, steps, trimmed=False):
        per_kernel, per_pass = {}, {}
        for kind, kern, fl, t, ab, _ in self.per_step(steps, trimmed):
            for d, k in ((per_kernel, kern), (per_pass, kind)):
                a = d.setdefault(k, [0.0, 0.0, 0, 0.0]); a[0] += fl; a[1] += t; a[2] += 1; a[3] += ab
        return per_kernel, per_pass

    def table(self, steps):
        """Per-layer rows (pass, kernel, shape, launches per step, median us, TFLOP/s), slowest first."""
        agg = {}
        for kind, kern, fl, tm, ab, shp in self.per_step(steps):
            a = agg.setdefault((kind, kern, shp), [0.0, 0.0, 0])
            a[0] += fl; a[1] += tm; a[2] += 1
        rows = ['%-13s %-28s N%d H%d W%d C%d K%d k%d s%d d%d  x%d  %8.1f us  %7.1f TF' %
                (k[0], k[1], *k[2], v[2], v[1] / v[2] * 1e6, v[0] / v[1] / 1e12)
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1] / kv[1][2])]
        return '\n'.join(rows)

    def roofline(self, steps, peak, whole_step_frac):
        per_kernel, per_pass = self.summary(steps)
        trimmed_kernel, _ = self.summary(steps, trimmed=True)
        steps = 1                                   # summary() is per step already
        dom = max(per_kernel, key=lambda k: per_kernel[k][1])
        fl, t, n, ab = per_kernel[dom]
        t_trim = trimmed_kernel[dom][1]
        pmc = None                                  # HBM traffic / MFMA-busy counters: measured live after the timed region (live_pmc_traffic), never read from a file
        tot_f = sum(v[0] for v in per_kernel.values()); tot_t = sum(v[1] for v in per_kernel.values())
        return {
            'bound': 'mfma', 'kernel': dom,
            'achieved': round(fl / t / 1e12, 2), 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': round(fl / t / peak, 4),
            'traffic': pmc,
            # SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (launch duration x 2.4 GHz x 1024 SIMDs)
            'mfma_busy_pct': None if (pmc is None or not pmc.get('mfma_busy_cycles')) else
            round(100.0 * pmc['mfma_busy_cycles'] / ((t / n) * 2.4e9 * 1024), 1),
            'hbm_gbps': None if pmc is None else round(pmc['hbm_bytes'] / (t / n) / 1e9, 1),
            'hbm_frac_of_8TBps': None if pmc is None else round(pmc['hbm_bytes'] / (t / n) / 8e12, 4),
            'launches_per_step': n // steps, 'avg_launch_us': round(t / n * 1e6, 2),
            'avg_launch_us_trimmed': round(t_trim / n * 1e6, 2), 'frac_trimmed': round(fl / t_trim / peak, 4),
            'per_step_ms': self.per_step_totals(dom),
            'algorithmic_gflop_per_launch': round(fl / n / 1e9, 2),
            'algorithmic_mb_per_launch': round(ab / n / 1e6, 1),
            'family': {'kernels': 'all conv kernels (fwd + dgrad + wgrad, every layer)',
                       'achieved': round(tot_f / tot_t / 1e12, 2), 'frac': round(tot_f / tot_t / peak, 4),
                       'launches_per_step': sum(v[2] for v in per_kernel.values()), 'conv_ms_per_step': round(tot_t / steps * 1e3, 3)},
            'by_kernel': {k: {'TFLOP/s': round(v[0] / v[1] / 1e12, 2), 'ms_per_step': round(v[1] / steps * 1e3, 3),
                              'launches_per_step': v[2] // steps} for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
            'by_pass': {k: {'TFLOP/s': round(v[0] / v[1] / 1e12, 2), 'ms_per_step': round(v[1] / steps * 1e3, 3)}
                        for k, v in per_pass.items()},
            'whole_step_frac_of_mfma_peak': round(whole_step_frac, 4),
        }


class ClockSampler:
    """Shader clock, board power and temperatures of the visible GPU over the timed region: the amdgpu hwmon files `freq1_input`, `power1_average` (or
    `power1_input`) and `temp{1,2,3}_input` (edge / junction / memory) of the card whose PCI address is HIP device `index`, read every 10 ms by a daemon
    thread (a few small file reads per sample).  `summary()` is None where the box has no such sysfs files; `board()` likewise (round-5 review, item 6: a 1 %
    move of the headline is unreadable without the power / thermal state next to the clock)."""

    def __init__(self, index):
        import glob
        self.path = None
        self.extra = {}                  # name -> (path, scale)
        self.samples = []
        self.extra_samples = {}
        self._stop = False
        self._thread = None
        try:
            p = torch.cuda.get_device_properties(index)
            bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for dev in sorted(glob.glob('/sys/class/drm/card*/device')):
                if os.path.basename(os.path.realpath(dev)) != bdf:
                    continue
                for h in sorted(glob.glob(os.path.join(dev, 'hwmon', 'hwmon*'))):
                    f = os.path.join(h, 'freq1_input')
                    if os.path.exists(f):
                        self.path = f
                    for name, files, scale in (('power_w', ('power1_average', 'power1_input'), 1e-6), ('temp_edge_c', ('temp1_input',), 1e-3),
                                               ('temp_junction_c', ('temp2_input',), 1e-3), ('temp_mem_c', ('temp3_input',), 1e-3)):
                        for fn in files:
                            q = os.path.join(h, fn)
                            if name not in self.extra and os.path.exists(q):
                                self.extra[name] = (q, scale)
        except Exception:                                        # noqa: BLE001
            self.path = None
        self.extra_samples = {k: [] for k in self.extra}

    def _run(self):
        i = 0
        while not self._stop:
            try:
                with open(self.path) as f:
                    self.samples.append(int(f.read().strip()) / 1e6)
            except Exception:                                    # noqa: BLE001
                pass
            if i % 5 == 0:                                       # power / temperature every 50 ms (they are averaged by the SMU anyway)
                for k, (q, scale) in self.extra.items():
                    try:
                        with open(q) as f:
                            self.extra_samples[k].append(int(f.read().strip()) * scale)
                    except Exception:                            # noqa: BLE001
                        pass
            i += 1
            time.sleep(0.01)

    def start(self):
        if self.path is not None:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {'median': round(v[len(v) // 2], 0), 'min': round(v[0], 0), 'max': round(v[-1], 0), 'samples': len(v),
                'source': 'amdgpu hwmon freq1_input of the visible device, 10 ms period, over the timed region'}

    def board(self):
        out = {}
        for k, v in self.extra_samples.items():
            if v:
                w = sorted(v)
                out[k] = {'median': round(w[len(w) // 2], 1), 'max': round(w[-1], 1), 'samples': len(w)}
        if not out:
            return None
        out['source'] = 'amdgpu hwmon power1_average / temp{1,2,3}_input of the visible device, 50 ms period, over the same region as sclk_mhz'
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='ssd300', choices=['ssd300', 'retinanet', 'yolov3', 'fcos', 'centernet'],
                    help="BASELINE.json's configuration: ssd300 = config 2 (the headline metric, the default); retinanet = config 3 (800x800, batch 16); "
                         'yolov3 = config 4 (416x416, 8 / GPU); fcos | centernet = config 5 (512x512, 16 / GPU) -- bench_configs.py')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the configuration\'s)')
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32', 'f32x3'], help='engine (default: the one the model class defaults to in training mode: bf16 for ssd300 / fcos / centernet, f32x3 -- f32 tensors, ODTK_F32X3 convolution descriptors -- for retinanet and, since round 6, yolov3)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-conv-events', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='N = 1, ssd300: skip what follows the headline measurement (sustained run, live PMC traffic passes, '
                                                             'test_one_image latency, the compact records of configurations 3-5)')
    ap.add_argument('--eager', action='store_true', help='no HIP-graph replay in the timed region (the default at N = 1 since round 3)')
    ap.add_argument('--launch-list', action='store_true', help="N = 1: use_graph='list' -- the step's C-ABI calls replayed from a recorded list of pre-bound argument "
                                                              'tuples (no Python wrappers, plain launches in stream order)')
    ap.add_argument('--graph', action='store_true', help='N = 1: replay forward + loss + backward from HIP graphs (A/B; measured 1-3 %% slower than eager launches)')
    ap.add_argument('--auto-launch', action='store_true', help="use_graph='auto': the faster of replay / eager launches, measured during warm-up")
    ap.add_argument('--wgrad-stream', action='store_true', help='A/B: filter gradients on a second HIP stream')
    ap.add_argument('--match-stream', action='store_true', help='A/B: box matching on a second HIP stream under the forward pass')
    ap.add_argument('--no-tail-stream', action='store_true', help='A/B: heads after the extra layers on one stream (round-1 order)')
    ap.add_argument('--no-fuse-pool', action='store_true', help='A/B: conv1_2 and pool1 as two launches (the un-pooled map is stored and read back)')
    ap.add_argument('--sync-bn', action='store_true', help='N > 1: batch-norm statistics over all replicas (SURVEY 8e option B; eager launches)')
    ap.add_argument('--conv-table', default=None, help='write the per-layer conv launch table (eager roofline pass) to this file')
    ap.add_argument('--kernel-dbg', type=int, default=0, help='A/B: odtk_debug_set(2, bits) dispatch switches of csrc/conv_v3.hip (bits >= 1<<26 only)')
    ap.add_argument('--debug-set', default='', help="A/B: comma list of KEY:VALUE for odtk_debug_set (keys that leave results intact: 3, 4, 5, 6, 7)")
    ap.add_argument('--model-cfg', default='', help="A/B: comma list of KEY=VALUE config overrides of the model class (e.g. heads_first=1)")
    ap.add_argument('--bucket-mb', type=int, default=25, help='N > 1: gradient all-reduce bucket size')
    ap.add_argument('--dp-world1', action='store_true',
                    help='N = 1 only: run the data-parallel path (RCCL process group of ONE rank, gradient buckets, per-bucket backward graphs, '
                         'an ncclAllReduce per bucket) on the one GPU -- the RCCL code path of N > 1 exercised where only one GPU exists')
    ap.add_argument('--collective', default='torch', choices=['torch', 'odtk'],
                    help="N > 1 / --dp-world1: the bucket all-reduce through torch.distributed (default) or through the C-ABI's own collective "
                         "(odtk_comm_allreduce: what a binder that is not PyTorch calls; RCCL underneath either way)")
    ap.add_argument('--grad-dtype', default='f32', choices=['f32', 'bf16'],
                    help='N > 1: all-reduce the gradient buckets as f32 (default, 105 MB/step) or as bf16 copies (52 MB/step)')
    ap.add_argument('--launch-check', action='store_true',
                    help='exercise ONLY the launcher / rendezvous / timing / one-JSON-line skeleton with a dummy all-reduce step '
                         '(gloo when there is no GPU); prints metric "launch-check", never a measurement')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], need_gpus=not args.launch_check))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; pass the same N to both '
                         f'(or drop the launcher: `python bench.py --gpus N` starts its own ranks)')
    if args.launch_check:
        return launch_check(args, world, rank, local_rank)
    import bench_configs as BC
    if args.batch is None:
        args.batch = BC.SHAPES[args.config][1]
    if args.dtype is None:
        args.dtype = BC.SHAPES[args.config][2]
    if args.config != 'ssd300':
        return bench_other(args, world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_pg = world > 1 or args.dp_world1
    if use_pg:
        init_group('nccl', rank, world, dev)

    import odtk
    from odtk import ops
    collective = args.collective
    if use_pg and collective == 'odtk' and not os.environ.get('ODTK_BENCH_LATE_COMM'):      # (the variable: communicator created at attach time, A/B)
        from odtk.dist import OdtkCollective
        collective = OdtkCollective(None, dev)                  # the communicator BEFORE the model's streams see their first launch (odtk/dist.py)
    B = args.batch
    apply_debug_switches(args)
    config = {
        'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
        'keep_prob': 0.5, 'batch_size': B, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20,
        'nms_iou_threshold': 0.5, 'pretraining_weight': os.path.join('.', 'vgg_16.ckpt'),
        'compute_dtype': args.dtype, 'verbose': False, 'seed': 0, 'wgrad_stream': args.wgrad_stream, 'match_stream': args.match_stream, 'tail_stream': not args.no_tail_stream, 'use_graph': 'auto' if args.auto_launch else ('list' if args.launch_list else bool(args.graph) and not args.eager), 'fuse_pool': not args.no_fuse_pool,
    }
    for item in filter(None, args.model_cfg.split(',')):             # A/B switches of the model class (tools/, profiles/)
        k, _, v = item.partition('=')
        config[k] = {'0': False, '1': True}.get(v, v)
    provider = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    model = odtk.SSD300(config, provider)
    if use_pg:
        model.attach_data_parallel(bucket_mb=args.bucket_mb, sync_bn=args.sync_bn, grad_dtype=args.grad_dtype, force_collectives=args.dp_world1,
                                   collective=collective)
    images, gt = synthetic_batch(B, 1000 + rank, dev)
    model.set_batch(images, gt)

    timer = ConvTimer(ops)
    timer.install()

    def barrier():
        torch.cuda.synchronize()
        if use_pg:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    lr = 0.01
    for _ in range(args.warmup):
        loss = model.train_step(lr)
    while model.use_graph and not args.eager and model._eager_steps < 2:
        loss = model.train_step(lr)                          # a capture needs every lazily allocated buffer to exist
    if model.use_graph and model._g_front is None and not args.eager:
        model._graphs_build_safe()                           # untimed; a capture executes nothing
    if args.eager:
        model.use_graph = False
        model._auto = None
    while model.launch_mode_pending:                         # use_graph='auto': replay vs eager launches, decided by measurement (untimed)
        loss = model.train_step(lr)
    import gc
    gc.collect()
    gc.disable()               # a generation-2 collection in the launching thread is a 10-30 ms host stall; with eager launches
    barrier()                  # (the HIP-event pass below) the GPU runs dry behind it and the stall lands in some kernel's events
    clock = ClockSampler(local_rank)
    clock.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_step(lr)
    barrier()
    dt = time.perf_counter() - t0
    clock.stop()
    final_loss_t = loss
    # roofline pass: the SAME step launched eagerly with HIP events around every conv launch on the launch
    # stream (graph replay cannot carry per-kernel events; the kernels and their durations are identical)
    if not args.no_conv_events:
        # ONE stream for this pass: the head stream, the tail filter-gradient stream, the optional filter-gradient stream and the side-stream front / tail
        # of the step are all switched off, so every conv kernel is timed with the chip to itself and the per-kernel table is not deflated by kernels
        # that share CUs (round 3 left the head / tail streams on: `conv_ms_per_step` exceeded `ms_per_step`).  The timed region above is untouched.
        saved = (model.use_graph, model.wgrad_stream, model._tail, model._twg, model.config.get('side_front', True))
        torch.cuda.synchronize()
        model.use_graph = False
        model.wgrad_stream = model._tail = model._twg = None
        model.config['side_front'] = False
        timer.enabled = True
        ev_steps = min(args.steps, 5) + 1
        t1 = time.perf_counter()
        for _ in range(ev_steps):
            model.train_step(lr)
        torch.cuda.synchronize()
        single_stream_ms = (time.perf_counter() - t1) / ev_steps * 1e3
        timer.enabled = False
        gc.enable()
        model.use_graph, model.wgrad_stream, model._tail, model._twg, model.config['side_front'] = saved
    gc.enable()
    loss = final_loss_t
    comm = None
    if use_pg:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        comm = comm_metrics(model, args, lr, barrier, dt / args.steps * 1e3, dev)
    final_loss = float(loss.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        out = {
            'metric': 'images/sec SSD300 VGG-16 batch=32 train',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'SSD300 VGG-16 300x300 train step, batch {B}/GPU (fwd + NMS-mined loss + bwd + SGD-momentum)',
                       'global_batch': B * world, 'parallelism': f'dp{world}' + ('+sync-bn' if args.sync_bn and world > 1 else ''), 'final_loss': round(final_loss, 4),
                       'launch_mode_calibration': model.launch_mode,
                       'launch': ('recorded launch list' if getattr(model, 'use_list', False) else 'eager') if not model.use_graph else ('hip-graph replay (fwd+loss+bwd)' if model._g_back is not None
                                                                        else f'hip-graph replay (fwd+loss; bwd as {len(model._g_back_segs or [])} '
                                                                             'bucket graphs with RCCL all-reduces between them)')},
        }
        peak = MFMA_PEAK_BF16 if args.dtype == 'bf16' else (MFMA_PEAK_BF16 / 3 if args.dtype == 'f32x3' else MFMA_PEAK_F32)
        if timer.records:
            out['roofline'] = timer.roofline(min(args.steps, 5) + 1, peak, value / world * 188.0e9 / peak)
            if args.conv_table:
                with open(args.conv_table, 'w') as f:
                    f.write(timer.table(min(args.steps, 5) + 1) + '\n')
            out['roofline']['measured_on'] = (f'{min(args.steps, 5) + 1} eager SINGLE-STREAM steps right after the timed region (head / tail / side streams off: '
                                              'every conv launch has the chip to itself; HIP events per conv launch on the launch stream; plain per-launch mean '
                                              'without the first eager step; *_trimmed additionally drops the slowest sample of every launch)')
            out['roofline']['single_stream_ms_per_step'] = round(single_stream_ms, 3)   # incl. the event records: an upper bound of conv_ms_per_step
        out['sclk_mhz'] = clock.summary()
        out['board'] = clock.board()
        if comm is not None:
            out['comm'] = comm
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        if world == 1 and not args.no_extras and not args.dp_world1:
            # Everything below runs AFTER the headline measurement and outside its timed region; the headline fields above are final.
            t_extras = time.perf_counter()
            try:
                out['sustained'] = sustained_run(model, lr, B, local_rank)
            except Exception as e:                                       # noqa: BLE001
                out['sustained'] = {'error': f'{type(e).__name__}: {e}'}
            if 'roofline' in out and not args.no_conv_events:
                out['roofline'].update(live_pmc_traffic(out['roofline']))
            try:
                out['inference'] = inference_bench(dev, with_cpu=not args.no_cpu_baseline)
            except Exception as e:                                       # noqa: BLE001
                out['inference'] = {'error': f'{type(e).__name__}: {e}'}
            try:
                out['epoch'] = epoch_bench(model, lr, B)
            except Exception as e:                                       # noqa: BLE001
                out['epoch'] = {'error': f'{type(e).__name__}: {e}'}
            out['configs'] = configs_bench(budget_s=300.0 - (time.perf_counter() - t_extras))
            out['extras_s'] = round(time.perf_counter() - t_extras, 1)
        print(json.dumps(out), flush=True)
    if use_pg:
        import torch.distributed as dist
        dist.destroy_process_group()


def sustained_run(model, lr, B, local_rank, min_steps=250):
    """The same eager step for >= 250 steps (~2 s: long enough for an outside sampler -- rocm-smi, the driver's gpu_busy -- to see the GPU busy, and for the
    clock / power state to settle) between two device synchronisations.  Reported NEXT to the headline, never instead of it."""
    torch.cuda.synchronize()
    clock = ClockSampler(local_rank)
    clock.start()
    t0 = time.perf_counter()
    for _ in range(min_steps):
        model.train_step(lr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clock.stop()
    return {'steps': min_steps, 'seconds': round(dt, 3), 'ms_per_step': round(dt / min_steps * 1e3, 3), 'images_per_sec': round(B * min_steps / dt, 1),
            'sclk_mhz': clock.summary(), 'board': clock.board()}


PMC_CHILD_STEPS, PMC_CHILD_WARMUP = 2, 3


def _rocprof_pass(counters, tag, timeout_s=150, child_args=()):
    """one `rocprofv3 --pmc` pass (counters only: no trace domains) over a 2-step child run of this script; -> {kernel short name: {counter: mean per dispatch, '_n': dispatches}}"""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    d = tempfile.mkdtemp(prefix=f'odtk_pmc_{tag}_', dir='/tmp')
    cmd = [exe, '--pmc'] + counters + ['-f', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__), '--steps', str(PMC_CHILD_STEPS), '--warmup', str(PMC_CHILD_WARMUP),
                                       '--no-cpu-baseline', '--no-conv-events', '--eager', '--no-extras'] + list(child_args)
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        if r.returncode != 0:
            return None, f'rocprofv3 exit code {r.returncode}: ' + r.stderr.decode(errors='replace')[-200:]
        acc = {}
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    k = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
                    k = re.sub(r'^void ', '', k).split('(')[0].split('<')[0].split('::')[-1]
                    e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, 0])
                    e[0] += float(row['Counter_Value']); e[1] += 1
        return {k: dict({c: v[0] / max(v[1], 1) for c, v in cs.items()}, _n=max(v[1] for v in cs.values())) for k, cs in acc.items()}, None
    except Exception as e:                                               # noqa: BLE001
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(d, ignore_errors=True)


def live_pmc_traffic(rf):
    """HBM bytes and MFMA-busy cycles per launch of the dominant kernel, measured NOW: two `rocprofv3 --pmc` child passes of this command at 2 steps
    (FETCH_SIZE + SQ_VALU_MFMA_BUSY_CYCLES; WRITE_SIZE -- the two TCC counters do not fit one pass), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes
    (KiB units; FETCH_SIZE doubled on gfx950 for wide coalesced reads).  Null with the reason when rocprofv3 is missing or a pass fails -- never a committed constant."""
    base = rf['kernel'].split('<')[0].split('+')[0]
    a, ea = _rocprof_pass(['FETCH_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'], 'a')
    b, eb = (None, 'skipped') if a is None else _rocprof_pass(['WRITE_SIZE'], 'b')
    if a is None or b is None or base not in a or base not in b:
        return {'traffic': None, 'traffic_note': 'no live PMC pass: ' + str(ea or eb or f'{base} not in the counter output')}
    rd = 2.0 * a[base]['FETCH_SIZE'] * 1024.0
    wr = b[base]['WRITE_SIZE'] * 1024.0
    t = rf['avg_launch_us'] * 1e-6
    return {'traffic': {'hbm_read_bytes': int(rd), 'hbm_write_bytes': int(wr), 'hbm_bytes': int(rd + wr), 'mfma_busy_cycles': a[base].get('SQ_VALU_MFMA_BUSY_CYCLES'),
                        'launches_counted': int(a[base]['_n']),
                        'source': 'THIS RUN: two rocprofv3 --pmc child passes of `bench.py --steps 2 --warmup 3 --eager` right after the timed region (mean per launch over every '
                                  f'template variant of {base}; FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024)'},
            'mfma_busy_pct': None if not a[base].get('SQ_VALU_MFMA_BUSY_CYCLES') else round(100.0 * a[base]['SQ_VALU_MFMA_BUSY_CYCLES'] / (t * 2.4e9 * 1024), 1),
            'hbm_gbps': round((rd + wr) / t / 1e9, 1), 'hbm_frac_of_8TBps': round((rd + wr) / t / 8e12, 4)}


def live_pmc_step(name, ms_per_step, budget_s):
    """Whole-step counters of a BASELINE configuration, measured NOW (round-5 review, item 6): two `rocprofv3 --pmc` child passes of `bench.py --config <name>` at
    2 + 3 steps; every dispatch of an odtk kernel is summed and divided by the steps the child ran (the one-off launches of the model's construction -- filter
    re-layout, casts -- are in the sum: < 1 %).  -> hbm_bytes_per_step (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, as MI355X_MICROARCH.md prescribes), the HBM
    rate that is of the step time, and mfma_busy_pct = sum of SQ_VALU_MFMA_BUSY_CYCLES / (step time x 2.4 GHz x 1024 SIMDs)."""
    t0 = time.perf_counter()
    steps = PMC_CHILD_STEPS + max(PMC_CHILD_WARMUP, 2)
    a, ea = _rocprof_pass(['FETCH_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'], name + '_a', timeout_s=max(20, min(90, budget_s)), child_args=['--config', name])
    if a is None:
        return {'pmc_note': 'no live PMC pass: ' + str(ea)}
    left = budget_s - (time.perf_counter() - t0)
    b, eb = (None, 'time budget of the default run spent') if left < 20 else _rocprof_pass(['WRITE_SIZE'], name + '_b', timeout_s=min(90, left), child_args=['--config', name])

    def total(res, counter):
        return sum(v.get(counter, 0.0) * v['_n'] for k, v in res.items() if not k.startswith(('vectorized_', 'elementwise_', '__amd', 'reduce_')))
    rd = 2.0 * total(a, 'FETCH_SIZE') * 1024.0 / steps
    busy = total(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / steps
    out = {'hbm_read_bytes_per_step': int(rd), 'mfma_busy_pct': round(100.0 * busy / (ms_per_step * 1e-3 * 2.4e9 * 1024), 1),
           'pmc_source': f'THIS RUN: rocprofv3 --pmc child passes of `bench.py --config {name} --steps {PMC_CHILD_STEPS} --warmup {PMC_CHILD_WARMUP} --eager`; sums over every '
                         'odtk kernel dispatch / steps run; mfma_busy_pct is relative to the un-profiled step time at 2.4 GHz x 1024 SIMDs'}
    if b is not None:
        wr = total(b, 'WRITE_SIZE') * 1024.0 / steps
        out.update({'hbm_write_bytes_per_step': int(wr), 'hbm_bytes_per_step': int(rd + wr), 'hbm_gbps_of_step': round((rd + wr) / (ms_per_step * 1e-3) / 1e9, 1),
                    'hbm_frac_of_8TBps': round((rd + wr) / (ms_per_step * 1e-3) / 8e12, 4)})
    else:
        out['pmc_note'] = 'WRITE_SIZE pass: ' + str(eb)
    return out


def epoch_bench(model, lr, B, steps=30):
    """What a user of the drop-in calls (round-5 review, item 6): `SSD300.train_one_epoch(lr)` (SSD300.py:473-484) through the PUBLIC method -- a host iterator
    yields (images f32 [B,300,300,3], ground truth) per step, set_batch copies them to the device, one train_step, one `.item()` of the loss per step -- on the
    benchmarked model, `steps` steps per variant: numpy arrays in pageable memory (what a tf.data replacement hands over) and torch tensors in pinned memory.
    `h2d_ms` is the copy of one batch alone (34.6 MB of f32 pixels + the ground truth), `h2d_share` its part of the epoch's step time.  The contract's headline
    (`value`) keeps the batch resident; this is the PCIe-inclusive rate next to it."""
    import numpy as np
    g = torch.Generator().manual_seed(77)
    imgs = torch.rand(B, 300, 300, 3, generator=g) * 255.
    _, gt = synthetic_batch(B, 1077, torch.device('cpu'))
    saved = (model.train_iterator, model.num_train, model.train_initializer, model.verbose)
    res = {'what': f'SSD300.train_one_epoch through the public method, {steps} steps per variant, host iterator -> set_batch (H2D) -> train_step -> float(loss)', 'steps': steps}
    try:
        for kind in ('numpy_pageable', 'torch_pinned'):
            if kind == 'numpy_pageable':
                hi, hg = np.ascontiguousarray(imgs.numpy()), np.ascontiguousarray(gt.cpu().numpy())
            else:
                hi, hg = imgs.pin_memory(), gt.cpu().pin_memory()
            model.train_iterator, model.num_train, model.train_initializer, model.verbose = [(hi, hg)] * steps, steps * B, None, False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mean_loss = model.train_one_epoch(lr)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            for _ in range(10):
                model.set_batch(hi, hg)
                torch.cuda.synchronize()
            h2d = (time.perf_counter() - t1) / 10
            res[kind] = {'images_per_sec': round(B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 3), 'h2d_ms': round(h2d * 1e3, 3),
                         'h2d_share': round(h2d / (dt / steps), 3), 'mean_loss': round(float(mean_loss), 4)}
    finally:
        model.train_iterator, model.num_train, model.train_initializer, model.verbose = saved
    return res


def inference_bench(dev, with_cpu=True, reps=20):
    """`SSD300.test_one_image` (SSD300.py:486-488; the one thing the reference times itself: YOLOv3.py:459-462): forward + decode + per-class NMS + the copy of the
    detections to the host, one 300 x 300 image, random-init weights (score threshold 0.5 as testSSD300.py), median wall time of `reps` calls per engine --
    'f32' is the class's default in test mode (exact f32 MFMA), 'f32x3' the operand-splitting engine, 'bf16' for contrast (it misses north_star's 1e-3)."""
    import numpy as np
    import odtk
    cfg = {'mode': 'test', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
           'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False, 'seed': 0}
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(1, 300, 300, 3, generator=g) * 255.).numpy()
    res = {'what': 'SSD300.test_one_image, 1 image 300x300, random-init weights; median ms of %d calls (host wall time incl. the device-to-host copy of the detections)' % reps}
    for engine in ('f32', 'f32x3', 'bf16'):
        m = odtk.SSD300(dict(cfg, compute_dtype=engine), None)
        for _ in range(3):
            out = m.test_one_image(img)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = m.test_one_image(img)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        res[engine] = {'ms': round(ts[len(ts) // 2], 3), 'min_ms': round(ts[0], 3), 'detections': int(len(out[0])), 'default_test_engine': engine == 'f32'}
        del m
    if with_cpu:
        from oracle import ssd300_ref as R
        cores = usable_cores()
        torch.set_num_threads(cores)
        p = R.init_params(0)
        x = torch.from_numpy(np.asarray(img))
        R.test_one_image(p, x)
        t0 = time.perf_counter()
        n = 0
        while n < 1 or (time.perf_counter() - t0 < 5 and n < 5):
            R.test_one_image(p, x)
            n += 1
        res['cpu_baseline'] = {'ms': round((time.perf_counter() - t0) / n * 1e3, 1), 'cores': cores, 'kind': 'port',
                               'sample': f'{n} call(s) of the PyTorch-CPU oracle\'s test_one_image on the same image'}
    return res


def configs_bench(budget_s=120.0, pmc=True):
    """BASELINE.json's configurations 3-5, one child `python bench.py --config <name> --steps 10 --warmup 3 --no-cpu-baseline` each (its own process: a failure there
    cannot touch the headline line), reduced to a compact record: images/s, ms/step, engine, dominant kernel and its fraction of the engine's MFMA peak, the conv
    family's fraction.  Stops starting children when the time budget of the default run is spent."""
    import subprocess
    res = {}
    t0 = time.perf_counter()
    # (retinanet twice: its default engine since round 4, f32x3, and the exact f32 engine it replaced -- the two are different arithmetic, not one kernel made faster;
    #  yolov3 twice since round 6: the class default f32x3 and the bf16 engine the gate does not admit, each line with its `engine_admission`)
    order = (('yolov3', []), ('yolov3_bf16', ['--dtype', 'bf16']), ('fcos', []), ('centernet', []), ('retinanet', []), ('retinanet_f32', ['--dtype', 'f32']))

    def remaining(name):                     # configurations still to be started after `name` (each needs ~25 s for its plain child run)
        names = [n for n, _ in order]
        return len(names) - 1 - names.index(name)
    for name, extra in order:
        left = budget_s - (time.perf_counter() - t0)
        if left < 25:
            res[name] = {'skipped': 'time budget of the default run spent'}
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--config', name.split('_')[0], '--steps', '10', '--warmup', '3', '--no-cpu-baseline'] + extra,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=min(left, 90))
            line = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not line:
                res[name] = {'error': f'exit code {r.returncode}: ' + r.stderr.decode(errors='replace')[-200:]}
                continue
            d = json.loads(line[-1])
            rf = d.get('roofline', {})
            res[name] = {'metric': d['metric'], 'images_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'dtype': d['dtype'], 'batch_per_gpu': d['config']['global_batch'],
                         'dominant_kernel': rf.get('kernel'), 'dominant_TFLOPs': rf.get('achieved'), 'dominant_frac': rf.get('frac'),
                         'conv_family_frac': rf.get('family', {}).get('frac'), 'conv_ms_per_step': rf.get('family', {}).get('conv_ms_per_step'),
                         'peak_TFLOPs': rf.get('peak'), 'final_loss': d['config'].get('final_loss'), 'engine_admission': d['config'].get('engine_admission')}
            if pmc and '_' not in name:
                left = budget_s - (time.perf_counter() - t0)
                res[name].update(live_pmc_step(name, d['ms_per_step'], left - 25 * remaining(name)) if left - 25 * remaining(name) >= 30 else
                                 {'pmc_note': 'time budget of the default run spent'})
        except Exception as e:                                           # noqa: BLE001
            res[name] = {'error': f'{type(e).__name__}: {e}'}
    return res


def apply_debug_switches(args):
    import odtk
    if args.kernel_dbg:
        assert (args.kernel_dbg & ~((1 << 15) | (1 << 16) | (1 << 17))) >> 26 << 26 == (args.kernel_dbg & ~((1 << 15) | (1 << 16) | (1 << 17))), 'only the dispatch switches leave results intact'
        odtk._lib.load().odtk_debug_set(2, args.kernel_dbg)
    for kv in filter(None, args.debug_set.split(',')):
        k, v = kv.split(':')
        assert int(k) in (3, 4, 5, 6, 7), 'only the dispatch switches leave results intact'  # (values may be negative: 4:-1)
        odtk._lib.load().odtk_debug_set(int(k), int(v))


def bench_other(args, world, rank, local_rank):
    """`--config retinanet | yolov3 | fcos | centernet`: the same contract for BASELINE.json's configurations 3-5 (one full training step of the model
    class at its stated shape on a resident synthetic batch; weak scaling, one rank per GPU, bucketed RCCL gradient all-reduce)."""
    import bench_configs as BC
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        init_group('nccl', rank, world, dev)
    from odtk import ops
    apply_debug_switches(args)
    name = args.config
    extra = {}
    for item in filter(None, args.model_cfg.split(',')):             # A/B switches of the model class
        k, _, v = item.partition('=')
        extra[k] = {'0': False, '1': True}.get(v, v)
    r = BC.make(name, batch=args.batch, dtype=args.dtype, seed=1000 + rank, use_graph=bool(args.graph), **extra)
    model, B, lr = r['model'], r['batch'], r['lr']
    if world > 1:
        model.attach_data_parallel(bucket_mb=args.bucket_mb, grad_dtype=args.grad_dtype)
    model.set_batch(r['images'], r['gt'])
    alg = {id(d): (cin, cout) for d, cin, cout, _ in BC.conv_layers(name, model)}
    timer = ConvTimer(ops, alg)
    timer.install()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):
        loss = model.train_step(lr)
    import gc
    gc.collect(); gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_step(lr)
    barrier()
    dt = time.perf_counter() - t0
    final_loss = float(loss.item() if hasattr(loss, 'item') else loss)
    ev_steps = min(args.steps, 3) + 1
    if not args.no_conv_events:
        timer.enabled = True
        for _ in range(ev_steps):
            model.train_step(lr)
        torch.cuda.synchronize()
        timer.enabled = False
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        value = B * world * args.steps / dt
        flops_step = BC.conv_flops_per_step(name, model)
        peak = MFMA_PEAK_BF16 if args.dtype == 'bf16' else (MFMA_PEAK_BF16 / 3 if args.dtype == 'f32x3' else MFMA_PEAK_F32)
        out = {'metric': BC.METRIC[name], 'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 2),
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
               'data': 'synthetic',
               'config': {'workload': BC.WORKLOAD[name].format(B=B), 'global_batch': B * world, 'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4),
                          'launch': 'hip-graph replay' if args.graph else 'eager', 'algorithmic_conv_gflop_per_image': round(flops_step / B / 1e9, 2),
                          'engine_admission': BC.ADMISSION.get((name, args.dtype)),
                          'engine_note': None if args.dtype == 'bf16' else
                          'f32 engine with operand splitting: every f32 product = three bf16 MFMA products (hi*hi + hi*lo + lo*hi), f32 accumulation; peak = bf16 dense / 3' if args.dtype == 'f32x3' else
                          'f32 engine (the class default): exact-f32 MFMA (v_mfma_f32_32x32x2_f32), peak 157.3 TFLOP/s'}}
        if timer.records:
            rf = timer.roofline(ev_steps, peak, value / world * (flops_step / B) / peak)
            if args.conv_table:
                with open(args.conv_table, 'w') as f:
                    f.write(timer.table(ev_steps) + '\n')
            rf['measured_on'] = f'{ev_steps} eager steps right after the timed region (HIP events per conv launch on the launch stream; plain per-launch mean without the first)'
            rf['peak_note'] = ('dense bf16 MFMA' if args.dtype == 'bf16' else 'dense bf16 MFMA / 3 (three bf16 products per f32 product)' if args.dtype == 'f32x3'
                               else 'f32-input MFMA = the f32 vector rate (MI355X_MICROARCH.md)')
            out['roofline'] = rf
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_other(name)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


def cpu_baseline_other(name):
    """The model's CPU oracle (oracle/<model>_net_ref.py: PyTorch-CPU restatement of the reference graph, pinned on two training steps of the reference's own
    class) timed on a BOUNDED sample of the same workload: full training steps at the configuration's resolution and a batch of 2 (one warm-up, then steps
    until ~20 s are spent, at most 3)."""
    import importlib
    import bench_configs as BC
    mod = importlib.import_module({'retinanet': 'oracle.retinanet_net_ref', 'yolov3': 'oracle.yolov3_net_ref', 'fcos': 'oracle.fcos_net_ref',
                                   'centernet': 'oracle.centernet_net_ref'}[name])
    cores = usable_cores()
    torch.set_num_threads(cores)
    size = BC.SHAPES[name][0]
    bs = 2
    images, gt = BC.synthetic_batch(name, bs, size, 7)
    p = mod.init_params(0)
    state = {'t': 0, 'm': {}, 'v': {}} if name == 'centernet' else {k: torch.zeros_like(v) for k, v in p.items()}
    lr = BC.SHAPES[name][3]
    t_w = time.perf_counter()
    mod.train_step(p, state, images, gt, lr)                      # warm-up
    t_w = time.perf_counter() - t_w
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 + t_w < 25 and n < 3):
        mod.train_step(p, state, images, gt, lr)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(bs * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'mkldnn': bool(torch.backends.mkldnn.is_available() and torch.backends.mkldnn.enabled),
            'sample': f'{n} full train step(s) at {size}x{size}, batch {bs} after one warm-up step (same synthetic generator), PyTorch-CPU fp32 oracle '
                      f'({mod.__name__}); the reference TF-1.13 graph itself cannot run here (no tensorflow)'}


def _free_port():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(n, argv, need_gpus=True):
    """`python bench.py --gpus N` without torchrun: start N copies of this script, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment, exactly what torch.distributed.run sets), rank 0 inherits stdout so that its ONE
    JSON line is this process's output.  Returns the exit code (first failing rank's; the others are then terminated by PID)."""
    import subprocess
    if need_gpus and (not torch.cuda.is_available() or torch.cuda.device_count() < n):
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        print(f'bench.py: --gpus {n} but only {have} GPU(s) visible on this node', file=sys.stderr)
        return 2
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), ODTK_BENCH_LAUNCHER='self')
    env0.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL needs it)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in live:                     # one rank failed: the others would hang in the next collective
                    q.terminate()
        time.sleep(0.05)
    return rc


def launch_check(args, world, rank, local_rank):
    """--launch-check: the launcher, the rendezvous, barrier + max-over-ranks timing and the one-JSON-line contract with a dummy
    step (an all-reduce of 1 M floats).  gloo on a box without GPUs -- this is what tests/test_bench_launch_cpu.py runs."""
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    dev = torch.device('cuda', local_rank) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(local_rank)
    # on a GPU box the process group is RCCL even for ONE rank: that exercises RCCL's init against this driver (device_id binding, the dmabuf
    # IPC setting HSA_ENABLE_IPC_MODE_LEGACY=0, the watchdog thread) and a real ncclAllReduce on the one GPU a builder box has
    use_pg = world > 1 or use_gpu
    if use_pg:
        init_group('nccl' if use_gpu else 'gloo', rank, world, dev if use_gpu else None)
    buf = torch.ones(1 << 20, device=dev)

    def barrier():
        if use_gpu:
            torch.cuda.synchronize()
        if use_pg:
            dist.barrier()

    def step():
        if use_pg:
            dist.all_reduce(buf)
            buf.mul_(1.0 / world)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_pg:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ok = bool(torch.allclose(buf, torch.ones_like(buf)))
    if rank == 0:
        print(json.dumps({'metric': 'launch-check', 'value': round(args.steps / dt, 2), 'unit': 'dummy steps/sec', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'launcher / rendezvous self-check (1 M-float all-reduce per step); NOT a measurement'},
                          'comm': {'backend': dist.get_backend() if use_pg else None,
                                   'world_size': dist.get_world_size() if use_pg else 1,
                                   'launcher': os.environ.get('ODTK_BENCH_LAUNCHER', 'external'), 'allreduce_ok': ok}}), flush=True)
    if use_pg:
        dist.destroy_process_group()
    return 0 if ok else 1


def init_group(backend, rank, world, dev):
    """torch.distributed rendezvous on 127.0.0.1 (the container hostname may not resolve); a failure is reported on stderr BY EVERY RANK with
    what it tried, so that a dead N-GPU run names its cause instead of timing out silently."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver (RCCL needs it)
    try:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300),
                                **({'device_id': dev} if dev is not None else {}))
        if backend == 'nccl':
            # first collective = communicator creation: fail here, with a message, not inside the first training step
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert float(probe.item()) == float(world), probe
    except Exception as e:                                        # noqa: BLE001
        print(f'bench.py: rank {rank}/{world}: {backend} rendezvous / communicator set-up failed at '
              f'{os.environ["MASTER_ADDR"]}:{os.environ["MASTER_PORT"]} (device {dev}, HSA_ENABLE_IPC_MODE_LEGACY='
              f'{os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}): {type(e).__name__}: {e}', file=sys.stderr, flush=True)
        raise


def comm_metrics(model, args, lr, barrier, ms_step, dev):
    """N > 1, after the timed region: what RCCL saw (world size, backend), the gradient all-reduce's own time per step
    (every bucket back to back, nothing to overlap with), the step time with the collectives switched off, and from the two
    the fraction of the all-reduce hidden under backward:  overlap = 1 - (step - step_without_comm) / allreduce_alone."""
    import torch.distributed as dist
    red = model.dist.red
    k = max(3, min(args.steps, 10))
    for _ in range(2):
        red.all_reduce_alone()
    barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        red.all_reduce_alone()
    barrier()
    t_ar = (time.perf_counter() - t0) / k * 1e3
    # the steps without collectives train every replica on its own batch only: the replicas' weights would drift apart, so the model state
    # (parameters, momentum, moving statistics, operand copies) is snapshotted here and put back afterwards -- these steps are UNTIMED
    # diagnostics and leave no trace in the model
    snap = {k: getattr(model, k).clone() for k in ('P', 'Mom', 'S', 'Pc') if isinstance(getattr(model, k, None), torch.Tensor)}
    gstep = model.global_step
    red.enabled = False
    model.train_step(lr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        model.train_step(lr)
    barrier()
    t_nc = (time.perf_counter() - t0) / k * 1e3
    red.enabled = True
    for k_, v_ in snap.items():
        getattr(model, k_).copy_(v_)
    model.global_step = gstep
    model.refresh_wt()
    v = torch.tensor([t_ar, t_nc], device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    t_ar, t_nc = (float(x) for x in v.tolist())
    nbytes = sum(red.bucket_bytes())
    w = dist.get_world_size()
    return {'backend': dist.get_backend(), 'world_size': w, 'launcher': os.environ.get('ODTK_BENCH_LAUNCHER', 'external'),
            'buckets': len(red.buckets), 'gradient_mb': round(nbytes / 1e6, 1),
            'allreduce_ms_per_step': round(t_ar, 3),
            'allreduce_busbw_gbps': round(nbytes * 2 * (w - 1) / w / (t_ar * 1e-3) / 1e9, 1),
            'gradient_dtype': getattr(red, 'comm_dtype', 'f32'),
            'collective': 'odtk_comm_allreduce (C-ABI over RCCL)' if getattr(red, 'collective', None) is not None else 'torch.distributed all_reduce',
            'ms_per_step_without_comm': round(t_nc, 3), 'without_comm_steps': 'untimed diagnostic steps; model state restored afterwards',
            'exposed_comm_ms': round(max(ms_step - t_nc, 0.0), 3),
            'overlap_frac': round(min(max(1.0 - max(ms_step - t_nc, 0.0) / max(t_ar, 1e-9), 0.0), 1.0), 3)}


def usable_cores():
    """Host threads this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:                                             # noqa: BLE001
        pass
    return n


def cpu_baseline():
    """PyTorch-CPU oracle (restatement of the reference graph) on a bounded sample of the SAME workload: full training steps
    at batch 32 (one warm-up, then steps until ~20 s are spent, at most 3)."""
    from oracle import ssd300_ref as R
    cores = usable_cores()
    torch.set_num_threads(cores)
    bs = 32
    p = R.init_params(0)
    mom = {k: torch.zeros_like(v) for k, v in p.items()}
    imgs, gt = R.synthetic_batch(bs, 0)
    anchors = R.priors()
    R.train_step(p, mom, imgs, gt, 0.01, 1e-4, anchors)          # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < 20 and n < 3):
        R.train_step(p, mom, imgs, gt, 0.01, 1e-4, anchors)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(bs * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'mkldnn': bool(torch.backends.mkldnn.is_available() and torch.backends.mkldnn.enabled),
            'sample': f'{n} full train step(s) at batch {bs} after one warm-up step (same synthetic generator), PyTorch-CPU fp32 '
                      'oracle; the reference TF-1.13 graph itself cannot run (no tensorflow, SSD300.py:41-43 syntax error)'}


if __name__ == '__main__':
    main()
