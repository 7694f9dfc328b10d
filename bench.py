#!/usr/bin/env python
"""Benchmark of the SSD300 VGG-16 training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without a launcher it starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one full training step (preprocess + forward + loss incl. hard-negative-mining NMS +
backward + RCCL gradient all-reduce + fused SGD-momentum) on a synthetic VOC-shaped batch of 32
images/GPU already resident in HBM.  Weak scaling: every rank keeps batch 32 (each replica is
exactly the reference computation, local BatchNorm; SURVEY.md 8e option A).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the DOMINANT device kernel of the step (largest summed duration; the 8-wave
                  implicit-GEMM gather kernel that serves conv forward and dgrad), HIP-event timed
                  inside the timed region on the launch stream: algorithmic FLOPs per launch / average
                  launch duration vs the dense bf16 MFMA peak (2.5 PFLOP/s).  `traffic` = HBM bytes
                  per launch of that kernel from the committed rocprofv3 PMC passes (profiles/*.json,
                  FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE).  `family` = the same
                  ratio over every conv kernel (fwd + dgrad + wgrad, all layers).
  cpu_baseline -- the CPU oracle (PyTorch-CPU restatement of the reference graph; TF 1.13 cannot
                  run here) timed on this box's host cores on a bounded sample (N=1 only).
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

    def __init__(self, ops):
        self.ops = ops
        self.records = []            # (kind, kernel, flops, start_evt, end_evt, algorithmic bytes)
        self.enabled = False
        self._orig = {}

    def install(self):
        ops = self.ops
        for kind in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'):
            self._orig[kind] = getattr(ops, kind)
            setattr(ops, kind, self._wrap(kind))

    def _wrap(self, kind):
        orig = self._orig[kind]

        def f(d, *args):
            if not self.enabled:
                return orig(d, *args)
            c_alg = 3 if (d.H == 300 and d.C <= 8) else d.C          # conv1_1: 3 real input channels (padded to one chunk)
            flops = 2.0 * d.N * d.Ho * d.Wo * d.K * d.R * d.S * c_alg    # algorithmic: 2*M*Cout*R*S*Cin for each pass
            esz = 2 if d.dtype == 0 else 4
            # algorithmic HBM bytes: every operand once (x, y / dy, dx as bf16; filter as bf16, dW as f32)
            abytes = (d.N * d.H * d.W * d.C + d.N * d.Ho * d.Wo * d.K) * esz + d.K * d.R * d.S * d.C * (4 if kind == 'conv2d_wgrad' else esz)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(d, *args)
            e.record()
            self.records.append((kind, self.ops.conv_last_kernel(), flops, s, e, abytes,
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
        pmc = pmc_traffic(dom)
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


def pmc_traffic(kernel):
    """HBM bytes and MFMA-busy cycles per launch of `kernel` from the newest committed PMC summary (profiles/*pmc*.json): the dispatch-weighted
    mean over the template variants of that kernel (the bench groups launches by the name odtk_conv_last_kernel reports, without template arguments),
    or None when no profile names the kernel."""
    import glob
    base = kernel.split('<')[0]
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc*.json')), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:                                         # noqa: BLE001
            continue
        rows = [(name, e) for name, e in d.items()
                if name.split('<')[0].split('::')[-1] == base and 'hbm_read_bytes_corrected' in e and 'hbm_write_bytes' in e and e.get('_dispatches', 0) > 0]
        if rows:
            n = sum(e['_dispatches'] for _, e in rows)
            avg = lambda key: sum(e.get(key, 0.0) * e['_dispatches'] for _, e in rows) / n        # noqa: E731
            rd, wr = avg('hbm_read_bytes_corrected'), avg('hbm_write_bytes')
            return {'hbm_read_bytes': int(rd), 'hbm_write_bytes': int(wr), 'hbm_bytes': int(rd + wr),
                    'mfma_busy_cycles': avg('SQ_VALU_MFMA_BUSY_CYCLES'),
                    'source': 'STATIC (committed rocprofv3 --pmc passes of this command, not re-measured in this run): '
                              + os.path.relpath(f, ROOT) + f' :: dispatch-weighted mean over {len(rows)} template variant(s) of {base}, {n} launches'}
    return None
