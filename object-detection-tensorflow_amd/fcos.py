"""FCOS (the reference's group-normalised pre-activation ResNet + pyramid + per-level heads) behind the reference's class surface, on libodtk.

Reference: /root/reference/FCOS.py
  * constructor, config keys ............. :12-49    (blocks 3, 4, 6, 3 and filters 16 * 2^i are fixed in the class, :29-31)
  * input ................................ :51-68    (images - mean; test mode feeds the tensor after the subtraction)
  * network .............................. :70-110, :350-382, :438-513: EVERY normalisation is tf.contrib.layers.group_norm(groups=8);
                                           stem conv + GN + ReLU + 3x3 / s2 max pool; bottleneck units [GN-ReLU-1x1 f, GN-ReLU-3x3 f (stride),
                                           GN-ReLU-1x1 4f] + [GN-ReLU-3x3 4f (stride)] shortcut; c3 / c4 / c5 1x1; bilinear top-down pyramid (the sum is
                                           handed down); p6 / p7; per level 4 x 3x3 -> classes and centre-ness, 4 x 3x3 -> exp(distances), the head WEIGHTS shared by the
                                           levels (variable_scope(..., reuse=tf.AUTO_REUSE), :351, :358)
  * loss, optimizer ...................... :111-192  (odtk_fcos_loss; mean over images + wd * l2; Momentum 0.9)
  * inference ............................ :193-265  (heads.fcos_detect; classes 0 .. C-2, sic)
  * train / test / checkpoints ........... :401-436
Same conventions as retinanet.py: layers l0 .. l85 in creation order (layer k = conv k + group norm k; l75 .. l85 are the heads, ONE set of
weights applied at all five pyramid levels -- instance names l<k>@<level>), one flat f32 parameter buffer.
Group norm has no batch statistics: nothing like moving averages exists, and data parallel needs no sync-BN.
"""
from __future__ import annotations

import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import heads, ops
from ._lib import BF16, F32, F32X3
from .warmup import F32Warmup

MEAN_RGB = (123.68, 116.779, 103.979)
BLOCKS = (3, 4, 6, 3)                                          # FCOS.py:30
FILTERS = (16, 32, 64, 128)                                    # :31
GROUPS = 8                                                     # :441
PI = 0.01                                                      # :486


def layer_specs(num_classes):
    """[(name, cin, cout, k, stride, gn_channels, bias_init)] in creation order (FCOS.py:70-110, :350-382, :504-513)"""
    specs = []

    def add(cin, cout, k, s, bias_init=0.):
        specs.append((f'l{len(specs)}', cin, cout, k, s, cin if specs else cout, bias_init))
        return cout
    c = add(3, 16, 7, 2)
    stage_out = []
    for i, blocks in enumerate(BLOCKS):
        f = FILTERS[i]
        for j in range(blocks):
            s = 2 if (i > 0 and j == 0) else 1
            add(c, f, 1, 1); add(f, f, 3, s); add(f, 4 * f, 1, 1)
            add(c, 4 * f, 3, s)
            c = 4 * f
        stage_out.append(c)
    e3, e4, e5 = stage_out[-3:]
    add(e3, 256, 1, 1); add(e4, 256, 1, 1); add(e5, 256, 1, 1)
    add(256, 256, 3, 1)
    add(256, 256, 1, 1); add(256, 256, 3, 1)
    add(256, 256, 1, 1); add(256, 256, 3, 1)
    add(256, 256, 3, 2); add(256, 256, 3, 2)
    bias = -math.log((1 - PI) / PI)
    # ONE set of head layers for all five levels: _detect_head enters variable_scope('classifier_head' / 'regress_head',
    # reuse=tf.AUTO_REUSE) once per level (FCOS.py:351, :358); leaving a variable scope resets the default-name counters of its
    # sub-scopes (variable_scope.py, close_variable_subscopes), so every level asks for conv2d, conv2d_1, ... and GroupNorm,
    # GroupNorm_1, ... again and AUTO_REUSE hands back the variables of the first level: convs AND group norms are shared.
    for _ in range(4):
        add(256, 256, 3, 1)
    add(256, num_classes, 3, 1, bias)
    add(256, 1, 3, 1, bias)
    for _ in range(4):
        add(256, 256, 3, 1)
    add(256, 4, 3, 1)
    return specs


HEAD_LAYERS = 11                                               # the last 11 specs: 6 classifier-head + 5 regress-head layers, shared by the levels


class _Act:
    def __init__(self, name, N, H, W, C, ld, dtype, dev):
        self.name, self.N, self.H, self.W, self.C, self.ld = name, N, H, W, C, ld
        self.M = N * H * W
        self.t = torch.zeros(self.M, ld, dtype=dtype, device=dev)
        self.gid = name


class FCOS(F32Warmup):
    def __init__(self, config, data_provider):
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.data_shape = config['data_shape']
        self.num_classes = config['num_classes']
        self.weight_decay = config['weight_decay']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.verbose = bool(config.get('verbose', True))
        self.dev = torch.device(config.get('device', 'cuda:0'))
        # engine: bf16 by default on the GPU since round 3 (warmup.py: the first f32_warmup_steps optimizer steps of a run from random initialisation go through
        # an f32 twin); an explicit 'compute_dtype' is taken literally; the CPU stand-in of the library (host-logic tests) stays on f32
        # (mode 'test' keeps f32 unless asked otherwise, as ssd300.py does: the bf16 gate checks training gradients, not thresholded detections)
        engine = config.get('compute_dtype', 'bf16' if (self.dev.type == 'cuda' and self.mode == 'train') else 'f32')
        # 'f32x3': f32 tensors, convolution descriptors of dtype ODTK_F32X3 (three bf16 MFMA products per f32 product where that is faster: include/odtk.h)
        self.DT = {'bf16': BF16, 'f32': F32, 'f32x3': F32}[engine]
        self.CDT = F32X3 if engine == 'f32x3' else self.DT
        self.tdt = torch.bfloat16 if self.DT == BF16 else torch.float32
        self.chunk = ops.chunk(self.DT)
        if self.mode == 'train':
            self.num_train = data_provider['num_train']
            self.num_val = data_provider['num_val']
            self.train_generator = data_provider['train_generator']
            if isinstance(self.train_generator, tuple) and len(self.train_generator) == 2:
                self.train_initializer, self.train_iterator = self.train_generator
            else:
                self.train_initializer, self.train_iterator = None, self.train_generator
            if data_provider.get('val_generator') is not None:
                self.val_generator = data_provider['val_generator']
        self.global_step = 0
        self.dist = None
        self.loss_divisor_batch = self.batch_size
        if self.dev.type == 'cuda':          # (a 'cpu' device only gets past ops._p with the mocked library of tests/mock_ops.py: host-logic tests)
            torch.cuda.set_device(self.dev)
        self.specs = layer_specs(self.num_classes)
        self._init_parameters(int(config.get('seed', 0)))
        self._build()
        self._warmup_setup(config, data_provider, 'compute_dtype' in config)

    # ------------------------------------------------------------------ parameters
    def param_layout(self):
        pinfo = OrderedDict()
        off = 0
        for name, cin, cout, k, _, gnc, _ in self.specs:
            for suffix, shape in (('.w', (cout, k, k, ops.pad_to(cin, self.chunk))), ('.b', (cout,)), ('.gamma', (gnc,)), ('.beta', (gnc,))):
                pinfo[name + suffix] = (off, shape)
                off += ops.pad_to(int(np.prod(shape)), 64)
        return pinfo, off

    def _init_parameters(self, seed):
        self.pinfo, off = self.param_layout()
        self.nparam = off
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.Mom = torch.zeros(off, device=dev)
        self.G = torch.zeros(off, device=dev)
        self.Pc = torch.zeros(off, dtype=self.tdt, device=dev) if self.DT == BF16 else self.P
        self.l2_partial = torch.zeros(ops.sgd_blocks(off), device=dev)
        self.l2_sum = torch.zeros(1, device=dev)
        self._cin = {s[0]: s[1] for s in self.specs}
        g = torch.Generator().manual_seed(seed)
        for name, cin, cout, k, _, _, bias_init in self.specs:
            self.set_param(name + '.w', torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k)))
            self.param(name + '.b').fill_(float(bias_init))
            self.param(name + '.gamma').fill_(1.0)

    def param(self, name, buf=None):
        off, shape = self.pinfo[name]
        buf = self.P if buf is None else buf
        return buf[off: off + int(np.prod(shape))].view(shape)

    def _flat(self, name, buf):
        off, shape = self.pinfo[name]
        return buf[off: off + int(np.prod(shape))]

    def set_param(self, name, value):
        dst = self.param(name)
        value = torch.as_tensor(value, dtype=torch.float32)
        if name.endswith('.w'):
            dst.zero_()
            dst[..., : value.shape[-1]] = value.to(self.dev)
        else:
            dst.copy_(value.to(self.dev).view(dst.shape))

    def get_param(self, name, buf=None):
        v = self.param(name, buf).detach().cpu().clone()
        if name.endswith('.w'):
            v = v[..., : self._cin[name[:-2]]].contiguous()
        return v

    def load_oracle_params(self, p):
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        for k, v in p.items():
            if k in self.pinfo:
                self.set_param(k, v)
        self._refresh_operand_copies()

    def export_params(self):
        self._sync_from_twin()
        return OrderedDict((k, self.get_param(k)) for k in self.pinfo)

    def _refresh_operand_copies(self):
        if self.DT == BF16:
            ops.cast_from_f32(self.P, self.Pc)
        if getattr(self, '_fp_batch', None) is not None:
            self._fp_batch.run()

    # ------------------------------------------------------------------ the graph
    def _build(self):
        N, dev, dt, ch = self.batch_size, self.dev, self.tdt, self.chunk
        H, W, _ = self.data_shape
        self.images = torch.zeros(N, H, W, 3, device=dev)
        c0 = ops.pad_to(3, ch)
        self.input = _Act('input', N, H, W, 3, c0, dt, dev)
        self.plan, self.desc, self.gnsave, self.acts = [], {}, {}, {}
        it = iter(self.specs)
        self._max_scr = 0
        groups = {}

        def find(g):
            while groups.setdefault(g, g) != g:
                g = groups[g]
            return g
        self.find = find

        def act(name, H_, W_, C_):
            a = _Act(name, N, H_, W_, C_, ops.pad_to(C_, ch), dt, dev)
            self.acts[name] = a
            return a

        def conv_desc(name, src, cout, k, stride, ldy):
            d = ops.conv_desc(N, src.H, src.W, src.ld, src.ld, cout, ldy, k, stride, 1, self.CDT, self.CDT)
            self.desc[name] = d
            return d

        def gnconv(x, spec=None, level=None):
            """group norm -> ReLU -> conv(bias): returns the conv output.  `spec`, `level`: a shared head layer applied at one
            pyramid level -- parameters under the layer's name `pname`, buffers / descriptor under the instance name pname@level"""
            pname, cin, cout, k, stride, gnc, _ = next(it) if spec is None else spec
            name = pname if level is None else f'{pname}@{level}'
            assert cin == x.C == gnc and gnc % GROUPS == 0, (name, cin, x.C)
            y = act(name + '.y', x.H, x.W, x.C)
            d = conv_desc(name, y, cout, k, stride, ops.pad_to(cout, ch))
            out = act(name, d.Ho, d.Wo, cout)
            self.gnsave[name] = torch.zeros(N, GROUPS, 2, device=dev)
            self._max_scr = max(self._max_scr, y.M * y.ld)
            self.plan.append(('gnconv', name, x, y, out, pname))
            return out

        def add(a, b):
            y = act(f'sum{len(self.plan)}', a.H, a.W, a.C)
            groups[find(a.gid)] = find(y.gid)
            groups[find(b.gid)] = find(y.gid)
            self.plan.append(('add', a, b, y))
            return y

        def resize_add(lat, top):
            y = act(f'total{len(self.plan)}', lat.H, lat.W, lat.C)
            groups[find(lat.gid)] = find(y.gid)
            self.plan.append(('resize_add', lat, top, y))
            return y

        name, cin, cout, k, stride, gnc, _ = next(it)
        d = conv_desc(name, self.input, cout, k, stride, ops.pad_to(cout, ch))
        z = act(name + '.z', d.Ho, d.Wo, cout)
        y = act(name, d.Ho, d.Wo, cout)
        self.gnsave[name] = torch.zeros(N, GROUPS, 2, device=dev)
        self._max_scr = max(self._max_scr, z.M * z.ld)
        self.plan.append(('stem', name, self.input, z, y))
        Hp, pt, _ = ops.same_pad(y.H, 3, 2)
        Wp, pl, _ = ops.same_pad(y.W, 3, 2)
        x = act('pool1', Hp, Wp, cout)
        self.plan.append(('pool', y, x, 3, 2, pt, pl))
        feats = []
        for blocks in BLOCKS:
            for _ in range(blocks):
                branch = gnconv(gnconv(gnconv(x)))
                x = add(branch, gnconv(x))
            feats.append(x)
        c3, c4, c5 = gnconv(feats[-3]), gnconv(feats[-2]), gnconv(feats[-1])
        p5 = gnconv(c5)
        total4 = resize_add(gnconv(c4), p5)
        p4 = gnconv(total4)
        total3 = resize_add(gnconv(c3), total4)
        p3 = gnconv(total3)
        p6 = gnconv(p5)
        p7 = gnconv(p6)
        self.levels = [p3, p4, p5, p6, p7]
        self.conf = [torch.zeros(N, a.H, a.W, self.num_classes, device=dev) for a in self.levels]
        self.reg = [torch.zeros(N, a.H, a.W, 4, device=dev) for a in self.levels]
        self.center = [torch.zeros(N, a.H, a.W, 1, device=dev) for a in self.levels]
        head = list(it)                                         # the 11 head layers: one set of weights, five levels
        assert len(head) == HEAD_LAYERS
        for l, lvl in enumerate(self.levels):
            c = lvl
            for j in range(4):
                c = gnconv(c, head[j], l)
            self.plan.append(('pred', gnconv(c, head[4], l), self.conf, l, False))
            self.plan.append(('pred', gnconv(c, head[5], l), self.center, l, False))
            r = lvl
            for j in range(4):
                r = gnconv(r, head[6 + j], l)
            self.plan.append(('pred', gnconv(r, head[10], l), self.reg, l, True))       # tf.exp on the distances (FCOS.py:363)
        self.wt, entries = {}, []
        for name, cin, cout, k, _, _, _ in self.specs[1:]:
            kp, cpad = ops.pad_to(cout, ch), ops.pad_to(cin, ch)        # dgrad-layout filter: one per PARAMETER (shared by the levels)
            self.wt[name] = torch.zeros(cpad * k * k * kp, dtype=dt, device=dev)
            entries.append((self._flat(name + '.w', self.P), self.wt[name], cout, k, k, cpad, kp))
        self._fp_batch = ops.FilterPrepareBatch(entries, self.DT, dev)
        if self.mode == 'train':
            self._build_backward(N, dt, dev)
        self._refresh_operand_copies()

    def _build_backward(self, N, dt, dev):
        find = self.find
        self.dconf = [torch.zeros_like(t) for t in self.conf]
        self.dreg = [torch.zeros_like(t) for t in self.reg]
        self.dcenter = [torch.zeros_like(t) for t in self.center]
        self.scr_y = torch.zeros(self._max_scr, dtype=dt, device=dev)          # d(relu(gn(x))): lives inside one layer
        self.gn_ws = ops.gn_workspace(N, max(s[5] for s in self.specs), dev)
        written = set()
        seen_params = set()
        self.bplan = []
        for op in reversed(self.plan):
            kind = op[0]
            if kind == 'pred':
                self.bplan.append(op)
                written.add(find(op[1].gid))
            elif kind == 'gnconv':
                _, name, x, y, out, pname = op
                assert find(out.gid) in written, name
                # a shared head layer: its norm's d(gamma), d(beta) accumulate over the levels (the filter / bias gradients always
                # accumulate); the layer is final -- for the gradient all-reduce -- after the LAST level processed (level 0)
                first_use = pname not in seen_params
                seen_params.add(pname)
                last_use = name == pname or name.endswith('@0')
                self.bplan.append(('gnconv', name, x, y, out, find(x.gid) in written, pname, not first_use, last_use))
                written.add(find(x.gid))
            elif kind == 'add':
                assert find(op[3].gid) in written
            elif kind == 'resize_add':
                _, lat, top, y = op
                assert find(y.gid) in written
                self.bplan.append(('resize_add', lat, top, y, find(top.gid) in written))
                written.add(find(top.gid))
            elif kind == 'pool':
                _, x, y, k, s, pt, pl = op
                assert find(y.gid) in written
                self.bplan.append(op)
                written.add(find(x.gid))
            else:
                self.bplan.append(op)
        self.g = {}
        for a in self.acts.values():
            gid = find(a.gid)
            if gid in written and gid not in self.g:
                self.g[gid] = torch.zeros(a.M, a.ld, dtype=dt, device=dev)
        self.loss_img = torch.zeros(N, device=dev)
        self.loss_ws = ops.fcos_workspace(self.conf, N, dev)
        self.gt = None

    def grad_of(self, a):
        return self.g[self.find(a.gid)]

    # ------------------------------------------------------------------ forward / loss / backward
    def _gn_relu(self, name, x, y, pname=None):
        pname = pname or name
        ops.gn_fwd(x.t, x.ld, y.t, y.ld, x.N, x.H * x.W, x.C, GROUPS, self.param(pname + '.gamma'), self.param(pname + '.beta'), 1, self.gnsave[name])

    def _forward(self, subtract_mean=True):
        ops.preprocess(self.images, MEAN_RGB if subtract_mean else (0., 0., 0.), self.input.ld, self.DT, self.input.t)
        for op in self.plan:
            kind = op[0]
            if kind == 'gnconv':
                _, name, x, y, out, pname = op
                self._gn_relu(name, x, y, pname)
                ops.conv2d_fwd(self.desc[name], y.t, self._flat(pname + '.w', self.Pc), self.param(pname + '.b'), out.t, False)
            elif kind == 'add':
                _, a, b, y = op
                ops.add2d(a.t, a.ld, b.t, b.ld, y.t, y.ld, y.M, y.ld)
            elif kind == 'resize_add':
                _, lat, top, y = op
                ops.add2d(lat.t, lat.ld, None, 0, y.t, y.ld, y.M, y.ld)
                ops.resize_bilinear_fwd(top.t, top.ld, y.t, y.ld, top.N, top.H, top.W, y.H, y.W, top.C, True)
            elif kind == 'pred':
                _, c, targets, l, is_exp = op
                if is_exp:
                    ops.exp_rows_to_f32(c.t, c.ld, targets[l], c.M, c.C)
                else:
                    ops.rows_to_f32(c.t, c.ld, targets[l], c.C, c.M, 0, c.M, c.C)
            elif kind == 'stem':
                _, name, src, z, y = op
                ops.conv2d_fwd(self.desc[name], src.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), z.t, False)
                self._gn_relu(name, z, y)
            else:
                _, x, y, k, s, pt, pl = op
                ops.maxpool_fwd(x.t, y.t, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pl)

    def _loss(self, grad_scale):
        ops.fcos_loss(self.conf, self.reg, self.center, self.gt, grad_scale, self.loss_img, self.dconf, self.dreg, self.dcenter, self.loss_ws)

    def _backward_iter(self):
        for op in self.bplan:
            kind = op[0]
            if kind == 'pred':
                _, c, targets, l, is_exp = op
                if is_exp:
                    ops.exp_rows_bwd(self.dreg[l], self.reg[l], self.grad_of(c), c.ld, c.M, c.C)
                else:
                    d = self.dconf[l] if targets is self.conf else self.dcenter[l]
                    ops.rows_from_f32(d, c.C, c.M, 0, self.grad_of(c), c.ld, c.M, c.C)
            elif kind == 'gnconv':
                _, name, x, y, out, acc, pname, acc_params, last_use = op
                dz = self.grad_of(out)
                ops.conv2d_wgrad(self.desc[name], y.t, dz, out.ld, self._flat(pname + '.w', self.G), self._flat(pname + '.b', self.G))
                dy = self.scr_y[: y.M * y.ld].view(y.M, y.ld)
                ops.conv2d_dgrad(self.desc[name], dz, out.ld, self.wt[pname], None, dy, False)
                ops.gn_bwd(x.t, x.ld, y.t, dy, y.ld, self.grad_of(x), x.ld, x.N, x.H * x.W, x.C, GROUPS, self.param(pname + '.gamma'),
                           self.gnsave[name], 1, int(acc) | (2 if acc_params else 0), self._flat(pname + '.gamma', self.G),
                           self._flat(pname + '.beta', self.G), self.gn_ws)
                if last_use:
                    yield pname
            elif kind == 'resize_add':
                _, lat, top, y, acc = op
                ops.resize_bilinear_bwd(self.grad_of(y), y.ld, self.grad_of(top), top.ld, top.N, top.H, top.W, y.H, y.W, top.C, acc)
            elif kind == 'pool':
                _, x, y, k, s, pt, pl = op
                ops.maxpool_bwd(x.t, y.t, self.grad_of(y), self.grad_of(x), x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pl)
            else:                                               # stem: conv -> GN -> ReLU (its bias DOES have a gradient: group norm is not
                _, name, src, z, y = op                        # invariant to a per-channel shift inside a group)
                dzs = self.scr_y[: z.M * z.ld].view(z.M, z.ld)
                ops.gn_bwd(z.t, z.ld, y.t, self.grad_of(y), y.ld, dzs, z.ld, z.N, z.H * z.W, z.C, GROUPS, self.param(name + '.gamma'),
                           self.gnsave[name], 1, False, self._flat(name + '.gamma', self.G), self._flat(name + '.beta', self.G), self.gn_ws)
                ops.conv2d_wgrad(self.desc[name], src.t, dzs, z.ld, self._flat(name + '.w', self.G), self._flat(name + '.b', self.G))
                yield name

    # ------------------------------------------------------------------ public: training
    def _set_batch_engine(self, images, ground_truth):
        images = torch.as_tensor(images, dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert tuple(images.shape) == tuple(self.images.shape), images.shape
        self.images.copy_(images, non_blocking=True)
        gt = torch.as_tensor(ground_truth, dtype=torch.float32)
        if self.gt is None or self.gt.shape != gt.shape:
            self.gt = torch.zeros(gt.shape, device=self.dev)
        self.gt.copy_(gt, non_blocking=True)

    def _train_step_engine(self, lr):
        """one optimizer step on the batch of set_batch(); returns the loss (data + L2) as a 1-element device tensor"""
        if self.dist is not None:
            self.dist.begin_step()
        self.G.zero_()
        self._forward()
        self._loss(1.0 / self.loss_divisor_batch)
        for name in self._backward_iter():
            if self.dist is not None:
                self.dist.layer_ready(name)
        if self.dist is not None:
            self.dist.finish_step()
        ops.sgd_momentum(self.P, self.Mom, self.G, lr, 0.9, self.weight_decay, 1.0, self.l2_partial, self.Pc if self.DT == BF16 else None)
        ops.sum_f32(self.l2_partial, self.l2_sum)
        self._fp_batch.run()
        self.global_step += 1
        return self.loss_img.mean() + self.weight_decay * self.l2_sum                              # FCOS.py:186-187

    def train_one_epoch(self, lr):
        if callable(self.train_initializer):
            self.train_initializer()
        mean_loss = []
        num_iters = self.num_train // self.batch_size
        it = iter(self.train_iterator)
        for i in range(num_iters):
            try:
                images, gt = next(it)
            except StopIteration:
                it = iter(self.train_iterator)
                images, gt = next(it)
            self.set_batch(images, gt)
            loss = float(self.train_step(lr).item())
            if self.verbose:
                sys.stdout.write('\r>> ' + 'iters ' + str(i) + str('/') + str(num_iters) + ' loss ' + str(loss))
                sys.stdout.flush()
            mean_loss.append(loss)
        if self.verbose:
            sys.stdout.write('\n')
        return np.mean(mean_loss)

    # ------------------------------------------------------------------ public: inference
    def test_one_image(self, images):
        images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert self.batch_size == 1 and tuple(images.shape) == tuple(self.images.shape), images.shape
        self.images.copy_(images)
        self._forward(subtract_mean=bool(self.config.get('test_subtract_mean', False)))
        scores, bbox, cid = heads.fcos_detect([t[0] for t in self.conf], [t[0] for t in self.reg], [t[0] for t in self.center],
                                              self.nms_score_threshold, self.nms_max_boxes, self.nms_iou_threshold)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    # ------------------------------------------------------------------ checkpoints / data parallel
    def _logical(self, name, buf):
        v = self.get_param(name, buf)
        return np.ascontiguousarray((v.permute(1, 2, 3, 0) if name.endswith('.w') else v).numpy())

    def export_tf_variables(self):
        """what the reference's `tf.train.Saver()` (FCOS.py:390-394) writes: every variable under its name (reference_variable_map),
        global_step, and the momentum slots `<variable>/Momentum` created under the 'head' scope of the graph (:111, :188)"""
        self._sync_from_twin()
        out = OrderedDict()
        for tfname, ours in reference_variable_map().items():
            out[tfname] = self._logical(ours, self.P)
            out[f'head/{tfname}/Momentum'] = self._logical(ours, self.Mom)
        out['global_step'] = np.asarray(self.global_step, dtype=np.int32)
        return out

    def load_tf_checkpoint(self, path, backbone_only=False):
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = reader.get_variable_to_shape_map()
        for tfname, ours in reference_variable_map().items():
            if backbone_only and not tfname.startswith('backone'):
                continue
            v = torch.from_numpy(reader.get_tensor(tfname))
            self.set_param(ours, v.permute(3, 0, 1, 2).contiguous() if ours.endswith('.w') else v)
            slot = [k for k in names if k.endswith(tfname + '/Momentum')]
            if slot and not backbone_only:
                mv = torch.from_numpy(reader.get_tensor(slot[0]))
                dst = self.param(ours, self.Mom)
                if ours.endswith('.w'):
                    dst.zero_()
                    dst[..., : mv.shape[2]] = mv.permute(3, 0, 1, 2).to(self.dev)
                else:
                    dst.copy_(mv.to(self.dev).view(dst.shape))
        if not backbone_only and reader.has_tensor('global_step'):
            self.global_step = int(reader.get_tensor('global_step'))
        self._refresh_operand_copies()

    def _save_weight_engine(self, mode, path):
        """FCOS.py:418-428.  config['checkpoint_format'] = 'tf' writes tf.train.Saver files (tf_checkpoint.py)."""
        assert (mode in ['latest', 'best'])
        dirname = os.path.dirname(path)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
            print(dirname, 'does not exist, create it done')
        if self.config.get('checkpoint_format', 'torch') == 'tf':
            from . import tf_checkpoint
            prefix = path + '-' + str(self.global_step)
            tf_checkpoint.write_bundle(prefix, self.export_tf_variables())
            tf_checkpoint.update_checkpoint_state(prefix)
            print('save', mode, 'model in', path, 'successfully')
            return
        blob = {'params': self.export_params(), 'momentum': self.Mom.detach().cpu(), 'global_step': self.global_step, 'layout': {k: (int(o), tuple(int(x) for x in shp)) for k, (o, shp) in self.pinfo.items()}}
        torch.save(blob, path + '-' + str(self.global_step))
        print('save', mode, 'model in', path, 'successfully')

    def load_weight(self, path):
        if os.path.exists(str(path) + '.index'):                 # a tf.train.Saver checkpoint prefix
            self.load_tf_checkpoint(path)
            print('load weight', path, 'successfully')
            return
        blob = torch.load(path, map_location='cpu', weights_only=True)
        unknown = sorted(k for k in blob['params'] if k not in self.pinfo and k not in getattr(self, 'sinfo', {}))
        if unknown:
            raise ValueError(f'{path}: {len(unknown)} parameters of the checkpoint are not part of this model (e.g. {unknown[:3]}): '
                             'it was written by a different layer layout')
        self.load_oracle_params(blob['params'])
        if tuple(blob['momentum'].shape) == tuple(self.Mom.shape) and dict(blob['layout']) == dict(self.pinfo):
            self.Mom.copy_(blob['momentum'].to(self.dev))
        else:
            import warnings
            warnings.warn(f'{path}: the parameter layout of the checkpoint differs from this model ({len(blob["layout"])} vs {len(self.pinfo)} entries): '
                          'momentum NOT restored (it stays as it is) although global_step is', RuntimeWarning)
        self.global_step = int(blob.get('global_step', 0))
        print('load weight', path, 'successfully')

    def load_pretrained_weight(self, path):
        """FCOS.py:434-436 restores the 'backone' variables: here the stem + unit layers of a saved file"""
        if os.path.exists(str(path) + '.index'):
            self.load_tf_checkpoint(path, backbone_only=True)
            print('load pretrained weight', path, 'successfully')
            return
        blob = torch.load(path, map_location='cpu', weights_only=True)['params']
        nb = 1 + 4 * sum(BLOCKS)
        self.load_oracle_params({k: v for k, v in blob.items() if int(k[1:].split('.')[0]) < nb})
        print('load pretrained weight', path, 'successfully')

    def attach_data_parallel(self, group=None, bucket_mb=25, grad_dtype='f32', force_collectives=False, collective='torch'):
        from .dist import GradAllReducer
        self.dist = GradAllReducer(self, group, bucket_mb, grad_dtype, force_collectives, collective)
        self.loss_divisor_batch = self.batch_size * self.dist.world
        return self.dist


def reference_variable_map():
    """name of every variable of the reference's FCOS graph -> our parameter name.  Default layer names (tf.layers: conv2d, conv2d_1, ...;
    contrib's group_norm: GroupNorm, GroupNorm_1, ...) are numbered PER ENCLOSING variable scope.  Scopes: 'backone' (sic, FCOS.py:71) with
    'block<b>_unit<u>/conv_branch|identity_branch' (:504-513), 'pyramid' (:98), 'head/classifier_head' and 'head/regress_head' (:351, :358) --
    entered once per level with reuse=tf.AUTO_REUSE, which SHARES the 6 + 5 head layers over the five levels (layer_specs).
    Pinned by tests/golden/fcos_variables.json (from the reference's own class on the shim)."""
    scopes = ['backone']
    for b, blocks in enumerate(BLOCKS):
        for u in range(blocks):
            base = f'backone/block{b + 1}_unit{u + 1}'
            scopes += [base + '/conv_branch'] * 3 + [base + '/identity_branch']
    scopes += ['pyramid'] * 10 + ['head/classifier_head'] * 6 + ['head/regress_head'] * 5
    m, count = OrderedDict(), {}
    for i, scope in enumerate(scopes):
        k = count.get(scope, 0)
        count[scope] = k + 1
        sfx = f'_{k}' if k else ''
        m[f'{scope}/conv2d{sfx}/kernel'], m[f'{scope}/conv2d{sfx}/bias'] = f'l{i}.w', f'l{i}.b'
        m[f'{scope}/GroupNorm{sfx}/beta'], m[f'{scope}/GroupNorm{sfx}/gamma'] = f'l{i}.beta', f'l{i}.gamma'
    return m
