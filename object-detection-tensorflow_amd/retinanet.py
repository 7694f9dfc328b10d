"""RetinaNet (the reference's pre-activation ResNet + pyramid + ten subnets) behind the reference's class surface, on libodtk.

Reference: /root/reference/RetinaNet.py -- the DETECTION graph (`is_pretraining: False`, testretinanet.py:22-41)
  * constructor, config keys ............. :12-77    (filters_list = [7 * 2^i] is built from the stem's KERNEL SIZE, :27 -- reproduced)
  * input ................................ :101-118  (images - mean; test mode feeds the tensor after the subtraction)
  * backbone ............................. :258-285, :634-643: 7x7 / s2 conv + batch norm + ReLU, 3x3 / s2 max pool, bottleneck units
                                           [BN-ReLU-1x1 f, BN-ReLU-3x3 f (stride), BN-ReLU-1x1 4f] + [BN-ReLU-3x3 4f (stride)] shortcut
  * pyramid, subnets ..................... :138-155, :287-319 (bilinear top-down path, the SUM is handed down; subnets not shared)
  * loss, optimizer ...................... :172-217  (odtk_retina_match / odtk_retina_loss; Momentum 0.9; L2 over all variables)
  * inference ............................ :218-256  (heads.retina_detect)
  * train / test / checkpoints ........... :488-535
The classification pre-training graph (`is_pretraining: True`, :120-135) is not built: NotImplementedError.
Same conventions as yolov3.py: layers l0 .. l121 in creation order (layer k = conv k + batch norm k; l0 is conv -> BN -> ReLU, every
other layer BN -> ReLU -> conv with a LIVE bias gradient), one flat f32 parameter buffer, NHWC rows with zero-filled pad columns for
the 7 / 14 / 28-channel maps.  A batch norm whose input feeds several consumers accumulates its dx through a scratch + odtk_add2d.
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

MEAN_RGB = (123.68, 116.779, 103.979)
FILTERS = (7, 14, 28, 56)                                     # RetinaNet.py:27 (sic)
ANCHOR_SIZES = (32, 64, 128, 256, 512)                        # :39
ASPECT_RATIOS = (1, 1 / 2, 2)                                 # :40
ANCHOR_SCALES = (2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3))          # :41
PI = 0.01                                                     # :48


def layer_specs(block_list, init_filters, num_classes, num_anchors=9):
    """[(name, cin, cout, k, stride, bn_channels, bias_init)] in creation order (RetinaNet.py:258-301, :634-643)"""
    specs = []

    def add(cin, cout, k, s, bias_init=0.):
        specs.append((f'l{len(specs)}', cin, cout, k, s, cin if specs else cout, bias_init))
        return cout
    c = add(3, init_filters, 7, 2)
    stage_out = []
    for i, blocks in enumerate(block_list):
        f = FILTERS[i]
        for j in range(blocks):
            s = 2 if (i > 0 and j == 0) else 1
            add(c, f, 1, 1); add(f, f, 3, s); add(f, 4 * f, 1, 1)
            add(c, 4 * f, 3, s)
            c = 4 * f
        stage_out.append(c)
    f1, f2, f3 = stage_out[-3:]
    add(f3, 256, 3, 1)
    add(f2, 256, 1, 1); add(256, 256, 3, 1)
    add(f1, 256, 1, 1); add(256, 256, 3, 1)
    add(256, 256, 3, 2); add(256, 256, 3, 2)
    for _ in range(5):
        for _ in range(4):
            add(256, 256, 3, 1)
        add(256, num_classes * num_anchors, 3, 1, -math.log((1 - PI) / PI))
        for _ in range(4):
            add(256, 256, 3, 1)
        add(256, 4 * num_anchors, 3, 1)
    return specs


def level_priors(size):
    """(h, w) of the 9 anchors of one level, python doubles in the reference's loop order (RetinaNet.py:332-338)"""
    out = []
    for ratio in ASPECT_RATIOS:
        for scale in ANCHOR_SCALES:
            out.append((size * scale * (ratio ** 0.5), size * scale / (ratio ** 0.5)))
    return out


class _Act:
    def __init__(self, name, N, H, W, C, ld, dtype, dev):
        self.name, self.N, self.H, self.W, self.C, self.ld = name, N, H, W, C, ld
        self.M = N * H * W
        self.t = torch.zeros(self.M, ld, dtype=dtype, device=dev)
        self.gid = name


class RetinaNet:
    def __init__(self, config, data_provider):
        assert len(config['data_shape']) == 3
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        if config.get('is_pretraining'):
            raise NotImplementedError('the classification pre-training graph (RetinaNet.py:120-135) is not built; is_pretraining must be False')
        assert config['is_bottleneck'], 'only the bottleneck units of testretinanet.py are built'
        self.config = config
        self.data_provider = data_provider
        self.block_list = list(config['residual_block_list'])
        self.data_shape = config['data_shape']
        self.num_classes = config['num_classes'] + 1
        self.weight_decay = config['weight_decay']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.gamma, self.alpha = config['gamma'], config['alpha']
        self.num_anchors = len(ASPECT_RATIOS) * len(ANCHOR_SCALES)
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.verbose = bool(config.get('verbose', True))
        self.dev = torch.device(config.get('device', 'cuda:0'))
        # f32 by default: the reference's identity-free residual units amplify bf16's rounding of the stored activations to O(1) by the
        # end of the backbone at random initialisation; 'bf16' runs (3.4x faster) but is not validated for training
        # 'f32x3' (round 4): the f32 engine -- every tensor stays f32 -- whose convolution DESCRIPTORS say ODTK_F32X3: the library runs a layer's three passes as
        # three bf16 MFMA products per f32 product (operand splitting, 2^-17 per product instead of bf16's 2^-9) where that is faster than the exact f32 MFMA
        # kernel, i.e. on the 256-channel pyramid and heads, and exactly on the narrow backbone layers (include/odtk.h).  Default for training: it clears the gate of
        # tests/test_gpu_bf16_gate.py at RANDOM INITIALISATION (every filter gradient within cosine 0.997 of the f32 engine's, 1.000 after 300 steps) at 1.9x the
        # f32 engine's throughput; mode 'test' keeps the exact 'f32'.
        engine = config.get('compute_dtype') or ('f32x3' if config['mode'] == 'train' else 'f32')
        self.x3 = engine == 'f32x3'
        self.DT = {'bf16': BF16, 'f32': F32, 'f32x3': F32}[engine]
        self.CDT = F32X3 if self.x3 else self.DT             # what the convolution descriptors carry
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
        self.specs = layer_specs(self.block_list, config['init_conv_filters'], self.num_classes, self.num_anchors)
        self._init_parameters(int(config.get('seed', 0)))
        self._build()

    # ------------------------------------------------------------------ parameters
    def param_layout(self):
        pinfo, sinfo = OrderedDict(), OrderedDict()
        off = soff = 0
        for name, cin, cout, k, _, bnc, _ in self.specs:
            for suffix, shape in (('.w', (cout, k, k, ops.pad_to(cin, self.chunk))), ('.b', (cout,)), ('.gamma', (bnc,)), ('.beta', (bnc,))):
                pinfo[name + suffix] = (off, shape)
                off += ops.pad_to(int(np.prod(shape)), 64)
            for suffix in ('.mmean', '.mvar'):
                sinfo[name + suffix] = (soff, (bnc,))
                soff += ops.pad_to(bnc, 64)
        return pinfo, off, sinfo, soff

    def _init_parameters(self, seed):
        self.pinfo, off, self.sinfo, soff = self.param_layout()
        self.nparam = off
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.Mom = torch.zeros(off, device=dev)
        self.G = torch.zeros(off, device=dev)
        self.Pc = torch.zeros(off, dtype=self.tdt, device=dev) if self.DT == BF16 else self.P
        self.S = torch.zeros(soff, device=dev)
        self.l2_partial = torch.zeros(ops.sgd_blocks(off), device=dev)
        self.l2_sum = torch.zeros(1, device=dev)
        self._cin = {s[0]: s[1] for s in self.specs}
        g = torch.Generator().manual_seed(seed)
        for name, cin, cout, k, _, _, bias_init in self.specs:
            self.set_param(name + '.w', torch.randn(cout, k, k, cin, generator=g) * math.sqrt(2.0 / (cin * k * k)))
            self.param(name + '.b').fill_(float(bias_init))
            self.param(name + '.gamma').fill_(1.0)
            self.stat(name + '.mvar').fill_(1.0)

    def param(self, name, buf=None):
        off, shape = self.pinfo[name]
        buf = self.P if buf is None else buf
        return buf[off: off + int(np.prod(shape))].view(shape)

    def stat(self, name):
        off, shape = self.sinfo[name]
        return self.S[off: off + int(np.prod(shape))].view(shape)

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
        for k, v in p.items():
            if k in self.pinfo:
                self.set_param(k, v)
            elif k in self.sinfo:
                self.stat(k).copy_(torch.as_tensor(v, dtype=torch.float32).to(self.dev))
        self._refresh_operand_copies()

    def export_params(self):
        out = OrderedDict((k, self.get_param(k)) for k in self.pinfo)
        for k in self.sinfo:
            out[k] = self.stat(k).detach().cpu().clone()
        return out

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
        self.plan, self.desc, self.bnsave, self.acts = [], {}, {}, {}
        it = iter(self.specs)
        self._max_ws = self._max_scr = 0
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

        def note(a):
            self._max_ws = max(self._max_ws, ops.bn_workspace_bytes(a.M, a.C))
            self._max_scr = max(self._max_scr, a.M * a.ld)

        def conv_desc(name, src, cout, k, stride, ldy):
            d = ops.conv_desc(N, src.H, src.W, src.ld, src.ld, cout, ldy, k, stride, 1, self.CDT, self.CDT)
            self.desc[name] = d
            return d

        def bnconv(x):
            """batch norm -> ReLU -> conv(bias): returns the conv output"""
            name, cin, cout, k, stride, bnc, _ = next(it)
            assert cin == x.C == bnc, (name, cin, x.C)
            y = act(name + '.y', x.H, x.W, x.C)                  # relu(bn(x)): the conv's operand
            d = conv_desc(name, y, cout, k, stride, ops.pad_to(cout, ch))
            out = act(name, d.Ho, d.Wo, cout)
            self.bnsave[name] = (torch.zeros(bnc, device=dev), torch.zeros(bnc, device=dev))
            note(x); note(out)
            self.plan.append(('bnconv', name, x, y, out))
            return out

        def add(a, b):
            y = act(f'sum{len(self.plan)}', a.H, a.W, a.C)
            groups[find(a.gid)] = find(y.gid)                      # both operands are conv outputs with this single consumer
            groups[find(b.gid)] = find(y.gid)
            self.plan.append(('add', a, b, y))
            return y

        def resize_add(lat, top):
            y = act(f'total{len(self.plan)}', lat.H, lat.W, lat.C)
            groups[find(lat.gid)] = find(y.gid)
            self.plan.append(('resize_add', lat, top, y))
            return y

        # stem: conv -> batch norm -> ReLU -> 3x3 / stride-2 max pool (RetinaNet.py:260-271)
        name, cin, cout, k, stride, bnc, _ = next(it)
        d = conv_desc(name, self.input, cout, k, stride, ops.pad_to(cout, ch))
        z = act(name + '.z', d.Ho, d.Wo, cout)
        y = act(name, d.Ho, d.Wo, cout)
        self.bnsave[name] = (torch.zeros(bnc, device=dev), torch.zeros(bnc, device=dev))
        note(z)
        self.plan.append(('stem', name, self.input, z, y))
        Hp, pt, _ = ops.same_pad(y.H, 3, 2)
        Wp, pl, _ = ops.same_pad(y.W, 3, 2)
        x = act('pool1', Hp, Wp, cout)
        self.plan.append(('pool', y, x, 3, 2, pt, pl))
        feats = []
        for i, blocks in enumerate(self.block_list):
            for _ in range(blocks):
                branch = bnconv(bnconv(bnconv(x)))
                x = add(branch, bnconv(x))
            feats.append(x)
        f1, f2, f3 = feats[-3:]
        p5 = bnconv(f3)
        total4 = resize_add(bnconv(f2), p5)
        p4 = bnconv(total4)
        total3 = resize_add(bnconv(f1), total4)
        p3 = bnconv(total3)
        p6 = bnconv(p5)
        p7 = bnconv(p6)
        self.levels = [p3, p4, p5, p6, p7]
        self.shapes = [(a.H, a.W) for a in self.levels]
        A = sum(h * w * self.num_anchors for h, w in self.shapes)
        self.num_anchor_boxes = A
        self.pconf = torch.zeros(N, A, self.num_classes, device=dev)
        self.pbox = torch.zeros(N, A, 4, device=dev)
        off = 0
        for lvl in self.levels:
            for target, width in ((self.pconf, self.num_classes), (self.pbox, 4)):
                c = lvl
                for _ in range(5):
                    c = bnconv(c)
                self.plan.append(('pred', c, target, off, width))
            off += lvl.H * lvl.W * self.num_anchors
        assert next(it, None) is None and off == A
        self.ws = torch.zeros(self._max_ws, dtype=torch.uint8, device=dev)
        # dgrad-layout filters (every conv but the stem)
        self.wt, entries = {}, []
        for name, cin, cout, k, _, _, _ in self.specs[1:]:
            d = self.desc[name]
            kp = self.acts[name].ld
            self.wt[name] = torch.zeros(d.C * k * k * kp, dtype=dt, device=dev)
            entries.append((self._flat(name + '.w', self.P), self.wt[name], cout, k, k, d.C, kp))
        self._fp_batch = ops.FilterPrepareBatch(entries, self.DT, dev)
        # anchors (RetinaNet.py:328-355; x uses the H rate: data_shape[1] is read as the height, :330)
        flat = [v for s in ANCHOR_SIZES for hw in level_priors(s) for v in hw]
        self.anc = ops.retina_anchors(self.data_shape[1], self.shapes, [self.num_anchors] * 5, flat, dev)     # y1x1, y2x2, yx, hw
        if self.mode == 'train':
            self._build_backward(N, A, dt, dev)
        self._refresh_operand_copies()

    def _build_backward(self, N, A, dt, dev):
        find = self.find
        self.dconf, self.dbox = torch.zeros_like(self.pconf), torch.zeros_like(self.pbox)
        self.scr_y = torch.zeros(self._max_scr, dtype=dt, device=dev)          # d(relu(bn(x))): lives inside one layer
        self.scr_x = torch.zeros(self._max_scr, dtype=dt, device=dev)          # dx of a batch norm whose input already holds a gradient
        written = set()
        self.bplan = []
        for op in reversed(self.plan):
            kind = op[0]
            if kind == 'pred':
                _, c, target, off, width = op
                self.bplan.append(op)
                written.add(find(c.gid))
            elif kind == 'bnconv':
                _, name, x, y, out = op
                assert find(out.gid) in written, name
                self.bplan.append(('bnconv', name, x, y, out, find(x.gid) in written))
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
        P = 1
        i32 = dict(dtype=torch.int32, device=dev)
        self.m_ngt = torch.zeros(N, **i32)
        self.m_status = torch.zeros(N, A, dtype=torch.uint8, device=dev)
        self.m_rg = torch.zeros(N, A, **i32)
        self.m_counts = torch.zeros(N, 4, **i32)
        self.m_best = None
        self.m_ws = None
        self.loss_parts = torch.zeros(N, 2, device=dev)
        self.gt = None

    def grad_of(self, a):
        return self.g[self.find(a.gid)]

    # ------------------------------------------------------------------ forward / loss / backward
    def _bn_relu(self, name, x, y, training):
        sm, si = self.bnsave[name]
        ops.bn_fwd(x.t, x.M, x.C, x.ld, self.param(name + '.gamma'), self.param(name + '.beta'), self.stat(name + '.mmean'),
                   self.stat(name + '.mvar'), sm, si, training, 1, y.t, y.ld, x.M, 0, self.ws)

    def _forward(self, training, subtract_mean=True):
        ops.preprocess(self.images, MEAN_RGB if subtract_mean else (0., 0., 0.), self.input.ld, self.DT, self.input.t)
        for op in self.plan:
            kind = op[0]
            if kind == 'bnconv':
                _, name, x, y, out = op
                self._bn_relu(name, x, y, training)
                ops.conv2d_fwd(self.desc[name], y.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), out.t, False)
            elif kind == 'add':
                _, a, b, y = op
                ops.add2d(a.t, a.ld, b.t, b.ld, y.t, y.ld, y.M, y.ld)
            elif kind == 'resize_add':
                _, lat, top, y = op
                ops.add2d(lat.t, lat.ld, None, 0, y.t, y.ld, y.M, y.ld)
                ops.resize_bilinear_fwd(top.t, top.ld, y.t, y.ld, top.N, top.H, top.W, y.H, y.W, top.C, True)
            elif kind == 'pred':
                _, c, target, off, width = op
                K = self.num_anchors * width
                ops.rows_to_f32(c.t, c.ld, target[0, off:], K, c.H * c.W, target.shape[1] * width, c.M, K)
            elif kind == 'stem':
                _, name, src, z, y = op
                ops.conv2d_fwd(self.desc[name], src.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), z.t, False)
                self._bn_relu(name, z, y, training)
            else:
                _, x, y, k, s, pt, pl = op
                ops.maxpool_fwd(x.t, y.t, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pl)

    def _loss(self, grad_scale):
        N, P = self.gt.shape[0], self.gt.shape[1]
        if self.m_best is None or self.m_best.shape[1] != P:
            self.m_best = torch.zeros(N, P, dtype=torch.int32, device=self.dev)
            self.m_ws = ops.retina_match_workspace(self.num_anchor_boxes, N, P, self.dev)
        y1x1, y2x2, yx, hw = self.anc
        ops.retina_match(y1x1, y2x2, hw, self.gt, self.m_ngt, self.m_best, self.m_status, self.m_rg, self.m_counts, self.m_ws)
        ops.retina_loss(self.pconf, self.pbox, yx, hw, self.gt, self.m_ngt, self.m_best, self.m_status, self.m_rg, self.m_counts, self.alpha,
                        self.gamma, grad_scale, self.loss_parts, self.dconf, self.dbox)

    def _backward_iter(self):
        for op in self.bplan:
            kind = op[0]
            if kind == 'pred':
                _, c, target, off, width = op
                d = self.dconf if target is self.pconf else self.dbox
                K = self.num_anchors * width
                ops.rows_from_f32(d[0, off:], K, c.H * c.W, d.shape[1] * width, self.grad_of(c), c.ld, c.M, K)
            elif kind == 'bnconv':
                _, name, x, y, out, acc = op
                dz = self.grad_of(out)
                ops.conv2d_wgrad(self.desc[name], y.t, dz, out.ld, self._flat(name + '.w', self.G), self._flat(name + '.b', self.G))
                dy = self.scr_y[: y.M * y.ld].view(y.M, y.ld)
                ops.conv2d_dgrad(self.desc[name], dz, out.ld, self.wt[name], None, dy, False)
                sm, si = self.bnsave[name]
                dx = self.scr_x[: x.M * x.ld].view(x.M, x.ld) if acc else self.grad_of(x)
                ops.bn_bwd(x.t, y.t, dy, x.M, x.C, x.ld, y.ld, x.M, 0, self.param(name + '.gamma'), sm, si, 1, dx,
                           self._flat(name + '.gamma', self.G), self._flat(name + '.beta', self.G), self.ws)
                if acc:
                    gx = self.grad_of(x)
                    ops.add2d(gx, x.ld, dx, x.ld, gx, x.ld, x.M, x.ld)
                yield name
            elif kind == 'resize_add':
                _, lat, top, y, acc = op
                ops.resize_bilinear_bwd(self.grad_of(y), y.ld, self.grad_of(top), top.ld, top.N, top.H, top.W, y.H, y.W, top.C, acc)
            elif kind == 'pool':
                _, x, y, k, s, pt, pl = op
                ops.maxpool_bwd(x.t, y.t, self.grad_of(y), self.grad_of(x), x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pl)
            else:                                               # stem: the conv bias feeds batch norm -> zero gradient
                _, name, src, z, y = op
                sm, si = self.bnsave[name]
                dzs = self.scr_y[: z.M * z.ld].view(z.M, z.ld)
                ops.bn_bwd(z.t, y.t, self.grad_of(y), z.M, z.C, z.ld, y.ld, z.M, 0, self.param(name + '.gamma'), sm, si, 1, dzs,
                           self._flat(name + '.gamma', self.G), self._flat(name + '.beta', self.G), self.ws)
                ops.conv2d_wgrad(self.desc[name], src.t, dzs, z.ld, self._flat(name + '.w', self.G), None)
                yield name

    # ------------------------------------------------------------------ public: training
    def set_batch(self, images, ground_truth):
        images = torch.as_tensor(images, dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert tuple(images.shape) == tuple(self.images.shape), images.shape
        self.images.copy_(images, non_blocking=True)
        gt = torch.as_tensor(ground_truth, dtype=torch.float32)
        if self.gt is None or self.gt.shape != gt.shape:
            self.gt = torch.zeros(gt.shape, device=self.dev)
        self.gt.copy_(gt, non_blocking=True)

    def train_step(self, lr):
        """one optimizer step on the batch of set_batch(); returns the loss (data + L2) as a 1-element device tensor"""
        if self.dist is not None:
            self.dist.begin_step()
        self.G.zero_()
        self._forward(True)
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
        return self.loss_parts.sum() / self.batch_size + self.weight_decay * self.l2_sum          # RetinaNet.py:205-213

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
        self._forward(False, subtract_mean=bool(self.config.get('test_subtract_mean', False)))
        scores, bbox, cid = heads.retina_detect(self.pconf[0], self.pbox[0], self.anc[2], self.anc[3], self.nms_score_threshold,
                                                self.nms_max_boxes, self.nms_iou_threshold)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    # ------------------------------------------------------------------ checkpoints / data parallel
    def _logical(self, name, buf):
        v = self.get_param(name, buf)
        return np.ascontiguousarray((v.permute(1, 2, 3, 0) if name.endswith('.w') else v).numpy())

    def export_tf_variables(self):
        """what the reference's detection `tf.train.Saver()` (RetinaNet.py:553-557) writes: every variable of the graph under its name
        (reference_variable_map), global_step, and the momentum slots, created inside the 'inference' scope (:172, :206)"""
        out = OrderedDict()
        for tfname, ours in reference_variable_map(self.block_list).items():
            if ours in self.pinfo:
                out[tfname] = self._logical(ours, self.P)
                out[f'inference/{tfname}/Momentum'] = self._logical(ours, self.Mom)
            else:
                out[tfname] = self.stat(ours).detach().cpu().numpy().copy()
        out['global_step'] = np.asarray(self.global_step, dtype=np.int32)
        return out

    def load_tf_checkpoint(self, path, backbone_only=False):
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = reader.get_variable_to_shape_map()
        nb = 1 + 4 * sum(self.block_list)
        for tfname, ours in reference_variable_map(self.block_list).items():
            if backbone_only and int(ours[1:].split('.')[0]) >= nb:
                continue
            v = torch.from_numpy(reader.get_tensor(tfname))
            if ours in self.sinfo:
                self.stat(ours).copy_(v.to(self.dev))
                continue
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

    def save_weight(self, mode, path):
        """RetinaNet.py:521-531.  config['checkpoint_format'] = 'tf' writes tf.train.Saver files (tf_checkpoint.py)."""
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

    def load_pretraining_weight(self, path):
        """RetinaNet.py:537-539 restores the 'feature_extractor' variables saved by the pre-training graph: here the backbone layers
        (stem + units) of a saved file"""
        if os.path.exists(str(path) + '.index'):
            self.load_tf_checkpoint(path, backbone_only=True)
            print('load pretraining weight', path, 'successfully')
            return
        blob = torch.load(path, map_location='cpu', weights_only=True)['params']
        nb = 1 + 4 * sum(self.block_list)
        self.load_oracle_params({k: v for k, v in blob.items() if int(k[1:].split('.')[0]) < nb})
        print('load pretraining weight', path, 'successfully')

    def attach_data_parallel(self, group=None, bucket_mb=25, grad_dtype='f32', force_collectives=False, collective='torch'):
        from .dist import GradAllReducer
        self.dist = GradAllReducer(self, group, bucket_mb, grad_dtype, force_collectives, collective)
        self.loss_divisor_batch = self.batch_size * self.dist.world
        return self.dist


def reference_variable_map(block_list=(3, 4, 6, 3)):
    """name of every variable of the reference's detection graph -> our parameter / statistic name.  tf.layers default layer names
    (conv2d, conv2d_1, ...; batch_normalization, _1, ...) are numbered PER ENCLOSING variable scope; scopes: 'feature_extractor' for the stem and the
    pyramid, 'feature_extractor/block<b>_unit<u>/conv_branch|identity_branch' for the units (RetinaNet.py:621-643), 'regressor' for the
    subnets (:145).  Pinned by tests/golden/retinanet_variables.json (collected from the reference's own class)."""
    scopes = ['feature_extractor']
    for b, blocks in enumerate(block_list):
        for u in range(blocks):
            base = f'feature_extractor/block{b + 1}_unit{u + 1}'
            scopes += [base + '/conv_branch'] * 3 + [base + '/identity_branch']
    scopes += ['feature_extractor'] * 7 + ['regressor'] * 50
    m, count = OrderedDict(), {}
    for i, scope in enumerate(scopes):
        k = count.get(scope, 0)                      # default layer names are numbered per enclosing variable scope
        count[scope] = k + 1
        sfx = '' if k == 0 else f'_{k}'
        m[f'{scope}/conv2d{sfx}/kernel'], m[f'{scope}/conv2d{sfx}/bias'] = f'l{i}.w', f'l{i}.b'
        bn = f'{scope}/batch_normalization{sfx}'
        m[bn + '/gamma'], m[bn + '/beta'] = f'l{i}.gamma', f'l{i}.beta'
        m[bn + '/moving_mean'], m[bn + '/moving_variance'] = f'l{i}.mmean', f'l{i}.mvar'
    return m
