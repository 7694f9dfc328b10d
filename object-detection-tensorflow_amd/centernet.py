"""CenterNet (DLA backbone + transposed-conv up-sampling tree + centre detector) behind the reference's class surface, on libodtk.

Reference: /root/reference/CenterNet.py
  * constructor, config keys ............. :11-47   (input_size, score_threshold, top_k_results_output; testcenternet.py:20-32)
  * input ................................ :49-70   ((images / 255 - mean) / std; test mode feeds the tensor AFTER that transform, so fed
                                                     pixels bypass it -- reproduced, config 'test_normalize' opts out)
  * network .............................. :72-134, :325-431: every layer conv(bias) -> batch norm -> ReLU; _basic_block's shortcut is
                                           tf.cond(channels == filters, identity, 1x1 conv): TensorFlow builds both branches, so the
                                           1x1 conv + batch norm exist as variables where they are never used ("ghost" layers: in the L2
                                           term, moved by weight decay only); DLA aggregation (one activation feeds several sums);
                                           4x4 / stride-2 transposed convolutions; 2x2 max and average pooling
  * loss, optimizer ...................... :136-157 (odtk_centernet_loss; mean over images + wd * l2(all trainables); AdamOptimizer)
  * inference ............................ :158-185 (heads.centernet_detect: 3x3 peak test + top-k, no NMS)
  * train / test / checkpoints ........... :298-323
Layers c0 .. c65 in creation order (layer k = conv k + batch norm k), one flat f32 parameter buffer + Adam's two moment buffers.
A transposed convolution runs as the DGRAD of the stride-2 conv it is the gradient of (its kernel stored as that conv's filter
[cin][4][4][cout]); its own backward is that conv's forward (input gradient) and wgrad with the operands swapped (filter gradient).
Every activation owns its gradient buffer; a consumer either writes it (first) or accumulates (later consumers) -- decided once, when
the plan is built -- so an activation may feed any number of sums and layers (the DLA tree needs that; yolov3.py's shared-buffer
residuals cover one sum per activation only).
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

MEAN = (0.485, 0.456, 0.406)                                  # CenterNet.py:52-53
STD = (0.229, 0.224, 0.225)
STRIDE = 4.0                                                   # :126
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8                  # tf.train.AdamOptimizer defaults (:154)
ADAM_SLOT_SCOPE = 'center_detector/'                            # optimizer.minimize is called inside this variable scope (CenterNet.py:131-156)


def layer_specs(num_classes):
    """[(name, kind, cin, cout, k, stride, relu, ghost)] in creation order (CenterNet.py:72-134, :378-402); kind 'conv' | 'dconv'"""
    specs = []

    def add(kind, cin, cout, k, s, relu=True, ghost=False):
        specs.append((f'c{len(specs)}', kind, cin, cout, k, s, relu, ghost))
        return cout

    def block(cin, f):
        add('conv', cin, f, 3, 1); add('conv', f, f, 3, 1)
        add('conv', cin, f, 1, 1, ghost=(cin == f))

    def dla(cin, f, levels):
        if levels == 1:
            block(cin, f); block(f, f)
        else:
            dla(cin, f, levels - 1); dla(f, f, levels - 1)
        add('conv', f, f, 3, 1)
    add('conv', 3, 16, 7, 1); add('conv', 16, 16, 3, 1); add('conv', 16, 32, 3, 2)
    dla(32, 64, 1)
    dla(64, 128, 2); add('conv', 64, 128, 1, 1)
    dla(128, 256, 2); add('conv', 128, 256, 1, 1)
    dla(256, 512, 1); add('conv', 256, 512, 1, 1)
    add('conv', 512, 256, 1, 1)
    for _ in range(3):
        add('dconv', 256, 256, 4, 2)
    add('conv', 256, 256, 1, 1); add('conv', 256, 256, 3, 1)
    add('dconv', 256, 256, 4, 2); add('dconv', 256, 256, 4, 2)
    add('conv', 128, 256, 1, 1); add('conv', 256, 256, 3, 1)
    add('dconv', 256, 256, 4, 2)
    add('conv', 256, 256, 3, 1); add('conv', 256, 256, 1, 1)
    add('conv', 256, num_classes, 3, 1, relu=False); add('conv', 256, 2, 3, 1, relu=False); add('conv', 256, 2, 3, 1, relu=False)
    return specs


class _Act:
    def __init__(self, name, N, H, W, C, ld, dtype, dev):
        self.name, self.N, self.H, self.W, self.C, self.ld = name, N, H, W, C, ld
        self.M = N * H * W
        self.t = torch.zeros(self.M, ld, dtype=dtype, device=dev)
        self.g = None                                          # gradient buffer (train mode, allocated by _build_backward)


class CenterNet(F32Warmup):
    OPT_BUFFERS = ('M1', 'M2')
    def __init__(self, config, data_provider):
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.input_size = config['input_size']
        self.data_shape = [self.input_size, self.input_size, 3] if config['data_format'] == 'channels_last' else [3, self.input_size, self.input_size]
        self.num_classes = config['num_classes']
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']                   # unused, as in the reference
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        assert self.input_size % 32 == 0, "CenterNet needs an input that is a multiple of 32 (five halvings; SAME pooling of odd maps is not implemented)"
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
        else:
            self.score_threshold = config['score_threshold']
            self.top_k_results_output = config['top_k_results_output']
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
    def _wshape(self, spec):
        _, kind, cin, cout, k, _, _, _ = spec
        kout, kin = (cout, cin) if kind == 'conv' else (cin, cout)             # dconv: the filter of the conv it is the gradient of
        return (kout, k, k, ops.pad_to(kin, self.chunk)), kin

    def _init_parameters(self, seed):
        pinfo, sinfo = OrderedDict(), OrderedDict()
        off = soff = 0
        self._kin = {}
        for spec in self.specs:
            name, cout = spec[0], spec[3]
            wshape, kin = self._wshape(spec)
            self._kin[name] = kin
            for suffix, shape in (('.w', wshape), ('.b', (cout,)), ('.gamma', (cout,)), ('.beta', (cout,))):
                pinfo[name + suffix] = (off, shape)
                off += ops.pad_to(int(np.prod(shape)), 64)
            for suffix in ('.mmean', '.mvar'):
                sinfo[name + suffix] = (soff, (cout,))
                soff += ops.pad_to(cout, 64)
        self.pinfo, self.sinfo, self.nparam = pinfo, sinfo, off
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.M1 = torch.zeros(off, device=dev)                 # Adam's m
        self.M2 = torch.zeros(off, device=dev)                 # Adam's v
        self.Mom = self.M1                                     # (name the data-parallel / checkpoint helpers of the other classes use)
        self.G = torch.zeros(off, device=dev)
        self.Pc = torch.zeros(off, dtype=self.tdt, device=dev) if self.DT == BF16 else self.P
        self.S = torch.zeros(soff, device=dev)
        self.l2_partial = torch.zeros(ops.sgd_blocks(off), device=dev)
        self.l2_sum = torch.zeros(1, device=dev)
        g = torch.Generator().manual_seed(seed)
        for name, kind, cin, cout, k, _, _, _ in self.specs:
            kout, kin = (cout, cin) if kind == 'conv' else (cin, cout)
            self.set_param(name + '.w', torch.randn(kout, k, k, kin, generator=g) * math.sqrt(2.0 / (cin * k * k)))
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
            v = v[..., : self._kin[name[:-2]]].contiguous()
        return v

    def load_oracle_params(self, p):
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        for k, v in p.items():
            if k in self.pinfo:
                if k.endswith('.b') and self._layer_kind[k[:-2]] == 'dconv' and float(torch.as_tensor(v).abs().max()) != 0.0:
                    raise NotImplementedError('non-zero bias of a transposed convolution (TensorFlow initialises it to 0 and, in front of a '
                                              'batch norm, neither the loss gradient nor the weight decay ever moves it)')
                self.set_param(k, v)
            elif k in self.sinfo:
                self.stat(k).copy_(torch.as_tensor(v, dtype=torch.float32).to(self.dev))
        self._refresh_operand_copies()

    def export_params(self):
        self._sync_from_twin()
        out = OrderedDict((k, self.get_param(k)) for k in self.pinfo)
        for k in self.sinfo:
            out[k] = self.stat(k).detach().cpu().clone()
        return out

    def _refresh_operand_copies(self):
        if self.DT == BF16:
            ops.cast_from_f32(self.P, self.Pc)
        if getattr(self, '_fp_batch', None) is not None:
            self._fp_batch.run()

    # ------------------------------------------------------------------ the graph: buffers + launch plan
    def _build(self):
        N, dev, dt, ch = self.batch_size, self.dev, self.tdt, self.chunk
        S_ = self.input_size
        self._layer_kind = {s[0]: s[1] for s in self.specs}
        self.images = torch.zeros(N, S_, S_, 3, device=dev)
        self.input = _Act('input', N, S_, S_, 3, ops.pad_to(3, ch), dt, dev)
        self.plan, self.desc, self.z, self.bnsave, self.acts = [], {}, {}, {}, {'input': self.input}
        it = iter(self.specs)
        self._max_ws = self._max_z = self._max_scr = 0

        def act(name, H_, W_, C_, f32=False):
            a = _Act(name, N, H_, W_, C_, C_ if f32 else ops.pad_to(C_, ch), torch.float32 if f32 else dt, dev)
            self.acts[name] = a
            return a

        def layer(src, out_f32=False):
            name, kind, cin, cout, k, stride, relu, ghost = next(it)
            assert not ghost and cin == src.C, (name, cin, src.C)
            ldz = ops.pad_to(cout, ch)
            if kind == 'conv':
                d = ops.conv_desc(N, src.H, src.W, ops.pad_to(cin, ch), src.ld, cout, ldz, k, stride, 1, self.CDT, self.CDT)
                Ho, Wo = d.Ho, d.Wo
            else:                                               # the stride-2 conv this layer is the gradient of: [2H,2W,cout] -> [H,W,cin]
                Ho, Wo = src.H * stride, src.W * stride
                d = ops.conv_desc(N, Ho, Wo, ldz, ldz, cin, src.ld, k, stride, 1, self.CDT, self.CDT)
                assert d.Ho == src.H and d.Wo == src.W
            self.desc[name] = d
            z = _Act(name + '.z', N, Ho, Wo, cout, ldz, dt, dev)
            y = act(name, Ho, Wo, cout, out_f32)
            self.z[name] = z
            self.bnsave[name] = (torch.zeros(cout, device=dev), torch.zeros(cout, device=dev))
            self._max_ws = max(self._max_ws, ops.bn_workspace_bytes(z.M, cout))
            self._max_z = max(self._max_z, z.M * ldz)
            self._max_scr = max(self._max_scr, src.M * src.ld)
            self.plan.append(('layer', name, kind, src, z, y, 1 if relu else 0))
            return y

        def ghost():
            spec = next(it)
            assert spec[7], spec

        def add(*ins):
            a = ins[0]
            y = act(f'sum{len(self.plan)}', a.H, a.W, a.C)
            self.plan.append(('add', list(ins), y))
            return y

        def pool(x, kind):
            y = act(f'{kind}{len(self.plan)}', x.H // 2, x.W // 2, x.C)
            self._max_scr = max(self._max_scr, x.M * x.ld)
            self.plan.append((kind, x, y))
            return y

        def block(x, f):
            c = layer(layer(x))
            if x.C == f:
                ghost()
                return add(c, x)
            return add(c, layer(x))

        def dla(x, f, levels):
            if levels == 1:
                b1 = block(x, f); b2 = block(b1, f)
            else:
                b1 = dla(x, f, levels - 1); b2 = dla(b1, f, levels - 1)
            return layer(add(b1, b2))

        x = layer(layer(layer(self.input)))
        stages = [pool(dla(x, 64, 1), 'maxpool')]
        for f, levels in ((128, 2), (256, 2), (512, 1)):
            prev = stages[-1]
            d_ = dla(prev, f, levels)
            res = pool(layer(prev), 'avgpool')
            stages.append(add(pool(d_, 'maxpool'), res))
        s3, s4, s5, s6 = stages
        u6 = layer(s6)
        u6_5 = layer(u6); u6_4 = layer(u6_5); u6_3 = layer(u6_4)
        u5 = layer(s5)
        u5_4 = layer(layer(add(u5, u6_5)))
        u5_3 = layer(u5_4)
        u4 = layer(s4)
        u4_3 = layer(layer(add(u4, u5_4, u6_4)))
        feat = layer(layer(add(u6_3, u5_3, u4_3)))
        self.kp_act, self.off_act, self.size_act = layer(feat, True), layer(feat, True), layer(feat, True)
        assert next(it, None) is None
        self.keypoints = self.kp_act.t.view(N, self.kp_act.H, self.kp_act.W, self.num_classes)
        self.offset = self.off_act.t.view(N, self.off_act.H, self.off_act.W, 2)
        self.size = self.size_act.t.view(N, self.size_act.H, self.size_act.W, 2)
        self.ws = torch.zeros(self._max_ws, dtype=torch.uint8, device=dev)
        # dgrad-layout filters: the backward of every conv but the first, and the FORWARD of every transposed conv
        self.wt, entries = {}, []
        for spec in self.specs:
            name, kind, cin, cout, k, _, _, ghost_ = spec
            if ghost_ or name == 'c0':
                continue
            (kout, _, _, kin_pad), _ = self._wshape(spec)
            kp = ops.pad_to(kout, ch)
            self.wt[name] = torch.zeros(kin_pad * k * k * kp, dtype=dt, device=dev)
            entries.append((self._flat(name + '.w', self.P), self.wt[name], kout, k, k, kin_pad, kp))
        self._fp_batch = ops.FilterPrepareBatch(entries, self.DT, dev)
        self.loss_ws = ops.centernet_workspace(N, self.kp_act.H, self.kp_act.W, self.num_classes, dev)
        if self.mode == 'train':
            self._build_backward(N, dt, dev)
        self._refresh_operand_copies()

    def _build_backward(self, N, dt, dev):
        self.zg = torch.zeros(self._max_z, dtype=dt, device=dev)             # d(pre-BN conv output): lives inside one layer
        self.scr = torch.zeros(self._max_scr, dtype=dt, device=dev)          # an input gradient on its way to being accumulated
        self.d_kp, self.d_off, self.d_size = torch.zeros_like(self.keypoints), torch.zeros_like(self.offset), torch.zeros_like(self.size)
        self.kp_act.g, self.off_act.g, self.size_act.g = (t.view(a.M, a.ld) for t, a in ((self.d_kp, self.kp_act), (self.d_off, self.off_act),
                                                                                        (self.d_size, self.size_act)))
        written = {id(self.kp_act), id(self.off_act), id(self.size_act)}

        def emit(a):
            """this consumer's contribution to d(a): first writer -> write, later ones -> accumulate"""
            acc = id(a) in written
            written.add(id(a))
            if a.g is None and a is not self.input:
                a.g = torch.zeros(a.M, a.ld, dtype=dt, device=dev)
            return acc
        # Round 6: an input of a sum whose ONLY consumer is that sum has d(input) = d(sum) exactly: it SHARES the sum's gradient buffer instead of receiving a
        # copy of it (DLA's aggregation nodes and residual sums: 85 add2d launches per step were 6.5 % of the step, a good part of them such copies).  Nobody
        # writes into a shared buffer again -- every reader (the producer's batch-norm / pool backward, another sum) only reads d(output).
        consumers = {}
        for op in self.plan:
            for a in ([op[3]] if op[0] == 'layer' else op[1] if op[0] == 'add' else [op[1]]):
                consumers[id(a)] = consumers.get(id(a), 0) + 1
        self.bplan = []
        self.shared_grads = 0
        for op in reversed(self.plan):
            kind = op[0]
            if kind == 'layer':
                _, name, lk, src, z, y, relu = op
                assert id(y) in written, name
                self.bplan.append(('layer', name, lk, src, z, y, relu, emit(src) if src is not self.input else False))
            elif kind == 'add':
                _, ins, y = op
                assert id(y) in written
                todo = []
                for a in ins:
                    if consumers[id(a)] == 1 and a.g is None and a is not self.input and (a.M, a.ld) == (y.M, y.ld) and self.config.get('share_sum_gradients', True):
                        a.g = y.g                                  # shared: no launch
                        written.add(id(a))
                        self.shared_grads += 1
                    else:
                        todo.append((a, emit(a)))
                self.bplan.append(('add', todo, y))
            else:
                _, x, y = op
                assert id(y) in written
                self.bplan.append((kind, x, y, emit(x)))
        self.loss_parts = torch.zeros(N, 4, device=dev)
        self.gt = None

    # ------------------------------------------------------------------ forward / loss / backward
    def _forward(self, training, normalize=True):
        if normalize:
            ops.preprocess_norm(self.images, 255., MEAN, STD, self.input.ld, self.DT, self.input.t)
        else:
            ops.preprocess(self.images, (0., 0., 0.), self.input.ld, self.DT, self.input.t)
        for op in self.plan:
            kind = op[0]
            if kind == 'layer':
                _, name, lk, src, z, y, relu = op
                if lk == 'conv':
                    ops.conv2d_fwd(self.desc[name], src.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), z.t, False)
                else:                                           # transposed conv = dgrad of its stride-2 conv (bias: exactly 0, see load_oracle_params)
                    ops.conv2d_dgrad(self.desc[name], src.t, src.ld, self.wt[name], None, z.t, False)
                sm, si = self.bnsave[name]
                ops.bn_fwd(z.t, z.M, z.C, z.ld, self.param(name + '.gamma'), self.param(name + '.beta'), self.stat(name + '.mmean'),
                           self.stat(name + '.mvar'), sm, si, training, relu, y.t, y.ld, z.M, 0, self.ws)
            elif kind == 'add':
                _, ins, y = op
                ops.add2d(ins[0].t, ins[0].ld, ins[1].t, ins[1].ld, y.t, y.ld, y.M, y.ld)
                for a in ins[2:]:
                    ops.add2d(y.t, y.ld, a.t, a.ld, y.t, y.ld, y.M, y.ld)
            elif kind == 'maxpool':
                _, x, y = op
                ops.maxpool_fwd(x.t, y.t, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, 2, 2, 0, 0)
            else:
                _, x, y = op
                ops.avgpool2x2_fwd(x.t, y.t, x.N, x.H, x.W, x.ld)

    def _loss(self, grad_scale):
        ops.centernet_loss(self.keypoints, self.offset, self.size, self.gt, STRIDE, grad_scale, self.loss_parts, self.d_kp, self.d_off, self.d_size,
                           self.loss_ws)

    def _into(self, a, acc):
        """destination of a kernel that can only overwrite: the gradient buffer itself (first writer) or the scratch (then added)"""
        return self.scr[: a.M * a.ld].view(a.M, a.ld) if acc else a.g

    def _fold(self, a, acc):
        if acc:
            ops.add2d(a.g, a.ld, self.scr[: a.M * a.ld].view(a.M, a.ld), a.ld, a.g, a.ld, a.M, a.ld)

    def _backward_iter(self):
        for op in self.bplan:
            kind = op[0]
            if kind == 'layer':
                _, name, lk, src, z, y, relu, acc = op
                zg = self.zg[: z.M * z.ld].view(z.M, z.ld)
                sm, si = self.bnsave[name]
                ops.bn_bwd(z.t, y.t if relu else None, y.g, z.M, z.C, z.ld, y.ld, z.M, 0, self.param(name + '.gamma'), sm, si, relu, zg,
                           self._flat(name + '.gamma', self.G), self._flat(name + '.beta', self.G), self.ws)
                # the bias feeds a batch norm: its gradient is exactly zero (only weight decay acts on it)
                if lk == 'conv':
                    ops.conv2d_wgrad(self.desc[name], src.t, zg, z.ld, self._flat(name + '.w', self.G), None)
                    if src is not self.input:
                        ops.conv2d_dgrad(self.desc[name], zg, z.ld, self.wt[name], None, src.g, acc)
                else:
                    # filter gradient of the underlying conv with the operands swapped: its "input" is d(z), its "output gradient" the layer's input
                    ops.conv2d_wgrad(self.desc[name], zg, src.t, src.ld, self._flat(name + '.w', self.G), None)
                    ops.conv2d_fwd(self.desc[name], zg, self._flat(name + '.w', self.Pc), None, self._into(src, acc), False)
                    self._fold(src, acc)
                yield name
            elif kind == 'add':
                _, ins, y = op
                for a, acc in ins:
                    if acc:
                        ops.add2d(a.g, a.ld, y.g, y.ld, a.g, a.ld, a.M, a.ld)
                    else:
                        ops.add2d(y.g, y.ld, None, 0, a.g, a.ld, a.M, a.ld)
            elif kind == 'maxpool':
                _, x, y, acc = op
                ops.maxpool_bwd(x.t, y.t, y.g, self._into(x, acc), x.N, x.H, x.W, x.C, x.ld, y.H, y.W, 2, 2, 0, 0)
                self._fold(x, acc)
            else:
                _, x, y, acc = op
                ops.avgpool2x2_bwd(y.g, self._into(x, acc), x.N, x.H, x.W, x.ld)
                self._fold(x, acc)

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
        """one AdamOptimizer step on the batch of set_batch(); returns the loss (data + L2) as a 1-element device tensor"""
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
        self.global_step += 1
        t = self.global_step
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        ops.adam(self.P, self.M1, self.M2, self.G, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, self.weight_decay, 1.0, self.l2_partial,
                 self.Pc if self.DT == BF16 else None)
        ops.sum_f32(self.l2_partial, self.l2_sum)
        self._fp_batch.run()
        return self.loss_parts[:, 3].mean() + self.weight_decay * self.l2_sum          # CenterNet.py:152-153 (pre-update weights)

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
                sys.stdout.write('\r>> ' + 'iters ' + str(i + 1) + str('/') + str(num_iters) + ' loss ' + str(loss))
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
        # Reference quirk, reproduced by default: test mode re-binds self.images to the normalised tensor and feeds THAT one
        # (CenterNet.py:66-67, :312), so fed pixels bypass the (x / 255 - mean) / std transform.
        self._forward(False, normalize=bool(self.config.get('test_normalize', False)))
        scores, bbox, cid = heads.centernet_detect(self.keypoints[0], self.offset[0], self.size[0], self.score_threshold,
                                                   self.top_k_results_output, STRIDE, self.loss_ws)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    # ------------------------------------------------------------------ checkpoints / data parallel
    def _logical(self, name, buf):
        """parameter `name` out of a flat buffer in TensorFlow's layout: kernels HWIO (transposed convs [h, w, out, in]), un-padded"""
        v = self.get_param(name, buf)
        return np.ascontiguousarray((v.permute(1, 2, 3, 0) if name.endswith('.w') else v).numpy())

    def export_tf_variables(self):
        """what the reference's `tf.train.Saver()` (CenterNet.py:296-301) would write: weights, moving statistics, global_step and AdamOptimizer's state --
        slots `center_detector/<variable>/Adam` (m), `…/Adam_1` (v) and the accumulators `center_detector/beta1_power` / `beta2_power`
        (= beta^(t + 1) after t steps): `optimizer.minimize` runs INSIDE `with tf.variable_scope('center_detector')` (:131-156), and both the slot
        creator (`variable_scope(None, primary.op.name + '/Adam')`) and the non-slot accumulators take the enclosing scope as a prefix"""
        self._sync_from_twin()
        out = OrderedDict()
        for tfname, ours in reference_variable_map(self.num_classes).items():
            if ours in self.pinfo:
                out[tfname] = self._logical(ours, self.P)
                out[ADAM_SLOT_SCOPE + tfname + '/Adam'] = self._logical(ours, self.M1)
                out[ADAM_SLOT_SCOPE + tfname + '/Adam_1'] = self._logical(ours, self.M2)
            else:
                out[tfname] = self.stat(ours).detach().cpu().numpy().copy()
        out[ADAM_SLOT_SCOPE + 'beta1_power'] = np.asarray(ADAM_B1 ** (self.global_step + 1), dtype=np.float32)
        out[ADAM_SLOT_SCOPE + 'beta2_power'] = np.asarray(ADAM_B2 ** (self.global_step + 1), dtype=np.float32)
        out['global_step'] = np.asarray(self.global_step, dtype=np.int32)
        return out

    def load_tf_checkpoint(self, path, backbone_only=False):
        """`saver.restore` (CenterNet.py:321-327) from tf.train.Saver files; backbone_only = the `pretrained_saver` over the trainables of 'backone'"""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = list(reader.get_variable_to_shape_map())
        missing_slots = []
        for tfname, ours in reference_variable_map(self.num_classes).items():
            if backbone_only and not (tfname.startswith('backone/') and ours in self.pinfo):
                continue
            v = torch.from_numpy(reader.get_tensor(tfname))                     # KeyError = Saver's NotFoundError
            if ours in self.pinfo:
                self.set_param(ours, v.permute(3, 0, 1, 2).contiguous() if ours.endswith('.w') else v)
                for slot, buf in (('/Adam', self.M1), ('/Adam_1', self.M2)):
                    if backbone_only:
                        continue
                    # the slot's scope prefix is whatever scope the optimizer was created in ('center_detector/' in the reference): match by suffix
                    found = [n for n in names if n.endswith(tfname + slot) and n[: len(n) - len(tfname + slot)] in ('', ADAM_SLOT_SCOPE)]
                    if not found:
                        missing_slots.append(tfname + slot)
                    else:
                        mv = torch.from_numpy(reader.get_tensor(found[0]))
                        dst = self.param(ours, buf)
                        if ours.endswith('.w'):
                            mv = mv.permute(3, 0, 1, 2)
                            dst.zero_()
                            dst[..., : mv.shape[-1]] = mv.to(self.dev)
                        else:
                            dst.copy_(mv.to(self.dev).view(dst.shape))
            else:
                self.stat(ours).copy_(v.to(self.dev))
        if not backbone_only and reader.has_tensor('global_step'):
            self.global_step = int(reader.get_tensor('global_step'))
        if missing_slots:
            import warnings
            warnings.warn(f'{path}: {len(missing_slots)} Adam slot variables not in the checkpoint (e.g. {missing_slots[0]}): their moments stay as they are, '
                          f'while global_step = {self.global_step} drives the bias correction', RuntimeWarning)
        self._refresh_operand_copies()

    def _save_weight_engine(self, mode, path):
        """CenterNet.py:314-319: one torch file `<path>-<step>` (parameters, moving statistics, Adam's moments and step), or with
        config['checkpoint_format'] = 'tf' the reference's own tf.train.Saver files"""
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
        blob = {'params': self.export_params(), 'adam_m': self.M1.detach().cpu(), 'adam_v': self.M2.detach().cpu(), 'global_step': self.global_step,
                'layout': {k: (int(o), tuple(int(x) for x in shp)) for k, (o, shp) in self.pinfo.items()}}
        torch.save(blob, path + '-' + str(self.global_step))
        print('save', mode, 'model in', path, 'successfully')

    def load_weight(self, path):
        if os.path.exists(str(path) + '.index'):                 # a tf.train.Saver checkpoint prefix
            self.load_tf_checkpoint(path)
            print('load weight', path, 'successfully')
            return
        blob = torch.load(path, map_location='cpu', weights_only=True)
        self.load_oracle_params(blob['params'])
        if tuple(blob['adam_m'].shape) == tuple(self.M1.shape) and dict(blob['layout']) == dict(self.pinfo):
            self.M1.copy_(blob['adam_m'].to(self.dev)); self.M2.copy_(blob['adam_v'].to(self.dev))
        else:
            import warnings
            warnings.warn(f'{path}: parameter layout differs from this model ({len(blob["layout"])} vs {len(self.pinfo)} entries): Adam moments NOT restored',
                          RuntimeWarning)
        self.global_step = int(blob.get('global_step', 0))
        print('load weight', path, 'successfully')

    def load_pretrained_weight(self, path):
        """CenterNet.py:321-323: `pretrained_saver` restores the trainable variables under 'backone' (c0 .. c49)"""
        if os.path.exists(str(path) + '.index'):
            self.load_tf_checkpoint(path, backbone_only=True)
            print('load pretrained weight', path, 'successfully')
            return
        blob = torch.load(path, map_location='cpu', weights_only=True)['params']
        self.load_oracle_params({k: v for k, v in blob.items() if k in self.pinfo and int(k[1:].split('.')[0]) < 50})
        print('load pretrained weight', path, 'successfully')

    def attach_data_parallel(self, group=None, bucket_mb=25, grad_dtype='f32', force_collectives=False, collective='torch'):
        from .dist import GradAllReducer
        self.dist = GradAllReducer(self, group, bucket_mb, grad_dtype, force_collectives, collective)
        self.loss_divisor_batch = self.batch_size * self.dist.world
        return self.dist


def reference_variable_map(num_classes=20):
    """name of every variable of the reference's graph -> our parameter / statistic name: default layer names numbered per enclosing
    variable scope ('backone' sic :73, 'upsampling' :111, 'center_detector' :131); conv2d and conv2d_transpose count separately, the
    batch norms over both.  Pinned by tests/golden/centernet_variables.json (from the reference's own class on the shim)."""
    m, count = OrderedDict(), {}
    for i, (name, kind, *_rest) in enumerate(layer_specs(num_classes)):
        scope = 'backone' if i < 50 else ('upsampling' if i < 63 else 'center_detector')
        base = 'conv2d' if kind == 'conv' else 'conv2d_transpose'
        k = count.get((scope, base), 0); count[(scope, base)] = k + 1
        b = count.get((scope, 'bn'), 0); count[(scope, 'bn')] = b + 1
        cn = f'{scope}/{base}' + (f'_{k}' if k else '')
        bn = f'{scope}/batch_normalization' + (f'_{b}' if b else '')
        m[cn + '/kernel'], m[cn + '/bias'] = name + '.w', name + '.b'
        m[bn + '/gamma'], m[bn + '/beta'] = name + '.gamma', name + '.beta'
        m[bn + '/moving_mean'], m[bn + '/moving_variance'] = name + '.mmean', name + '.mvar'
    return m
