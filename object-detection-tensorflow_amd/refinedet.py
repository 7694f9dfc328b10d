"""RefineDet320 / 512 (VGG-16 + anchor refinement module + transfer connection blocks + object detection module) behind the reference's class
surface, on libodtk.

Reference: /root/reference/RefineDet.py (class RefineDet320; `input_size` 320 or 512, testrefinedet.py:20-33)
  * constructor, config keys ............. :11-49
  * input ................................ :51-70   (images - mean; test mode feeds the tensor after the subtraction -- reproduced, 'test_subtract_mean' opts out)
  * VGG trunk + extras ................... :232-385  (conv1_1 .. conv5_3: conv + bias + ReLU; conv6 .. conv10_2: conv(bias) + batch norm + ReLU)
  * L2-normalised conv4_3 / conv5_3 ...... :74-95    (one learnable scalar each, 10 and 8)
  * ARM / TCB / ODM ...................... :387-415  (4 x [3x3(256) + BN + ReLU] + two 3x3 + BN outputs per level; TCB: 3x3 + BN + ReLU, 3x3 + BN,
                                                       + [4x4 / s2 transposed conv + BN] of the level above, ReLU)
  * loss, optimizer ...................... :160-187  (heads.RefineDetLoss: matching, NMS-mined ARM negatives, two-stage loss; Momentum 0.9)
  * inference ............................ :189-230  (heads.refinedet_detect)
  * train / test / checkpoints ........... :583-617
Layers by name (oracle/refinedet_net_ref.layer_specs' names), one flat f32 parameter buffer in TensorFlow's creation order.  The graph engine is
centernet.py's (every activation owns its gradient buffer; a consumer writes it first or accumulates later), extended by the VGG convention of
ssd300.py: the gradient buffer of a bias + ReLU activation holds d(pre-activation) and every consumer applies the ReLU mask itself (dgrad's
`relu_src`, the L2-norm backward's; a max pool routes only to unmasked cells).
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
VGG_SEQ = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool1", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool2",
           ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool3",
           ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "pool4",
           ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), "pool5"]
EXTRAS = [("conv6", 512, 1024, 3, 1, 2), ("conv7", 1024, 1024, 1, 1, 1), ("conv8_1", 1024, 256, 1, 1, 1), ("conv8_2", 256, 512, 3, 2, 1),
          ("conv9_1", 512, 256, 1, 1, 1), ("conv9_2", 256, 512, 3, 2, 1), ("conv10_1", 512, 256, 1, 1, 1), ("conv10_2", 256, 256, 3, 1, 1)]
FEAT_CH = [512, 512, 512, 256]
NA = 3


def layer_specs(num_classes):
    """[(name, kind, cin, cout, k, stride, dil, relu)] in creation order; kind 'vgg' (bias + ReLU) | 'conv' | 'dconv' (both + batch norm)"""
    s = []
    for l in VGG_SEQ:
        if isinstance(l, tuple):
            s.append((l[0], 'vgg', l[1], l[2], 3, 1, 1, True))
    for (n, ci, co, k, st, d) in EXTRAS:
        s.append((n, 'conv', ci, co, k, st, d, True))

    def head(prefix, cin, ncls):
        c = cin
        for j in range(1, 5):
            s.append((f'{prefix}.c{j}', 'conv', c, 256, 3, 1, 1, True)); c = 256
        s.append((f'{prefix}.loc', 'conv', 256, 4 * NA, 3, 1, 1, False))
        s.append((f'{prefix}.conf', 'conv', 256, ncls * NA, 3, 1, 1, False))
    for l in range(4):
        head(f'arm{l + 1}', FEAT_CH[l], 2)
    for l in (4, 3, 2, 1):
        s.append((f'tcb{l}.c1', 'conv', FEAT_CH[l - 1], 256, 3, 1, 1, True))
        s.append((f'tcb{l}.c2', 'conv', 256, 256, 3, 1, 1, l == 4))
        if l < 4:
            s.append((f'tcb{l}.d', 'dconv', 256, 256, 4, 2, 1, False))
    for l in range(4):
        head(f'odm{l + 1}', 256, num_classes)
    return s


class _Act:
    def __init__(self, name, N, H, W, C, ld, dtype, dev, vgg=False, alloc=True):
        self.name, self.N, self.H, self.W, self.C, self.ld, self.vgg = name, N, H, W, C, ld, vgg
        self.M = N * H * W
        self.t = torch.zeros(self.M, ld, dtype=dtype, device=dev) if alloc else None
        self.g = None


class RefineDet320(F32Warmup):
    # RefineDet320 / PFPNetR do not pass the bf16 gate (one head layer loses its direction); YOLOv2 does.  Round 4: their default is the f32 engine with
    # ODTK_F32X3 convolution descriptors (three bf16 MFMA products per f32 product): it passes the same gate at RANDOM INITIALISATION (minimum cosine 0.998 / 0.999
    # against the exact f32 engine) at 2.1x the exact engine's throughput (607 / 598 against 282 / 286 images/s)
    DEFAULT_ENGINE = 'f32x3'
    VGG_SEQ = VGG_SEQ                       # the trunk this class builds (pfpnet.PFPNetR stops at conv4_3)
    L2_AFTER = 'conv10_2'                   # creation order: the two L2-norm scalars follow the feature extractor (:77, :79)
    NAME = 'RefineDet'

    @staticmethod
    def layer_specs(num_classes):
        return layer_specs(num_classes)

    def __init__(self, config, data_provider):
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.input_size = config['input_size']
        self.data_shape = [self.input_size, self.input_size, 3] if config['data_format'] == 'channels_last' else [3, self.input_size, self.input_size]
        self.num_classes = config['num_classes'] + 1          # background = LAST index
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.anchor_ratios = [0.5, 1.0, 2.0]
        self.num_anchors = NA
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.pretraining_weight = config.get('pretraining_weight')
        assert self.input_size % 64 == 0, f"{self.NAME}: the input size must be a multiple of 64 (320 or 512 in the reference)"
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
        self.verbose = bool(config.get('verbose', True))
        self.dev = torch.device(config.get('device', 'cuda:0'))
        engine = config.get('compute_dtype', self.DEFAULT_ENGINE if (self.dev.type == 'cuda' and self.mode == 'train') else 'f32')
        # 'f32x3': f32 tensors, convolution descriptors of dtype ODTK_F32X3 (three bf16 MFMA products per f32 product where that is faster: include/odtk.h)
        self.DT = {'bf16': BF16, 'f32': F32, 'f32x3': F32}[engine]
        self.CDT = F32X3 if engine == 'f32x3' else self.DT
        self.tdt = torch.bfloat16 if self.DT == BF16 else torch.float32
        self.chunk = ops.chunk(self.DT)
        self.global_step = 0
        self.dist = None
        self.loss_divisor_batch = self.batch_size
        if self.dev.type == 'cuda':
            torch.cuda.set_device(self.dev)
        self.specs = self.layer_specs(self.num_classes)
        self._init_parameters(int(config.get('seed', 0)))
        self._build()
        self._warmup_setup(config, data_provider, 'compute_dtype' in config)
        self._load_pretraining_weight()

    # ------------------------------------------------------------------ parameters
    def _wshape(self, spec):
        _, kind, cin, cout, k, _, _, _ = spec
        kout, kin = (cout, cin) if kind != 'dconv' else (cin, cout)
        return (kout, k, k, ops.pad_to(kin, self.chunk)), kin

    def _extra_layer_params(self, spec):
        return ()

    def _init_parameters(self, seed):
        pinfo, sinfo = OrderedDict(), OrderedDict()
        off = soff = 0
        self._kin, self._kind = {}, {}

        def add(name, shape):
            nonlocal off
            pinfo[name] = (off, tuple(shape))
            off += ops.pad_to(int(np.prod(shape)), 64)
        for i, spec in enumerate(self.specs):
            name, kind, cout = spec[0], spec[1], spec[3]
            wshape, kin = self._wshape(spec)
            self._kin[name], self._kind[name] = kin, kind
            for extra, shape in self._extra_layer_params(spec):       # e.g. the depthwise filter of a separable layer (lhrcnn.LHRCNN), created first
                add(extra, shape)
            add(name + '.w', wshape); add(name + '.b', (cout,))
            if kind != 'vgg':
                add(name + '.gamma', (cout,)); add(name + '.beta', (cout,))
                for suffix in ('.mmean', '.mvar'):
                    sinfo[name + suffix] = (soff, (cout,))
                    soff += ops.pad_to(cout, 64)
            if name == self.L2_AFTER:
                add('feat1_l2_norm', (1,)); add('feat2_l2_norm', (1,))
        self.pinfo, self.sinfo, self.nparam = pinfo, sinfo, off
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.Mom = torch.zeros(off, device=dev)
        self.G = torch.zeros(off, device=dev)
        self.Pc = torch.zeros(off, dtype=self.tdt, device=dev) if self.DT == BF16 else self.P
        self.S = torch.zeros(soff, device=dev)
        self.l2_partial = torch.zeros(ops.sgd_blocks(off), device=dev)
        self.l2_sum = torch.zeros(1, device=dev)
        g = torch.Generator().manual_seed(seed)
        for name, kind, cin, cout, k, _, _, _ in self.specs:
            kout, kin = (cout, cin) if kind != 'dconv' else (cin, cout)
            self.set_param(name + '.w', torch.randn(kout, k, k, kin, generator=g) * math.sqrt(2.0 / (cin * k * k)))
            if kind != 'vgg':
                self.param(name + '.gamma').fill_(1.0)
                self.stat(name + '.mvar').fill_(1.0)
        if 'feat1_l2_norm' in pinfo:
            self.param('feat1_l2_norm').fill_(10.0)
            self.param('feat2_l2_norm').fill_(8.0)

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
                if k.endswith('.b') and self._kind.get(k[:-2]) == 'dconv' and float(torch.as_tensor(v).abs().max()) != 0.0:
                    raise NotImplementedError('non-zero bias of a transposed convolution (zero-initialised, in front of a batch norm: it never moves)')
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

    def _load_pretraining_weight(self):
        """the 13 VGG convolutions from slim's vgg_16.ckpt (RefineDet.py:33, :232-365), as ssd300.py does"""
        path = self.pretraining_weight
        if not path or not (os.path.exists(str(path)) or os.path.exists(str(path) + '.index')):
            return
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()                                   # weights are loaded: the run does not start from random initialisation
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        for l in self.VGG_SEQ:
            if isinstance(l, tuple):
                n = l[0]
                key = f'vgg_16/{n.split("_")[0]}/{n}'
                if reader.has_tensor(key + '/weights'):
                    self.set_param(n + '.w', torch.from_numpy(reader.get_tensor(key + '/weights')).permute(3, 0, 1, 2).contiguous())
                    self.set_param(n + '.b', torch.from_numpy(reader.get_tensor(key + '/biases')))
        self._refresh_operand_copies()

    # ------------------------------------------------------------------ the graph
    def _input_hw(self):
        return self.input_size, self.input_size

    def _build(self):
        """the graph engine: helpers that append to the forward plan (shared with pfpnet.PFPNetR and yolov2.YOLOv2), the model, the buffers"""
        N, dev, dt, ch = self.batch_size, self.dev, self.tdt, self.chunk
        Hin, Win = self._input_hw()
        spec = {s[0]: s for s in self.specs}
        self.images = torch.zeros(N, Hin, Win, 3, device=dev)
        self.input = _Act('input', N, Hin, Win, 3, ops.pad_to(3, ch), dt, dev)
        self.plan, self.desc, self.z, self.bnsave, self.acts = [], {}, {}, {}, {'input': self.input}
        self._max_ws = self._max_z = self._max_scr = 0
        self.use_graph = bool(self.config.get('use_graph', False))     # measured: replay is 6-7 % SLOWER than eager launches for these models
        self._graph, self._graph_gt, self._eager_steps = None, None, 0

        def act(name, H_, W_, C_, vgg=False):
            a = _Act(name, N, H_, W_, C_, ops.pad_to(C_, ch), dt, dev, vgg)
            self.acts[name] = a
            return a

        def vgg(name, src):
            _, _, cin, cout, k, _, _, _ = spec[name]
            self.desc[name] = ops.conv_desc(N, src.H, src.W, ops.pad_to(cin, ch), src.ld, cout, ops.pad_to(cout, ch), 3, 1, 1, self.CDT, self.CDT)
            y = act(name, src.H, src.W, cout, vgg=True)
            self._max_scr = max(self._max_scr, src.M * src.ld)
            self.plan.append(('vgg', name, src, y))
            return y

        def bn(name, src, out=None):
            """conv / transposed conv + batch norm (+ ReLU); out = (tensor, first row, width): the output goes straight into a prediction tensor"""
            _, kind, cin, cout, k, stride, dil, relu = spec[name]
            assert cin == src.C, (name, cin, src.C)
            ldz = ops.pad_to(cout, ch)
            if kind == 'conv':
                d = ops.conv_desc(N, src.H, src.W, ops.pad_to(cin, ch), src.ld, cout, ldz, k, stride, dil, self.CDT, self.CDT)
                Ho, Wo = d.Ho, d.Wo
            else:
                Ho, Wo = src.H * stride, src.W * stride
                d = ops.conv_desc(N, Ho, Wo, ldz, ldz, cin, src.ld, k, stride, 1, self.CDT, self.CDT)
                assert d.Ho == src.H and d.Wo == src.W
            self.desc[name] = d
            z = _Act(name + '.z', N, Ho, Wo, cout, ldz, dt, dev)
            self.z[name] = z
            if out is None:
                y = act(name, Ho, Wo, cout)
            else:
                y = _Act(name, N, Ho, Wo, cout, cout, torch.float32, dev, alloc=False)
                self.acts[name] = y
            self.bnsave[name] = (torch.zeros(cout, device=dev), torch.zeros(cout, device=dev))
            self._max_ws = max(self._max_ws, ops.bn_workspace_bytes(z.M, cout))
            self._max_z = max(self._max_z, z.M * ldz)
            self._max_scr = max(self._max_scr, src.M * src.ld)
            self.plan.append(('bn', name, kind, src, z, y, int(relu), out))          # 0 none | 1 ReLU | 2 leaky_relu(0.1)
            return y

        def pool(name, x, k, s):
            Ho, pt, _ = ops.same_pad(x.H, k, s)
            Wo, pl, _ = ops.same_pad(x.W, k, s)
            assert pl == pt, "the pool launch takes one pad for both axes"
            y = act(name, Ho, Wo, x.C, vgg=x.vgg)
            self._max_scr = max(self._max_scr, x.M * x.ld)
            self.plan.append(('pool', x, y, k, s, pt))
            return y

        def l2norm(name, x, gname):
            y = act(name, x.H, x.W, x.C)
            self.plan.append(('l2norm', x, y, gname))
            return y

        def resize(name, x, Ho, Wo):
            """tf.image.resize_bilinear(align_corners=True)"""
            y = act(name, Ho, Wo, x.C)
            self.plan.append(('resize', x, y))
            return y

        def avgpool(name, x):
            assert x.H % 2 == 0 and x.W % 2 == 0
            y = act(name, x.H // 2, x.W // 2, x.C)
            self._max_scr = max(self._max_scr, x.M * x.ld)
            self.plan.append(('avgpool', x, y))
            return y

        def add(name, a, b):
            assert (a.M, a.C, a.ld) == (b.M, b.C, b.ld) and not a.vgg and not b.vgg
            y = act(name, a.H, a.W, a.C)
            self.plan.append(('add', a, b, y))
            return y

        def concat(name, srcs):
            """tf.concat over the channels; the pieces need not start on 16-byte boundaries (odtk_copy_channels)"""
            y = act(name, srcs[0].H, srcs[0].W, sum(a.C for a in srcs))
            assert all(a.M == y.M for a in srcs)
            self.plan.append(('concat', tuple(srcs), y))
            return y
        def dw(name, src, kh, kw, stop_grad=False):
            """the depthwise half of tf.layers.separable_conv2d (filter `name`.dw [kh][kw][C], stride 1, SAME); the pointwise half is the 1x1 `bn` that follows.
            stop_grad: the input gradient is not formed (a consumer whose loss does not train the producer, LH_RCNN.py:190-191)"""
            mid = act(name + '.dw', src.H, src.W, src.C)
            self.plan.append(('dw', name, src, mid, kh, kw, stop_grad))
            return mid

        def sink(a):
            """`a` is read by a stage outside the plan, which writes a.g before the backward plan runs"""
            self.plan.append(('sink', a))
        from types import SimpleNamespace
        self._build_model(SimpleNamespace(vgg=vgg, bn=bn, pool=pool, l2norm=l2norm, resize=resize, avgpool=avgpool, add=add, concat=concat, act=act, dw=dw, sink=sink))
        self.ws = torch.zeros(self._max_ws, dtype=torch.uint8, device=dev)
        self.wt, entries = {}, []
        for sp in self.specs:
            name, kind, cin, cout, k = sp[0], sp[1], sp[2], sp[3], sp[4]
            if name == self.specs[0][0]:                        # the first layer: the input needs no gradient
                continue
            (kout, _, _, kin_pad), _ = self._wshape(sp)
            kp = ops.pad_to(kout, ch)
            self.wt[name] = torch.zeros(kin_pad * k * k * kp, dtype=dt, device=dev)
            entries.append((self._flat(name + '.w', self.P), self.wt[name], kout, k, k, kin_pad, kp))
        self._fp_batch = ops.FilterPrepareBatch(entries, self.DT, dev)
        if self.mode == 'train':
            self._build_backward(N, dt, dev)
        self._refresh_operand_copies()

    def _build_model(self, h):
        """RefineDet.py:72-158: anchors, the four prediction tensors, features, ARM heads, the top-down TCB chain, ODM heads"""
        N, dev = self.batch_size, self.dev
        self.anc = heads.refinedet_anchors(self.input_size, dev)         # y1x1, y2x2, yx, hw, nmsbox
        A = self.anc[0].shape[0]
        self.A = A
        C = self.num_classes
        self.arm_loc = torch.zeros(N, A, 4, device=dev); self.arm_conf = torch.zeros(N, A, 2, device=dev)
        self.odm_loc = torch.zeros(N, A, 4, device=dev); self.odm_conf = torch.zeros(N, A, C, device=dev)
        bn, act = h.bn, h.act
        f = self._build_features(h)
        self.level_off, off = [], 0
        for a in f:
            self.level_off.append(off)
            off += a.H * a.W * NA
        assert off == A, (off, A)

        def head(prefix, x, lvl, loc_t, conf_t, ncls):
            c = x
            for j in range(1, 5):
                c = bn(f'{prefix}.c{j}', c)
            bn(f'{prefix}.loc', c, (loc_t, self.level_off[lvl], 4))
            bn(f'{prefix}.conf', c, (conf_t, self.level_off[lvl], ncls))
        for l in range(4):
            head(f'arm{l + 1}', f[l], l, 'arm_loc', 'arm_conf', 2)
        tcb = {}
        for l in (4, 3, 2, 1):
            c2 = bn(f'tcb{l}.c2', bn(f'tcb{l}.c1', f[l - 1]))
            if l == 4:
                tcb[l] = c2
            else:
                d_ = bn(f'tcb{l}.d', tcb[l + 1])
                y = act(f'tcb{l}', c2.H, c2.W, c2.C)
                self.plan.append(('add_relu', c2, d_, y))
                tcb[l] = y
        for l in range(4):
            head(f'odm{l + 1}', tcb[l + 1], l, 'odm_loc', 'odm_conf', C)

    def _build_features(self, h):
        """-> the four feature activations of the ARM / TCB (RefineDet.py:232-385, :74-95): conv4_3 and conv5_3 L2-normalised, conv8_2, conv10_2"""
        x = self.input
        feats = {}
        for l in self.VGG_SEQ:
            if isinstance(l, tuple):
                x = h.vgg(l[0], x)
                feats[l[0]] = x
            else:
                x = h.pool(l, x, 3, 1) if l == 'pool5' else h.pool(l, x, 2, 2)
        for e in EXTRAS:
            x = h.bn(e[0], x)
            feats[e[0]] = x
        return [h.l2norm('feat1', feats['conv4_3'], 'feat1_l2_norm'), h.l2norm('feat2', feats['conv5_3'], 'feat2_l2_norm'), feats['conv8_2'], feats['conv10_2']]

    def _pred_view(self, out, grad=False):
        """(flat tensor starting at this level's first row, row width, rows per image of the level is implied by the caller, image stride)"""
        tname, first, width = out
        t = getattr(self.loss, 'd_' + tname) if grad else getattr(self, tname)
        return t.view(-1)[first * width:], width, self.A * width

    def _build_backward(self, N, dt, dev):
        self.loss = None                                        # heads.RefineDetLoss, built with the first batch (needs the ground-truth pad length)
        self.zg = torch.zeros(self._max_z, dtype=dt, device=dev)
        self.scr = torch.zeros(self._max_scr, dtype=dt, device=dev)
        written = set()

        def emit(a):
            acc = id(a) in written
            written.add(id(a))
            if a.g is None and a is not self.input:
                a.g = torch.zeros(a.M, a.ld, dtype=dt, device=dev)
            return acc
        self.bplan = []
        for op in reversed(self.plan):
            kind = op[0]
            if kind == 'bn':
                _, name, lk, src, z, y, relu, out = op
                assert out is not None or id(y) in written, name
                self.bplan.append(('bn', name, lk, src, z, y, relu, out, emit(src)))
            elif kind == 'vgg':
                _, name, src, y = op
                assert id(y) in written, name
                self.bplan.append(('vgg', name, src, y, emit(src) if src is not self.input else False))
            elif kind == 'pool':
                _, x, y, k, s, pt = op
                assert id(y) in written
                self.bplan.append(('pool', x, y, k, s, pt, emit(x)))
            elif kind == 'l2norm':
                _, x, y, gname = op
                assert id(y) in written
                self.bplan.append(('l2norm', x, y, gname, emit(x)))
            elif kind in ('resize', 'avgpool'):
                _, x, y = op
                assert id(y) in written
                self.bplan.append((kind, x, y, emit(x)))
            elif kind == 'add':
                _, a, b, y = op
                assert id(y) in written
                self.bplan.append(('add', (a, emit(a)), (b, emit(b)), y))
            elif kind == 'concat':
                _, srcs, y = op
                assert id(y) in written
                self.bplan.append(('concat', tuple((a, emit(a)) for a in srcs), y))
            elif kind == 'dw':
                _, name, src, mid, kh, kw, stop = op
                assert id(mid) in written, name
                self.bplan.append(('dw', name, src, mid, kh, kw, stop, False if stop else emit(src)))
            elif kind == 'sink':
                emit(op[1])
            else:
                _, a, b, y = op
                assert id(y) in written
                self.bplan.append(('add_relu', (a, emit(a)), (b, emit(b)), y))
        self.gt = None

    # ------------------------------------------------------------------ forward / loss / backward
    def _preprocess_input(self, subtract_mean):
        ops.preprocess(self.images, MEAN_RGB if subtract_mean else (0., 0., 0.), self.input.ld, self.DT, self.input.t)

    def _forward(self, training, subtract_mean=True):
        self._preprocess_input(subtract_mean)
        for op in self.plan:
            kind = op[0]
            if kind == 'sink':
                continue
            if kind == 'dw':
                _, name, src, mid, kh, kw, _ = op
                ops.depthwise_conv(src.t, src.ld, self.param(name + '.dw'), mid.t, mid.ld, src.N, src.H, src.W, src.C, kh, kw)
            elif kind == 'vgg':
                _, name, src, y = op
                ops.conv2d_fwd(self.desc[name], src.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), y.t, True)
            elif kind == 'bn':
                _, name, lk, src, z, y, relu, out = op
                if lk == 'conv':
                    ops.conv2d_fwd(self.desc[name], src.t, self._flat(name + '.w', self.Pc), self.param(name + '.b'), z.t, False)
                else:
                    ops.conv2d_dgrad(self.desc[name], src.t, src.ld, self.wt[name], None, z.t, False)
                sm, si = self.bnsave[name]
                if out is None:
                    dst, ldy, rpi, stride = y.t, y.ld, z.M, 0
                else:
                    dst, _, stride = self._pred_view(out)
                    ldy, rpi = z.C, z.H * z.W                   # a cell's NA * width outputs are contiguous rows of the prediction tensor
                ops.bn_fwd(z.t, z.M, z.C, z.ld, self.param(name + '.gamma'), self.param(name + '.beta'), self.stat(name + '.mmean'),
                           self.stat(name + '.mvar'), sm, si, training, relu, dst, ldy, rpi, stride, self.ws)
            elif kind == 'pool':
                _, x, y, k, s, pt = op
                ops.maxpool_fwd(x.t, y.t, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
            elif kind == 'l2norm':
                _, x, y, gname = op
                ops.l2norm_fwd(x.t, y.t, x.M, x.ld, x.ld, self.param(gname))        # (pad channels are zero: they do not change the norm)
            elif kind == 'resize':
                _, x, y = op
                ops.resize_bilinear2_fwd(x.t, x.ld, y.t, y.ld, x.N, x.H, x.W, y.H, y.W, x.ld, True)
            elif kind == 'avgpool':
                _, x, y = op
                ops.avgpool2x2_fwd(x.t, y.t, x.N, x.H, x.W, x.ld)
            elif kind == 'add':
                _, a, b, y = op
                ops.add2d(a.t, a.ld, b.t, b.ld, y.t, y.ld, y.M, y.ld)
            elif kind == 'concat':
                _, srcs, y = op
                off = 0
                for a in srcs:
                    ops.copy_channels(a.t, a.ld, 0, y.t, y.ld, off, y.M, a.C)
                    off += a.C
            else:
                _, a, b, y = op
                ops.add_relu_fwd(a.t, a.ld, b.t, b.ld, y.t, y.ld, y.M, y.ld)

    def _into(self, a, acc):
        return self.scr[: a.M * a.ld].view(a.M, a.ld) if acc else a.g

    def _fold(self, a, acc):
        if acc:
            ops.add2d(a.g, a.ld, self.scr[: a.M * a.ld].view(a.M, a.ld), a.ld, a.g, a.ld, a.M, a.ld)

    def _backward_iter(self):
        for op in self.bplan:
            kind = op[0]
            if kind == 'bn':
                _, name, lk, src, z, y, relu, out, acc = op
                zg = self.zg[: z.M * z.ld].view(z.M, z.ld)
                sm, si = self.bnsave[name]
                if out is None:
                    dy, ldy, rpi, stride, yt = y.g, y.ld, z.M, 0, (y.t if relu else None)
                else:
                    dy, _, stride = self._pred_view(out, grad=True)
                    ldy, rpi, yt = z.C, z.H * z.W, None
                ops.bn_bwd(z.t, yt, dy, z.M, z.C, z.ld, ldy, rpi, stride, self.param(name + '.gamma'), sm, si, relu, zg,
                           self._flat(name + '.gamma', self.G), self._flat(name + '.beta', self.G), self.ws)
                mask = src.t if src.vgg else None
                if lk == 'conv':
                    ops.conv2d_wgrad(self.desc[name], src.t, zg, z.ld, self._flat(name + '.w', self.G), None)
                    if src is not self.input:
                        ops.conv2d_dgrad(self.desc[name], zg, z.ld, self.wt[name], mask, src.g, acc)
                else:
                    ops.conv2d_wgrad(self.desc[name], zg, src.t, src.ld, self._flat(name + '.w', self.G), None)
                    ops.conv2d_fwd(self.desc[name], zg, self._flat(name + '.w', self.Pc), None, self._into(src, acc), False)
                    self._fold(src, acc)
                yield name
            elif kind == 'vgg':
                _, name, src, y, acc = op
                ops.conv2d_wgrad(self.desc[name], src.t, y.g, y.ld, self._flat(name + '.w', self.G), self._flat(name + '.b', self.G))
                if src is not self.input:
                    ops.conv2d_dgrad(self.desc[name], y.g, y.ld, self.wt[name], src.t if src.vgg else None, src.g, acc)
                yield name
            elif kind == 'pool':
                _, x, y, k, s, pt, acc = op
                ops.maxpool_bwd(x.t, y.t, y.g, self._into(x, acc), x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
                self._fold(x, acc)
            elif kind == 'l2norm':
                _, x, y, gname, acc = op
                ops.l2norm_bwd(x.t, y.g, x.g, x.M, x.ld, x.ld, self.param(gname), self._flat(gname, self.G), acc, x.t if x.vgg else None)
                yield gname
            elif kind == 'resize':
                _, x, y, acc = op
                ops.resize_bilinear2_bwd(y.g, y.ld, x.g, x.ld, x.N, x.H, x.W, y.H, y.W, x.ld, True, acc, x.t if x.vgg else None)
            elif kind == 'avgpool':
                _, x, y, acc = op
                ops.avgpool2x2_bwd(y.g, self._into(x, acc), x.N, x.H, x.W, x.ld)
                self._fold(x, acc)
            elif kind == 'add':
                _, (a, acc_a), (b, acc_b), y = op
                for t, acc in ((a, acc_a), (b, acc_b)):
                    ops.add2d(y.g, y.ld, t.g if acc else None, t.ld, t.g, t.ld, y.M, y.ld)
            elif kind == 'concat':
                _, srcs, y = op
                off = 0
                for a, acc in srcs:
                    ops.copy_channels(y.g, y.ld, off, a.g, a.ld, 0, y.M, a.C, acc, a.t if a.vgg else None)
                    off += a.C
            elif kind == 'dw':
                _, name, src, mid, kh, kw, stop, acc = op
                ops.depthwise_wgrad(src.t, src.ld, mid.g, mid.ld, self._flat(name + '.dw', self.G), src.N, src.H, src.W, src.C, kh, kw)
                if not stop:
                    ops.depthwise_conv(mid.g, mid.ld, self.param(name + '.dw'), src.g, src.ld, src.N, src.H, src.W, src.C, kh, kw, True, acc)
                yield name + '.dw'
            else:
                _, (a, acc_a), (b, acc_b), y = op
                ops.relu_bwd(y.t, y.g, y.ld, a.g, a.ld, y.M, y.ld, acc_a)
                ops.relu_bwd(y.t, y.g, y.ld, b.g, b.ld, y.M, y.ld, acc_b)

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
            self.loss = self._make_loss(gt.shape[1])
        self.gt.copy_(gt, non_blocking=True)

    def _make_loss(self, pad):
        return heads.RefineDetLoss(self.anc, self.batch_size, self.num_classes, pad, self.dev)

    def _loss_step(self):
        """loss kernels on the predictions of _forward (gradients into self.loss.d_*) -> sum of the per-image losses / batch, a device scalar"""
        parts = self.loss(self.arm_loc, self.arm_conf, self.odm_loc, self.odm_conf, self.gt, 1.0 / self.loss_divisor_batch)
        return parts[:, 6].sum() / self.batch_size

    def _step_body(self):
        self.G.zero_()
        self._forward(True)
        self._data_loss = self._loss_step()
        for name in self._backward_iter():
            # gradient segments of the all-reduce are the blocks before the first '.' (arm1, tcb3, ...); a block is final when its FIRST
            # layer in creation order (.c1) has been processed -- backward walks a block's layers in reverse
            if self.dist is not None and ('.' not in name or name.endswith('.c1')):
                self.dist.layer_ready(name.split('.')[0])

    def _train_step_engine(self, lr):
        """one MomentumOptimizer step on the batch of set_batch(); returns the loss (data + L2) as a 1-element device tensor.
        Single device, config key 'use_graph' (default OFF): after two eager steps (the library's lazily grown scratch buffers exist by then) forward +
        loss + backward replay from ONE HIP graph; the optimizer launches stay outside (lr is a launch argument).  Measured on MI355X at batch 32 bf16,
        same box, replay | eager: YOLOv2 13.4 | 12.55 ms, RefineDet320 21.4 | 20.1 ms -- the host keeps the queue full in eager mode (the step is not
        launch-bound) and a graph's kernel nodes cost more per node than queued launches, so eager stays the default.  Data parallel: eager."""
        if self.dist is not None:
            self.dist.begin_step()
        if self.use_graph and self.dist is None and self.dev.type == 'cuda' and self._eager_steps >= 2:
            if self._graph is None or self._graph_gt is not self.gt:
                self._graph = torch.cuda.CUDAGraph()
                self._graph_gt = self.gt                       # the captured launches hold this buffer's pointer and pad length
                with torch.cuda.graph(self._graph):
                    self._step_body()
            self._graph.replay()
        else:
            self._step_body()
            self._eager_steps += 1
        if self.dist is not None:
            self.dist.finish_step()
        ops.sgd_momentum(self.P, self.Mom, self.G, lr, 0.9, self.weight_decay, 1.0, self.l2_partial, self.Pc if self.DT == BF16 else None)
        ops.sum_f32(self.l2_partial, self.l2_sum)
        self._fp_batch.run()
        self.global_step += 1
        return self._data_loss + self.weight_decay * self.l2_sum      # RefineDet.py:180-184 (pre-update weights)

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
        self._forward(False, subtract_mean=bool(self.config.get('test_subtract_mean', False)))      # reference quirk: the fed tensor is `images - mean`
        scores, bbox, cid = heads.refinedet_detect(self.arm_loc[0], self.arm_conf[0], self.odm_loc[0], self.odm_conf[0], self.anc[2], self.anc[3],
                                                   self.nms_score_threshold, self.nms_max_boxes, self.nms_iou_threshold)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    # ------------------------------------------------------------------ checkpoints / data parallel
    # ------------------------------------------------------------------ the reference's variable names / tf.train.Saver files
    VGG_NAME_TYPOS = {'conv2_1.w': 'kenrel_conv2_1', 'conv3_1.b': 'bias_conv_3_1'}       # as spelt in the reference (RefineDet.py:251, :271; PFPNetR.py alike)
    MOMENTUM_SLOT_SCOPE = 'inference/'      # the optimizer is created inside `with tf.variable_scope('inference')` (RefineDet.py:97, :176)

    def _tf_scope(self, name):
        """(variable scope, explicit layer name or None) of a layer of self.specs: default layer names (conv2d, conv2d_transpose, batch_normalization) are
        numbered per enclosing scope in creation order"""
        if '.' in name:
            block = name.split('.')[0]
            return {'arm': 'ARM', 'tcb': 'TCB', 'odm': 'ODM'}[block[:3]] + '/' + block, None
        return 'feature_extractor', (name if name.startswith('conv') else None)

    def reference_variable_map(self):
        """our parameter / statistic name -> the variable name in the reference's graph (checked against the shim's graph: tests/golden/*_names.json)"""
        out, count = OrderedDict(), {}

        def numbered(scope, base):
            k = count.get((scope, base), 0)
            count[(scope, base)] = k + 1
            return f'{scope}/{base}' + (f'_{k}' if k else '')
        for name, kind, *_ in self.specs:
            if kind == 'vgg':
                out[name + '.w'] = 'feature_extractor/' + self.VGG_NAME_TYPOS.get(name + '.w', 'kernel_' + name)
                out[name + '.b'] = 'feature_extractor/' + self.VGG_NAME_TYPOS.get(name + '.b', 'bias_' + name)
                continue
            scope, explicit = self._tf_scope(name)
            layer = f'{scope}/{explicit}' if explicit else numbered(scope, 'conv2d_transpose' if kind == 'dconv' else 'conv2d')
            bn = numbered(scope, 'batch_normalization')
            out[name + '.w'], out[name + '.b'] = layer + '/kernel', layer + '/bias'
            for a, t in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
                out[f'{name}.{a}'] = f'{bn}/{t}'
        for k in ('feat1_l2_norm', 'feat2_l2_norm'):
            if k in self.pinfo:
                out[k] = 'feature_extractor/' + k
        return out

    def _logical(self, name, buf):
        """parameter `name` out of a flat buffer (P or Mom) in TensorFlow's layout: kernels HWIO (transposed convs: [h, w, out, in]), un-padded"""
        v = self.get_param(name, buf)
        return np.ascontiguousarray((v.permute(1, 2, 3, 0) if name.endswith('.w') else v).numpy())

    def export_tf_variables(self):
        """what the reference's `tf.train.Saver()` would write: every global variable -- weights, moving statistics, global_step and the MomentumOptimizer slots"""
        self._sync_from_twin()
        out = OrderedDict()
        for ours, tfname in self.reference_variable_map().items():
            if ours in self.pinfo:
                out[tfname] = self._logical(ours, self.P)
                out[f'{self.MOMENTUM_SLOT_SCOPE}{tfname}/Momentum'] = self._logical(ours, self.Mom)
            else:
                out[tfname] = self.stat(ours).detach().cpu().numpy().copy()
        out['global_step'] = np.asarray(self.global_step, dtype=np.int32)
        return out

    def load_tf_checkpoint(self, path):
        """`saver.restore(sess, path)` from the files of a reference-trained model (or ours)"""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = reader.get_variable_to_shape_map()
        for ours, tfname in self.reference_variable_map().items():
            if ours in self.pinfo:
                v = torch.from_numpy(reader.get_tensor(tfname))                 # KeyError = Saver's NotFoundError
                self.set_param(ours, v.permute(3, 0, 1, 2).contiguous() if ours.endswith('.w') else v)
                slot = [k for k in names if k.endswith(tfname + '/Momentum')]
                if slot:
                    mv = torch.from_numpy(reader.get_tensor(slot[0]))
                    dst = self.param(ours, self.Mom)
                    if ours.endswith('.w'):
                        mv = mv.permute(3, 0, 1, 2)
                        dst.zero_()
                        dst[..., : mv.shape[-1]] = mv.to(self.dev)
                    else:
                        dst.copy_(mv.to(self.dev).view(dst.shape))
            else:
                self.stat(ours).copy_(torch.from_numpy(reader.get_tensor(tfname)).to(self.dev))
        if reader.has_tensor('global_step'):
            self.global_step = int(reader.get_tensor('global_step'))
        self._refresh_operand_copies()

    def _save_weight_engine(self, mode, path):
        """config['checkpoint_format'] = 'tf' writes the reference's own files (`<path>-<step>.index` + `.data-00000-of-00001` + `checkpoint`, readable by its
        `load_weight`); the default keeps one torch file `<path>-<step>`."""
        assert (mode in ['latest', 'best'])
        if self.config.get('checkpoint_format', 'torch') == 'tf':
            from . import tf_checkpoint
            dirname = os.path.dirname(path)
            if dirname and not os.path.exists(dirname):
                os.makedirs(dirname)
                print(dirname, 'does not exist, create it done')
            prefix = path + '-' + str(self.global_step)
            tf_checkpoint.write_bundle(prefix, self.export_tf_variables())
            tf_checkpoint.update_checkpoint_state(prefix)
            print('save', mode, 'model in', path, 'successfully')
            return
        dirname = os.path.dirname(path)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
            print(dirname, 'does not exist, create it done')
        blob = {'params': self.export_params(), 'momentum': self.Mom.detach().cpu(), 'global_step': self.global_step,
                'layout': {k: (int(o), tuple(int(x) for x in shp)) for k, (o, shp) in self.pinfo.items()}}
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

    def attach_data_parallel(self, group=None, bucket_mb=25, grad_dtype='f32', force_collectives=False, collective='torch'):
        from .dist import GradAllReducer
        self.dist = GradAllReducer(self, group, bucket_mb, grad_dtype, force_collectives, collective)
        self.loss_divisor_batch = self.batch_size * self.dist.world
        return self.dist
