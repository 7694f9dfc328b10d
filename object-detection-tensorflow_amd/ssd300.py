"""SSD300 (VGG-16) behind the reference class surface.

Drop-in for /root/reference/SSD300.py: same constructor (`config`, `data_provider` dicts with
the keys of testSSD300.py:21-59), same methods -- train_one_epoch(lr) -> mean loss,
test_one_image(images) -> [scores, bbox, class_id], save_weight(mode, path), load_weight(path).
The TF-1.13 graph + sess.run is replaced by explicit HIP kernel launches through libodtk
(include/odtk.h); torch only owns device memory, streams and the RCCL communicator.

Data contract (replaces the tf.data iterator): `data_provider['train_generator']`
is a re-iterable (or a `(initializer, iterable)` pair, mirroring the reference tuple) yielding
`(images f32 [B,300,300,3] RGB 0..255, ground_truth f32 [B,pad,5] = [yc,xc,h,w,cls] px, pad rows -1)`
as torch tensors or numpy arrays -- exactly what utils/image_augmentor.py:24-27 documents.

Extra, optional config keys (absent in the reference): 'compute_dtype' ('bf16' | 'f32' | 'f32x3'; default bf16 in train mode, f32 in test mode),
'device', 'seed', 'verbose', 'test_subtract_mean' (False = reproduce the reference's test-mode feed quirk),
'use_graph' (False, default since round 3: eager launches -- measured 1-3 % FASTER than graph replay on every box once the step was down to ~200
launches (profiles/r03e_launch_mode_ab.md) | True: replay the step's kernel launches from HIP graphs after two eager steps | 'auto' = the faster of the two,
measured at start-up),
'tail_stream' (True: the six heads run on a second stream beside the extra-layer chain).
"""
from __future__ import annotations

import contextlib
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops
from ._lib import BF16, F32, F32X3
from .warmup import F32Warmup

INPUT_SIZE = 300
MEAN_RGB = (123.68, 116.779, 103.979)            # reference SSD300.py:55 (sic: 103.979)
FEATURE_SIZES = [38, 19, 10, 5, 5, 3]            # conv10_2 has stride 1 (reference SSD300.py:311)
ANCHORS_PER_CELL = [4, 6, 6, 6, 4, 4]
ASPECTS = [[2, 1 / 2], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2], [2, 1 / 2]]
NUM_PRIORS = sum(f * f * a for f, a in zip(FEATURE_SIZES, ANCHORS_PER_CELL))   # 8828

# reference SSD300.py:193-303: (name, cin, cout) convs (3x3 s1 SAME, bias, ReLU, no BN) and pools
VGG_SEQ = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64), ("pool1", 2, 2),
    ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("pool2", 2, 2),
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("pool3", 2, 2),
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("pool4", 2, 2),
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), ("pool5", 3, 1),
]
# reference SSD300.py:304-313: conv + bias + BN + ReLU     (name, cin, cout, k, stride, dil)
EXTRA_SEQ = [
    ("conv6", 512, 1024, 3, 1, 2), ("conv7", 1024, 1024, 1, 1, 1),
    ("conv8_1", 1024, 256, 1, 1, 1), ("conv8_2", 256, 512, 3, 2, 1),
    ("conv9_1", 512, 128, 1, 1, 1), ("conv9_2", 128, 256, 3, 2, 1),
    ("conv10_1", 256, 128, 1, 1, 1), ("conv10_2", 128, 256, 3, 1, 1),
    ("conv11_1", 256, 128, 1, 1, 1), ("conv11_2", 128, 256, 3, 2, 1),
]
FEAT_SRC = ["feat1", "conv7", "conv8_2", "conv9_2", "conv10_2", "conv11_2"]   # reference SSD300.py:314


def reference_variable_map(extra_seq=None, n_heads=6):
    """name of every variable of the reference's graph -> our parameter / statistic name.
    SSD300.py:193-300 (`kernel_convX_Y` / `bias_convX_Y` under 'feature_extractor', with the two misspelt names
    `kenrel_conv2_1` :212 and `bias_conv_3_1` :232), :77 l2_norm_factor, :304-313 + :85-90 tf.layers.conv2d
    (`<scope>/<name>/kernel|bias`) each followed by tf.layers.batch_normalization, whose default layer name is made
    unique PER ENCLOSING variable scope (Layer._set_scope -> variable_scope(None, default_name='batch_normalization')):
    feature_extractor/batch_normalization, _1 ... _9 and regressor/batch_normalization, _1 ... _5."""
    m = OrderedDict()
    for item in VGG_SEQ:
        n = item[0]
        if n.startswith('conv'):
            m['feature_extractor/' + ('kenrel_' if n == 'conv2_1' else 'kernel_') + n] = n + '.w'
            m['feature_extractor/' + ('bias_conv_3_1' if n == 'conv3_1' else 'bias_' + n)] = n + '.b'
    m['feature_extractor/l2_norm_factor'] = 'l2norm.gamma'
    extra_seq = EXTRA_SEQ if extra_seq is None else extra_seq
    for scope, names in (('feature_extractor', [e[0] for e in extra_seq]), ('regressor', [f'pred{i}' for i in range(1, n_heads + 1)])):
        for bn, n in enumerate(names):
            bns = f'{scope}/batch_normalization' + (f'_{bn}' if bn else '')
            m[f'{scope}/{n}/kernel'], m[f'{scope}/{n}/bias'] = n + '.w', n + '.b'
            m[bns + '/gamma'], m[bns + '/beta'] = n + '.gamma', n + '.beta'
            m[bns + '/moving_mean'], m[bns + '/moving_variance'] = n + '.mmean', n + '.mvar'
    return m


def prior_scales(input_size):
    """SSD300.py:112-113: the (s_k, sqrt(s_k s_k+1)) pair of every level, python doubles"""
    s = [(0.2 + (0.9 - 0.2) / 5 * (i - 1)) * input_size for i in range(1, 8)]
    return [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 6)]


def prior_spec(input_size=INPUT_SIZE, scales=None, aspects=None, feature_sizes=None, anchors_per_cell=None):
    """Host part of SSD300._get_abbox (reference SSD300.py:112-119, 333-336): the python-double
    (h, w) list per level, flattened; the per-cell arithmetic runs in odtk_ssd_priors."""
    s = prior_scales(input_size) if scales is None else scales
    aspects = ASPECTS if aspects is None else aspects
    feature_sizes = FEATURE_SIZES if feature_sizes is None else feature_sizes
    anchors_per_cell = ANCHORS_PER_CELL if anchors_per_cell is None else anchors_per_cell
    flat = []
    for size, ar in zip(s, aspects):
        pr = [[size[0], size[0]], [size[1], size[1]]]
        for a in ar:
            pr.append([size[0] * (a ** 0.5), size[0] / (a ** 0.5)])
        for h, w in pr:
            flat += [h, w]
    return feature_sizes, anchors_per_cell, flat


_SIDE_STREAMS = {}


def _side_stream(dev, role):
    """The side streams of a device are created ONCE and shared by every model instance of the process (instances step one after the other).  The HIP
    runtime deals streams onto four hardware queues (GPU_MAX_HW_QUEUES): a second instance with its own streams -- the f32 warm-up twin of warmup.py, a
    validation model -- got side streams that alias the hardware queue of the main stream, and its head / filter-gradient launches serialised with the trunk:
    1.4-2 % slower for the 2nd instance, 7 % for the 4th / 6th / 8th, whatever the configuration (round 3, tools/ab_bench.py --own-streams)."""
    key = (str(dev), role)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class _Act:
    """One NHWC activation: rows x pitch buffer (+ lazily allocated gradient buffer)."""

    def __init__(self, N, H, W, C, ld, dtype, dev):
        self.N, self.H, self.W, self.C, self.ld = N, H, W, C, ld
        self.M = N * H * W
        self.t = torch.zeros(self.M, ld, dtype=dtype, device=dev)
        self._g = None

    @property
    def g(self):
        if self._g is None:
            self._g = torch.zeros_like(self.t)
        return self._g


class _Conv:
    def __init__(self, name, cin, cout, k, stride, dil, bn, relu):
        self.name, self.cin, self.cout, self.k, self.stride, self.dil = name, cin, cout, k, stride, dil
        self.bn, self.relu = bn, relu


class SSD300(F32Warmup):
    # the variant: SSD512 (ssd512.py) overrides these
    INPUT_SIZE = INPUT_SIZE
    FEATURE_SIZES = FEATURE_SIZES
    ANCHORS_PER_CELL = ANCHORS_PER_CELL
    ASPECTS = ASPECTS
    EXTRA_SEQ = EXTRA_SEQ
    FEAT_SRC = FEAT_SRC
    FEAT_CH = [512, 1024, 512, 256, 256, 256]

    @classmethod
    def prior_scales(cls):
        return prior_scales(cls.INPUT_SIZE)

    @property
    def NH(self):
        return len(self.FEAT_SRC)

    @property
    def NUM_PRIORS(self):
        return sum(f * f * a for f, a in zip(self.FEATURE_SIZES, self.ANCHORS_PER_CELL))

    def __init__(self, config, data_provider):
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.input_size = self.INPUT_SIZE
        self.data_format = config['data_format']
        self.data_shape = [self.INPUT_SIZE, self.INPUT_SIZE, 3] if self.data_format == 'channels_last' else [3, self.INPUT_SIZE, self.INPUT_SIZE]
        self.num_classes = config['num_classes'] + 1          # background = LAST index
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']                  # unused, as in the reference
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.pretraining_weight = config['pretraining_weight']
        assert self.num_classes + 4 == 25 or True
        self.row = self.num_classes + 4

        # training defaults to the bf16 engine (the benchmarked configuration); inference to f32: north_star's 1e-3 bound on
        # boxes / scores holds for the f32 engine only (tests/test_gpu_ssd300_b32.py::test_bf16_test_one_image_vs_oracle)
        # 'f32x3' (round 5, the engine the other seven classes have had since round 4): f32 tensors, convolution descriptors of dtype ODTK_F32X3 -- the library runs
        # a layer's passes as three bf16 MFMA products per f32 product (hi*hi + hi*lo + lo*hi, f32 accumulation: 3e-5 of the exact f32 result) where that is
        # faster than the exact-f32 MFMA: the engine that meets north_star's 1e-3 on boxes / scores at a multiple of the exact engine's rate
        # Round 6: SSD300 went through the bf16 admission gate of the other classes (tests/test_gpu_bf16_gate.py; profiles/r06_bf16_gate_table.md): from the synthetic
        # He initialisation the bf16 engine's filter gradients sit at cosine 0.78 (minimum) / 0.85 (input-side third) against the f32 engine's -- under the bar --
        # and clear it after 300 f32 steps.  So, as for FCOS / CenterNet / YOLOv2 / YOLOv3 (warmup.py): when NO engine is named, training starts from random
        # initialisation (no VGG checkpoint found, no weights loaded) and the device is a GPU, the first `f32_warmup_steps` (300) optimizer steps run on an f32x3 twin;
        # an explicit 'compute_dtype' (bench.py, every test) is taken literally
        cd = config.get('compute_dtype', 'bf16' if config['mode'] == 'train' else 'f32')
        assert cd in ('bf16', 'f32', 'f32x3')
        self.DT = BF16 if cd == 'bf16' else F32
        self.CDT = F32X3 if cd == 'f32x3' else self.DT       # what the convolution descriptors carry
        self.tdt = ops.torch_dtype(self.DT)
        self.chunk = ops.chunk(self.DT)
        self.verbose = config.get('verbose', True)
        dev = config.get('device', None)
        if dev is None:
            if not torch.cuda.is_available():
                raise ops._lib.OdtkError("SSD300 needs an MI355X (HIP device); there is no CPU fallback")
            dev = torch.device('cuda', torch.cuda.current_device())
        self.dev = torch.device(dev)
        ops._lib.load()

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
        self.sync_bn = None
        self.checkpoint_format = config.get('checkpoint_format', 'torch')          # 'tf': tf.train.Saver files (tf_checkpoint.py)
        # HIP-graph replay of the step after 2 eager steps: True (default) | False | 'auto'.  'auto' builds the graphs, then
        # times AUTO_STEPS steps replayed and AUTO_STEPS steps launched eagerly (real training steps, synchronised only
        # while calibrating) and keeps the faster mode: replay wins when the host cannot issue ~220 launches per step as
        # fast as the GPU retires them.  On a host that can, the two modes measured within 1 % of each other (9.63-9.66 ms
        # eager vs 9.75-9.82 ms replayed, same box), so replay -- which does not depend on the host -- is the default.
        ug = config.get('use_graph', False)
        self.use_list = ug == 'list'             # recorded launch list: the step's C-ABI calls replayed from pre-bound argument tuples (_lib.record_begin)
        self._cmds = None
        self._events = {}
        self.use_graph = True if ug == 'auto' else (False if ug == 'list' else bool(ug))
        self._auto = {'left': 2 * self.AUTO_STEPS, 't': {}} if ug == 'auto' else None
        # optional: filter gradients on a second HIP stream (wgrad(L) only needs dy(L) and the stored input of L, nothing
        # on the dgrad chain needs its result before the optimizer).  Measured neutral on MI355X (both chains are
        # full-chip kernels with one workgroup per CU), so it is off by default; config key 'wgrad_stream'.
        on_gpu = self.dev.type == 'cuda'      # (a 'cpu' device only gets past ops._p with the mocked library of tests/mock_ops.py: host-logic tests)
        # config key 'deterministic_wgrad' (default since round 6: True = what the library does anyway): filter gradients by partial tiles + a fixed-order reduction,
        # the whole step bit-identical from run to run; False = float atomics (round 1-5's default; the step time is the same to 0.1 %: 8.107 / 8.13 / 8.12 ms
        # against 8.097 / 8.121 / 8.117, gpurun r6e).  A process-wide switch of the library (include/odtk.h, odtk_debug_set key 5): a model built with the
        # key sets it for every model of the process until another one does
        if on_gpu and 'deterministic_wgrad' in config:
            ops.debug_set(5, 1 if config['deterministic_wgrad'] else 0)
        self.wgrad_stream = _side_stream(self.dev, 'wgrad') if (on_gpu and config.get('wgrad_stream', False)) else None
        self._side = _side_stream(self.dev, 'match') if on_gpu else None             # box matching under the forward pass
        # The six heads run on a second stream BESIDE the extra-layer chain (forward and backward): conv8_1 .. conv11_2 and
        # pred3 .. pred6 work on 10 x 10 ... 3 x 3 maps -- ~130 launches of 5-25 us that leave most of the chip idle and are
        # bound by launch-to-launch latency, 14 % of the step for 0.5 % of its FLOPs.  Each head only depends on its own
        # feature map, so the head chain (incl. the two large heads pred1 / pred2, which fill the idle CUs) overlaps with the
        # sequential extras.  Config key 'tail_stream' (default on).  The two chains use separate batch-norm workspaces and
        # separate split-K scratch slots (odtk_scratch_slot).
        self._tail = _side_stream(self.dev, 'tail') if (on_gpu and config.get('tail_stream', True) and self.wgrad_stream is None) else None
        # The filter gradients of the extras and heads on a THIRD stream (config key 'tail_wgrad_stream'): in the backward pass of the small-map region every
        # layer is bn_bwd -> wgrad -> dgrad (+ split-K finish) and only the dgrad feeds the next layer, so the ~390 us of filter-gradient launches of that
        # region (conv6 154, pred1 92, pred2 73, conv7 50 ...) leave the latency-bound chain; they join in front of the optimizer.
        self._twg = _side_stream(self.dev, 'tail_wgrad') if (on_gpu and self._tail is not None and config.get('tail_wgrad_stream', True)) else None
        self._cur_slot = 0
        self._g_front = self._g_back = None
        self._g_back_segs = None
        self._eager_steps = 0
        self.dist = None                       # set by attach_data_parallel()
        self.loss_divisor_batch = self.batch_size

        self._define_layers()
        self._init_parameters(config.get('seed', 0))
        self._build_buffers()
        self._warmup_setup(config, data_provider, 'compute_dtype' in config)
        self._load_pretraining_weight()

    # ------------------------------------------------------------------ structure
    def _define_layers(self):
        self.convs = OrderedDict()
        for item in VGG_SEQ:
            if item[0].startswith('conv'):
                self.convs[item[0]] = _Conv(item[0], item[1], item[2], 3, 1, 1, False, True)
        for (n, ci, co, k, s, d) in self.EXTRA_SEQ:
            self.convs[n] = _Conv(n, ci, co, k, s, d, True, True)
        for i, (ch, a) in enumerate(zip(self.FEAT_CH, self.ANCHORS_PER_CELL)):
            self.convs[f'pred{i + 1}'] = _Conv(f'pred{i + 1}', ch, a * self.row, 3, 1, 1, True, False)

    def _cin_pad(self, c):
        return ops.pad_to(c, self.chunk)

    def _init_parameters(self, seed):
        """Flat f32 master buffer in forward order (so gradients complete suffix-first in
        backward: what the bucketed all-reduce relies on)."""
        order = []
        for item in VGG_SEQ:
            if item[0].startswith('conv'):
                order.append(item[0])
            if item[0] == 'conv4_3':
                order.append('l2norm')
        order += [e[0] for e in self.EXTRA_SEQ] + [f'pred{i}' for i in range(1, self.NH + 1)]
        self.layer_order = order
        self.pinfo = OrderedDict()
        off = 0

        def add(name, shape):
            nonlocal off
            n = int(np.prod(shape))
            self.pinfo[name] = (off, tuple(shape))
            off += ops.pad_to(n, 64)

        self.sinfo = OrderedDict()
        soff = 0
        for ln in order:
            if ln == 'l2norm':
                add('l2norm.gamma', (1,))
                continue
            c = self.convs[ln]
            add(ln + '.w', (c.cout, c.k, c.k, self._cin_pad(c.cin)))
            add(ln + '.b', (c.cout,))
            if c.bn:
                add(ln + '.gamma', (c.cout,))
                add(ln + '.beta', (c.cout,))
                self.sinfo[ln + '.mmean'] = (soff, (c.cout,)); soff += ops.pad_to(c.cout, 64)
                self.sinfo[ln + '.mvar'] = (soff, (c.cout,)); soff += ops.pad_to(c.cout, 64)
        self.nparam = off
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.Mom = torch.zeros(off, device=dev)
        self.G = torch.zeros(off, device=dev)
        self.Pc = torch.zeros(off, dtype=self.tdt, device=dev) if self.DT == BF16 else self.P
        self.S = torch.zeros(soff, device=dev)
        self.l2_partial = torch.zeros(ops.sgd_blocks(off), device=dev)
        self.l2_sum = torch.zeros(1, device=dev)
        # synthetic initialisation (no checkpoint available offline): He-normal conv, zero bias,
        # BN gamma 1 / beta 0 / moving (0, 1), L2-norm scale 20 (reference SSD300.py:77)
        g = torch.Generator().manual_seed(seed)
        for ln in order:
            if ln == 'l2norm':
                self.param('l2norm.gamma').fill_(20.0)
                continue
            c = self.convs[ln]
            w = torch.randn(c.cout, c.k, c.k, c.cin, generator=g) * math.sqrt(2.0 / (c.cin * c.k * c.k))
            self.set_param(ln + '.w', w)
            if c.bn:
                self.param(ln + '.gamma').fill_(1.0)
                self.stat(ln + '.mvar').fill_(1.0)
        self._refresh_operand_copies()

    def param(self, name, buf=None):
        off, shape = self.pinfo[name]
        buf = self.P if buf is None else buf
        return buf[off: off + int(np.prod(shape))].view(shape)

    def stat(self, name):
        off, shape = self.sinfo[name]
        return self.S[off: off + int(np.prod(shape))].view(shape)

    def set_param(self, name, value):
        """value: logical shape (conv weights [K,R,S,Cin] un-padded)."""
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
            v = v[..., : self.convs[name[:-2]].cin].contiguous()
        return v

    def load_oracle_params(self, p):
        """Load a dict name -> tensor in the oracle's naming ([K,R,S,Cin] weights)."""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()                                   # weights are loaded: the run does not start from random initialisation
        for k, v in p.items():
            if k in self.pinfo:
                self.set_param(k, v)
            elif k in self.sinfo:
                self.stat(k).copy_(torch.as_tensor(v, dtype=torch.float32).to(self.dev))
        self._refresh_operand_copies()

    def export_params(self):
        self._sync_from_twin()                                     # mid-warm-up: the live weights are the twin's
        out = OrderedDict((k, self.get_param(k)) for k in self.pinfo)
        for k in self.sinfo:
            out[k] = self.stat(k).detach().cpu().clone()
        return out

    def _load_pretraining_weight(self):
        """The reference initialises the 13 VGG convs from slim's vgg_16.ckpt (SSD300.py:31,193-299).
        Accepts the TensorFlow checkpoint itself (tf_checkpoint.NewCheckpointReader, V1 or V2) or a torch/npz file
        holding 'vgg_16/convX/convX_Y/weights' (HWIO) and '.../biases'; otherwise keeps the synthetic He init."""
        path = self.pretraining_weight
        if not path or not (os.path.exists(str(path)) or os.path.exists(str(path) + '.index')):
            return
        try:
            if str(path).endswith('.npz'):
                blob = dict(np.load(path))
            elif str(path).endswith(('.pt', '.pth')):
                blob = torch.load(path, map_location='cpu', weights_only=True)
            else:                                                # slim's vgg_16.ckpt (V1) or a Saver prefix (V2): SSD300.py:31
                from .tf_checkpoint import NewCheckpointReader
                reader = NewCheckpointReader(str(path))
                blob = {k: reader.get_tensor(k) for k in reader.get_variable_to_shape_map() if k.startswith('vgg_16/conv')}
        except Exception as e:                                   # noqa: BLE001
            print(f'[odtk] could not read pretraining weights {path}: {e}; keeping synthetic init')
            return
        for item in VGG_SEQ:
            n = item[0]
            if not n.startswith('conv'):
                continue
            key = f'vgg_16/{n.split("_")[0]}/{n}'
            if key + '/weights' in blob:
                w = torch.as_tensor(np.asarray(blob[key + '/weights']), dtype=torch.float32)   # HWIO
                self.set_param(n + '.w', w.permute(3, 0, 1, 2).contiguous())
                self.set_param(n + '.b', torch.as_tensor(np.asarray(blob[key + '/biases'])))
                if getattr(self, 'f32_warmup_steps', 0):
                    self.cancel_warmup()                           # a pre-trained trunk: the run does not start from random initialisation
        self._refresh_operand_copies()

    # ------------------------------------------------------------------ buffers
    def _build_buffers(self):
        N, dev, dt = self.batch_size, self.dev, self.tdt
        self.images = torch.zeros(N, self.INPUT_SIZE, self.INPUT_SIZE, 3, device=dev)
        self.acts = OrderedDict()
        c0 = self._cin_pad(3)
        self.acts['input'] = _Act(N, self.INPUT_SIZE, self.INPUT_SIZE, c0, c0, dt, dev)
        H = self.INPUT_SIZE
        cur_c = c0
        self.desc = {}
        prev = 'input'
        self.vgg_plan = []
        self.pool_idx = {}                      # 2x2/s2 pools: recorded arg-max (uint16 per 16-byte output chunk)
        self.pool_arg = {}                      # other pools (pool5): recorded arg-max (int32 per 16-byte output chunk)
        for item in VGG_SEQ:
            name = item[0]
            if name.startswith('conv'):
                c = self.convs[name]
                self.desc[name] = ops.conv_desc(N, H, H, cur_c, cur_c, c.cout, c.cout, 3, 1, 1, self.CDT, self.CDT)
                self.acts[name] = _Act(N, H, H, c.cout, c.cout, dt, dev)
                self.vgg_plan.append(('conv', name, prev))
                cur_c = c.cout
            else:
                k, s = item[1], item[2]
                Ho, pt, _ = ops.same_pad(H, k, s)
                self.acts[name] = _Act(N, Ho, Ho, cur_c, cur_c, dt, dev)
                self.vgg_plan.append(('pool', name, prev, k, s, pt))
                if k == 2 and s == 2 and pt == 0 and self.mode == 'train' and bool(self.config.get('pool_index', True)):
                    self.pool_idx[name] = torch.zeros(N * Ho * Ho * (cur_c // ops.chunk(self.DT)), dtype=torch.int16, device=dev)
                elif k <= 3 and self.mode == 'train' and bool(self.config.get('pool_index', True)):
                    # overlapping windows (pool5, 3x3 / stride 1): 4-bit window positions, one int32 per 16-byte output chunk
                    self.pool_arg[name] = torch.zeros(N * Ho * Ho * (cur_c // ops.chunk(self.DT)), dtype=torch.int32, device=dev)
                H = Ho
            prev = name
            if name == 'conv4_3':
                self.acts['feat1'] = _Act(N, H, H, 512, 512, dt, dev)
        # conv -> 2x2 pool pairs that libodtk runs as ONE launch (conv1_2 + pool1: the 64 -> 64 halo kernel pools in its epilogue); config key
        # 'fuse_pool' (default on), 'keep_unpooled' (default off: the un-pooled map is not even stored -- tests that inspect it turn it on)
        self.fused_pool = {}
        self.keep_unpooled = bool(self.config.get('keep_unpooled', False))
        if self.mode == 'train' and bool(self.config.get('fuse_pool', True)):
            for step in self.vgg_plan:
                if step[0] == 'pool' and step[1] in self.pool_idx and step[2] != 'conv4_3' and step[2] in self.desc and \
                        ops.conv2d_fwd_pool2x2_fused(self.desc[step[2]]):
                    # (round 4: conv2_2 + pool2 and conv3_3 + pool3 in the raster-run halo kernel; config key 'fuse_pool_halo' = False keeps only conv1_2 + pool1, A/B)
                    if self.convs[step[2]].cin != 64 and not bool(self.config.get('fuse_pool_halo', True)):
                        continue
                    self.fused_pool[step[2]] = step[1]
        self.zbuf, self.bnsave = {}, {}
        max_ws = 0
        for (name, ci, co, k, s, d) in self.EXTRA_SEQ:
            src = self.acts[prev]
            self.desc[name] = ops.conv_desc(N, src.H, src.W, ci, src.ld, co, co, k, s, d, self.CDT, self.CDT)
            Ho = self.desc[name].Ho
            self.zbuf[name] = _Act(N, Ho, Ho, co, co, dt, dev)
            self.acts[name] = _Act(N, Ho, Ho, co, co, dt, dev)
            self.bnsave[name] = (torch.zeros(co, device=dev), torch.zeros(co, device=dev))
            max_ws = max(max_ws, ops.bn_workspace_bytes(N * Ho * Ho, co))
            self.extra_src = getattr(self, 'extra_src', {})
            self.extra_src[name] = prev
            prev = name
        # heads write straight into pred [N, 8828, 25] (reference SSD300.py:316-321 reshape/concat)
        self.pred = torch.zeros(N, self.NUM_PRIORS, self.row, device=dev)
        self.dpred = torch.zeros_like(self.pred)
        self.head_off = []
        off = 0
        for i, src_name in enumerate(self.FEAT_SRC):
            name = f'pred{i + 1}'
            src = self.acts[src_name]
            c = self.convs[name]
            kp = ops.pad_to(c.cout, 8)
            self.desc[name] = ops.conv_desc(N, src.H, src.W, src.C, src.ld, c.cout, kp, 3, 1, 1, self.CDT, self.CDT)
            self.zbuf[name] = _Act(N, src.H, src.W, c.cout, kp, dt, dev)
            self.bnsave[name] = (torch.zeros(c.cout, device=dev), torch.zeros(c.cout, device=dev))
            max_ws = max(max_ws, ops.bn_workspace_bytes(src.M, c.cout))
            self.head_off.append(off)
            off += src.H * src.W * self.ANCHORS_PER_CELL[i]
        assert off == self.NUM_PRIORS
        for a in self.acts.values():
            max_ws = max(max_ws, ops.bn_workspace_bytes(a.M, a.C))
        self.ws = torch.zeros(max_ws, dtype=torch.uint8, device=dev)
        self.ws_tail = torch.zeros(max_ws, dtype=torch.uint8, device=dev) if self._tail is not None else None
        # dgrad-layout filters
        self.wt = {}
        for name, c in self.convs.items():
            if name == 'conv1_1':
                continue
            d = self.desc[name]
            kp = ops.pad_to(c.cout, 8) if name.startswith('pred') else c.cout
            self.wt[name] = torch.zeros(d.C * c.k * c.k * kp, dtype=dt, device=dev)
        # ReLU masks as sign bits (round 4): conv1_2's input-gradient pass reads one byte per 16-byte chunk of conv1_1's activation instead of the chunk
        # (369 MB at batch 32); written by conv1_1's forward kernel.  Config key 'relu_bits' (default on), where the kernel pair supports it.
        self.relu_bits = {}
        if self.mode == 'train' and bool(self.config.get('relu_bits', True)) and 'conv1_1' in self.desc and 'conv1_2' in self.desc and \
                ops.conv2d_relu_bits_supported(self.desc['conv1_1'], self.desc['conv1_2'], self.acts['conv1_2'].ld):
            a1 = self.acts['conv1_1']
            self.relu_bits['conv1_1'] = torch.zeros(a1.M * (a1.ld // 8), dtype=torch.uint8, device=dev)
        # box side
        fs, nas, hw = prior_spec(self.INPUT_SIZE, self.prior_scales(), self.ASPECTS, self.FEATURE_SIZES, self.ANCHORS_PER_CELL)
        self.pri = ops.ssd_priors(self.INPUT_SIZE, fs, nas, hw, dev)       # y1x1, y2x2, yx, hw, nmsbox
        A = self.NUM_PRIORS
        self.gt = None
        i32 = dict(dtype=torch.int32, device=dev)
        self.m_ngt = torch.zeros(N, **i32)
        self.m_status = torch.zeros(N, A, dtype=torch.uint8, device=dev)
        self.m_rg = torch.zeros(N, A, **i32)
        self.m_counts = torch.zeros(N, 4, **i32)
        self.negloss = torch.zeros(N, A, device=dev)
        self.sel_idx = torch.zeros(N, A, **i32)
        self.sel_cnt = torch.zeros(N, **i32)
        self.loss_parts = torch.zeros(N, 4, device=dev)
        self.data_loss = torch.zeros(1, device=dev)
        self.loss_ring = torch.zeros(8, device=dev)
        nc = self.num_classes - 1
        self.d_conf = torch.zeros(A, nc, device=dev)
        self.d_boxes = torch.zeros(A, 4, device=dev)
        self.d_keep = torch.zeros(A, dtype=torch.uint8, device=dev)
        self.d_cand = torch.zeros(A, nc, dtype=torch.uint8, device=dev)
        self.refresh_wt()

    def _wslice(self, name, buf):
        off, shape = self.pinfo[name]
        return buf[off: off + int(np.prod(shape))]

    def _refresh_operand_copies(self):
        if self.DT == BF16:
            ops.cast_from_f32(self.P, self.Pc)
        if hasattr(self, 'wt'):
            self.refresh_wt()

    def refresh_wt(self):
        """Flipped/transposed dgrad filters from the f32 master (after every optimizer step): one batched launch."""
        if getattr(self, '_wt_pending', False) and self._side is not None and torch.cuda.current_stream() != self._side:
            # a refresh of the last training step may still be queued on the side stream (_finish_step): whoever refreshes again from another
            # stream -- load_weight, a checkpoint restore -- goes behind it, so the LAST refresh is the one of the current parameters
            torch.cuda.current_stream().wait_stream(self._side)
            self._wt_pending = False
        if getattr(self, '_fp_batch', None) is None:
            entries = []
            for name, wt in self.wt.items():
                c = self.convs[name]
                d = self.desc[name]
                kp = ops.pad_to(c.cout, 8) if name.startswith('pred') else c.cout
                entries.append((self._wslice(name + '.w', self.P), wt, c.cout, c.k, c.k, d.C, kp))
            self._fp_batch = ops.FilterPrepareBatch(entries, self.DT, self.dev)
        self._fp_batch.run()

    # ------------------------------------------------------------------ batch norm: local, or over the global batch (sync_bn)
    def _bn_fwd(self, z, M, C_, ldz, gamma, beta, mmean, mvar, sm, si, training, relu, y, ldy, rows_per_img, y_img_stride, ws):
        if training and self.sync_bn is not None:
            self.sync_bn.fwd(z, M, C_, ldz, gamma, beta, mmean, mvar, sm, si, relu, y, ldy, rows_per_img, y_img_stride, ws)
        else:
            ops.bn_fwd(z, M, C_, ldz, gamma, beta, mmean, mvar, sm, si, training, relu, y, ldy, rows_per_img, y_img_stride, ws)

    def _bn_bwd(self, z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, sm, si, relu, dz, dgamma, dbeta, ws):
        if self.sync_bn is not None:
            self.sync_bn.bwd(z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, sm, si, relu, dz, dgamma, dbeta, ws)
        else:
            ops.bn_bwd(z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, sm, si, relu, dz, dgamma, dbeta, ws)

    # ------------------------------------------------------------------ forward
    def _py(self, fn):
        """a torch-level action inside the step (stream fork / join, event, fill): run it, and keep it in the launch list when one is being recorded"""
        fn()
        rec = _lib.recording()
        if rec is not None:
            def replay(fn=fn):
                fn()                                             # (whatever fn returns is not a status code)
            rec.append((replay, ()))

    _phases = None           # tools/phase_times.py: list of (name, event) of the current step -- events on the MAIN stream at the phase boundaries of the step

    def _phase(self, name):
        if self._phases is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self._phases.append((name, ev))

    def _event(self, key):
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        return ev

    def _conv_fwd(self, name, src, dst, bias, relu):
        ops.conv2d_fwd(self.desc[name], src.t, self._wslice(name + '.w', self.Pc), bias, dst.t, relu)

    # ---- train_one_epoch: the NEXT batch's pixels cross PCIe under the running step (round 6) ----------------------------------------------------------
    # The f32 pixels of a batch (34.6 MB at batch 32: 0.64 ms at PCIe rate, 8 % of a step) are read exactly once, by the preprocess launch that opens the step.
    # train_one_epoch therefore hands the batch AFTER the current one to the step (`_prefetch_next`): once the forward pass is enqueued, the copy into
    # `self.images` is queued on the side stream behind an event recorded after the preprocess launch, and the next step's preprocess waits for the copy's
    # event.  The ground truth (38 KB) still travels at the start of its own step (matching and the loss read it all step long).  Eager launches only.
    _prefetch_next = None        # host tensor of the next batch's pixels, consumed by _step_front
    _img_ready = None            # event on the side stream: the prefetched pixels are in self.images
    _img_prefetched = None       # the host object whose pixels self.images holds (set_batch skips the copy for it)

    def _prefetch_issue(self, side):
        images, self._prefetch_next = self._prefetch_next, None
        if images is None or side is None or _lib.recording() is not None:
            return
        consumed = self._event('img_consumed')
        side.wait_event(consumed)
        with torch.cuda.stream(side):
            self.images.copy_(images[1], non_blocking=True)
            ev = self._event('img_ready')
            ev.record(side)
        self._img_ready, self._img_prefetched = ev, images[0]

    def _forward(self, training, subtract_mean=True):
        a = self.acts
        if self._img_ready is not None:                       # (eager launches only: never set while a graph is captured or a launch list recorded)
            torch.cuda.current_stream().wait_event(self._img_ready)
            self._img_ready = None
        ops.preprocess(self.images, MEAN_RGB if subtract_mean else (0., 0., 0.), a['input'].ld, self.DT,
                       a['input'].t)
        if self._prefetch_next is not None:
            self._event('img_consumed').record(torch.cuda.current_stream())
        for step in self.vgg_plan:
            if step[0] == 'conv':
                _, name, prev = step
                if training and name in self.fused_pool:
                    # conv + bias + ReLU + the 2x2 pool behind it in one launch; the un-pooled map is not stored (nothing else reads it: the
                    # backward pass routes by the recorded arg-max and masks by the sign of the pooled value)
                    pname = self.fused_pool[name]
                    ops.conv2d_fwd_pool2x2(self.desc[name], a[prev].t, self._wslice(name + '.w', self.Pc), self.param(name + '.b'),
                                           a[name].t if self.keep_unpooled else None, True, a[pname].t, self.pool_idx[pname])
                    continue
                if training and name in self.relu_bits:
                    ops.conv2d_fwd_bits(self.desc[name], a[prev].t, self._wslice(name + '.w', self.Pc), self.param(name + '.b'), a[name].t, True,
                                        self.relu_bits[name])
                    continue
                self._conv_fwd(name, a[prev], a[name], self.param(name + '.b'), True)
            else:
                _, name, prev, k, s, pt = step
                x, y = a[prev], a[name]
                if training and prev in self.fused_pool:
                    continue
                if training and name in self.pool_idx:
                    ops.maxpool2x2_fwd_idx(x.t, y.t, self.pool_idx[name], x.N, x.H, x.W, x.C, x.ld, y.H, y.W)
                elif training and name in self.pool_arg:
                    ops.maxpool_fwd_argmax(x.t, y.t, self.pool_arg[name], x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
                else:
                    ops.maxpool_fwd(x.t, y.t, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
        c43 = a['conv4_3']
        self._phase('fwd: conv1_1 .. conv4_3 done')
        ops.l2norm_fwd(c43.t, a['feat1'].t, c43.M, 512, c43.ld, self.param('l2norm.gamma'))
        tail = self._tail if self.sync_bn is None else None
        main = torch.cuda.current_stream() if tail is not None else None
        if tail is not None:
            self._py(lambda: tail.wait_stream(main))      # fork: feat1 is final
            with self._on_tail():
                self._head_fwd(0, training)
        for (name, ci, co, k, s, d) in self.EXTRA_SEQ:
            src = a[self.extra_src[name]]
            z, y = self.zbuf[name], a[name]
            if name == 'conv8_1':
                self._phase('fwd: conv6, conv7 done')
            self._conv_fwd(name, src, z, self.param(name + '.b'), False)
            self._phase(f'  fwd {name}: conv done')
            sm, si = self.bnsave[name]
            self._bn_fwd(z.t, z.M, co, z.ld, self.param(name + '.gamma'), self.param(name + '.beta'),
                       self.stat(name + '.mmean'), self.stat(name + '.mvar'), sm, si, training, True,
                       y.t, y.ld, z.M, 0, self.ws)
            self._phase(f'  fwd {name}: batch norm done')
            if tail is not None and name in self.FEAT_SRC:
                self._py(lambda: tail.wait_stream(main))  # this feature map is final: its head may start
                with self._on_tail():
                    self._head_fwd(self.FEAT_SRC.index(name), training)
        self._phase('fwd: conv8_1 .. conv11_2 done (main chain)')
        if tail is not None:
            self._py(lambda: main.wait_stream(tail))      # join: pred is complete
        else:
            for i in range(self.NH):
                self._head_fwd(i, training)
        self._phase('fwd: heads joined')

    def _head_fwd(self, i, training):
        """pred<i+1>: 3x3 conv + bias + batch norm, written straight into pred [N, 8828, 25] (SSD300.py:85-90, :316-321)"""
        a = self.acts
        A25 = self.NUM_PRIORS * self.row
        name = f'pred{i + 1}'
        src, z = a[self.FEAT_SRC[i]], self.zbuf[name]
        self._conv_fwd(name, src, z, self.param(name + '.b'), False)
        sm, si = self.bnsave[name]
        co = self.convs[name].cout
        out = self.pred.view(-1)[self.head_off[i] * self.row:]
        self._bn_fwd(z.t, z.M, co, z.ld, self.param(name + '.gamma'), self.param(name + '.beta'),
                   self.stat(name + '.mmean'), self.stat(name + '.mvar'), sm, si, training, False,
                   out, co, src.H * src.W, A25, self.ws)

    @contextlib.contextmanager
    def _on_tail(self):
        """launches inside go to the head stream, with its own batch-norm workspace and split-K scratch slot"""
        ops.scratch_slot(1)
        self._cur_slot = 1
        ws, self.ws = self.ws, self.ws_tail
        try:
            with torch.cuda.stream(self._tail):
                yield
        finally:
            self.ws = ws
            self._cur_slot = 0
            ops.scratch_slot(0)

    # ------------------------------------------------------------------ loss
    def _match(self):
        """Prior <-> ground-truth matching (SSD300.py:347-426): needs the boxes only, not the network output."""
        pri = self.pri
        if self.m_best is None or self.m_best.shape[1] != self.gt.shape[1]:
            self.m_best = torch.zeros(self.batch_size, self.gt.shape[1], dtype=torch.int32, device=self.dev)
        ops.ssd_match(pri[0], pri[1], pri[3], self.gt, self.m_ngt, self.m_best, self.m_status, self.m_rg, self.m_counts)

    def _loss(self, grad_scale, matched=False):
        N, A = self.batch_size, self.NUM_PRIORS
        pri = self.pri
        if not matched:
            self._match()
        ops.softmax_ce_const(self.pred, N * A, self.num_classes, self.row, self.num_classes - 1, self.negloss)
        ops.nms_batched(pri[4], 0, self.negloss, A, 1, self.m_status, A, 1, 2, A, N, self.m_counts[:, 2:], 4, 0,
                        0.7, self.sel_idx, A, self.sel_cnt)
        ops.ssd_loss(self.pred, self.num_classes, pri[2], pri[3], self.gt, self.m_ngt, self.m_best, self.m_status,
                     self.m_rg, self.m_counts, self.negloss, self.sel_idx, self.sel_cnt, grad_scale,
                     self.loss_parts, self.dpred)
        # (the reported scalar -- sum of loss_parts[:, 3] and the L2 term -- is one launch behind the optimizer: _finish_step)

    m_best = None

    # ------------------------------------------------------------------ backward
    def _grad(self, name):
        return self._wslice(name, self.G)

    def _conv_bwd_params(self, name, x, dy_t, lddy):
        """Filter gradient (+ fused bias gradient).  A bias that feeds BatchNorm has an exactly
        zero gradient (BN subtracts the batch mean; TF only produces round-off noise there), so
        it is left at zero and only sees weight decay."""
        d = self.desc[name]
        dbias = None if self.convs[name].bn else self._grad(name + '.b')
        # data parallel: the SAME streams as the single-device step (round 4) -- a bucket's all-reduce is launched from a stream that waits for every
        # stream carrying gradient kernels (_comm_launch); only the per-bucket graph replay (use_graph with dist) still runs one stream
        multi = self._dp_multi_stream()
        side = self.wgrad_stream if multi else None
        if side is None and self._twg is not None and multi and self.sync_bn is None and self.convs[name].bn:
            side = self._twg                                          # extras and heads (the batch-normalised layers): off the latency-bound chain
        if side is None:
            ops.conv2d_wgrad(d, x.t, dy_t, lddy, self._grad(name + '.w'), dbias)
            return
        cur = torch.cuda.current_stream()
        self._py(lambda: side.wait_stream(cur))                       # dy(L) is complete at this point of the launching stream
        ops.scratch_slot(2)                                           # the split partials of this launch: not the launching stream's scratch
        try:
            with torch.cuda.stream(side):
                ops.conv2d_wgrad(d, x.t, dy_t, lddy, self._grad(name + '.w'), dbias)
        finally:
            ops.scratch_slot(self._cur_slot)

    def _backward(self):
        for name in self._backward_iter():
            self._py(lambda name=name: self._mark_ready(name))

    def _backward_iter(self):
        """Backward pass as a generator: it hands back a layer name as soon as every gradient of that layer (and of all
        later layers) has been launched -- the data-parallel hooks and the segmented graph capture hang on these points."""
        a = self.acts
        if getattr(self, '_wt_pending', False):           # (the front's join normally covers this; a backward pass driven by hand does not)
            self._py(lambda: torch.cuda.current_stream().wait_stream(self._side))
            self._wt_pending = False
        # data parallel (round 4): the head stream stays on.  Layer names are handed back in the order their launches were ENQUEUED (pred6 .. pred1 on the head
        # stream, then conv11_2 .. conv6 on the main stream -- still suffix-first in the flat gradient buffer); a bucket that closes is all-reduced from a launch
        # stream that waits for the main, head and filter-gradient streams (_comm_launch), so readiness is an event on the stream that ran the kernels, not a
        # position in the main chain.  Per-bucket graph replay (use_graph with dist) keeps the single-stream backward of round 3.
        tail = self._tail if (self.sync_bn is None and self.wgrad_stream is None and self._dp_multi_stream()) else None
        evs = {}
        if tail is None:
            # heads (pred6 .. pred1): dpred -> BN bwd -> wgrad / dgrad into the feature map
            for i in reversed(range(self.NH)):
                self._head_bwd(i)
                yield f'pred{i + 1}'
        else:
            # ... on the head stream, beside the extras' chain; an event per head tells the chain when a feature map's
            # gradient holds the head's contribution
            main = torch.cuda.current_stream()
            self._py(lambda: tail.wait_stream(main))      # fork: d(pred) is final
            for i in reversed(range(self.NH)):
                with self._on_tail():
                    self._head_bwd(i)
                    ev = evs[self.FEAT_SRC[i]] = self._event(('head', i))
                    self._py(lambda ev=ev: ev.record(tail))
                yield f'pred{i + 1}'                      # (outside the stream context: whoever resumes us runs on the main stream)
        # extra layers conv11_2 .. conv6
        self._phase('loss done')
        for (name, ci, co, k, s, d) in reversed(self.EXTRA_SEQ):
            src = a[self.extra_src[name]]
            z, y = self.zbuf[name], a[name]
            sm, si = self.bnsave[name]
            if name == 'conv7':
                self._phase('bwd: conv11_2 .. conv8_1 done (main chain)')
            if name == self.EXTRA_SEQ[-1][0] and name in evs:
                self._py(lambda ev=evs[name]: main.wait_event(ev))      # the last feature map: only its head wrote y.g
            self._bn_bwd(z.t, y.t, y.g, z.M, co, z.ld, y.ld, z.M, 0, self.param(name + '.gamma'), sm, si, True,
                       z.g, self._grad(name + '.gamma'), self._grad(name + '.beta'), self.ws)
            self._phase(f'  bwd {name}: batch norm done')
            self._conv_bwd_params(name, src, z.g, z.ld)
            self._dp_grad_point(name)
            # the source already holds the head's gradient when it is a feature map
            acc = self.extra_src[name] in self.FEAT_SRC
            if acc and self.extra_src[name] in evs:
                self._py(lambda ev=evs[self.extra_src[name]]: main.wait_event(ev))
            relu_src = src.t if name == 'conv6' else None       # pool5 output: post-ReLU values
            ops.conv2d_dgrad(self.desc[name], z.g, z.ld, self.wt[name], relu_src, src.g, acc)
            self._phase(f'  bwd {name}: input gradient done')
            yield name
        self._phase('bwd: conv7, conv6 done (main chain)')
        if tail is not None:
            self._py(lambda: main.wait_stream(tail))      # join: pred1 -> feat1.g is final before the trunk reads it
        self._phase('bwd: heads joined')
        # VGG trunk
        for step in reversed(self.vgg_plan):
            if step[0] == 'pool':
                _, name, prev, k, s, pt = step
                x, y = a[prev], a[name]
                if name in self.pool_idx:
                    ops.maxpool2x2_bwd_idx(self.pool_idx[name], y.g, x.g, x.N, x.H, x.W, x.C, x.ld, y.H, y.W)
                elif name in self.pool_arg:
                    ops.maxpool_bwd_argmax(self.pool_arg[name], y.g, x.g, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
                else:
                    ops.maxpool_bwd(x.t, y.t, y.g, x.g, x.N, x.H, x.W, x.C, x.ld, y.H, y.W, k, s, pt, pt)
                if prev == 'conv4_3':       # second consumer: L2-norm -> pred1 (accumulate, ReLU mask)
                    f1 = a['feat1']
                    ops.l2norm_bwd(x.t, f1.g, x.g, x.M, 512, x.ld, self.param('l2norm.gamma'),
                                   self._grad('l2norm.gamma'), True, x.t)
                    yield 'l2norm'
            else:
                _, name, prev = step
                x, y = a[prev], a[name]
                self._conv_bwd_params(name, x, y.g, y.ld)
                self._dp_grad_point(name)
                if name != 'conv1_1' and prev in self.relu_bits:
                    ops.conv2d_dgrad_bits(self.desc[name], y.g, y.ld, self.wt[name], self.relu_bits[prev], x.g, False)
                elif name != 'conv1_1':
                    ops.conv2d_dgrad(self.desc[name], y.g, y.ld, self.wt[name], x.t, x.g, False)
                if name in ('conv5_1', 'conv4_1', 'conv3_1', 'conv2_1', 'conv1_1'):
                    self._phase(f'bwd: {name[:5]} block done')
                yield name
        if self._twg is not None and self._dp_multi_stream() and self.sync_bn is None:
            cur = torch.cuda.current_stream()
            self._py(lambda: cur.wait_stream(self._twg))                   # the tail's filter gradients join before the optimizer
        if self.wgrad_stream is not None and self._dp_multi_stream():
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)     # join before the optimizer

    def _head_bwd(self, i):
        a = self.acts
        A25 = self.NUM_PRIORS * self.row
        name = f'pred{i + 1}'
        src, z = a[self.FEAT_SRC[i]], self.zbuf[name]
        co = self.convs[name].cout
        sm, si = self.bnsave[name]
        dyv = self.dpred.view(-1)[self.head_off[i] * self.row:]
        self._bn_bwd(z.t, None, dyv, z.M, co, z.ld, co, src.H * src.W, A25, self.param(name + '.gamma'), sm, si,
                   False, z.g, self._grad(name + '.gamma'), self._grad(name + '.beta'), self.ws)
        self._conv_bwd_params(name, src, z.g, z.ld)
        ops.conv2d_dgrad(self.desc[name], z.g, z.ld, self.wt[name], None, src.g, False)

    def _mark_ready(self, layer_name):
        if self.dist is not None:
            self.dist.layer_ready(layer_name)
        self._dp_point = None                             # a point belongs to the layer it was recorded for

    def _dp_multi_stream(self):
        """True when the backward pass may use the head / filter-gradient streams: always on one device; under data parallel with eager launches
        (the default) too -- only the per-bucket graph replay keeps the single-stream backward (a captured segment cannot end with a forked stream)"""
        return self.dist is None or not self.use_graph

    _dp_point = None         # event on the main stream behind the last GRADIENT launch of the layer being handed to the data-parallel hooks (None: none recorded)

    def _dp_grad_point(self, name):
        """Data parallel with the step's streams on (round-5 review, item 8): record, on the launching stream, the point behind the last gradient-producing launch
        of layer `name` -- called in front of the layer's INPUT-gradient launch, which no bucket waits for.  _comm_launch makes the collective's stream wait for
        this event instead of `wait_stream(main)`, i.e. instead of everything the main chain holds when the bucket closes (the boundary layer's input gradient,
        100-250 us on the trunk).  One event per layer, created once."""
        if self.dist is None or not self._dp_multi_stream() or (self._tail is None and self._twg is None and self.wgrad_stream is None) or not self.config.get('dp_layer_events', True):
            return
        # (only the layers whose readiness CLOSES a bucket: an event record costs the main chain ~2.5 us -- 25 of them per step measured +0.06 ms in a world of one
        #  rank, gpurun r6y; 4-5 do not)
        bl = self.__dict__.get('_dp_boundary')
        if bl is None or bl[0] is not self.dist:
            bl = self._dp_boundary = (self.dist, frozenset(self.dist.boundary_layers()))
        if name not in bl[1]:
            return
        evs = self.__dict__.setdefault('_dp_point_events', {})
        ev = evs.get(name)
        if ev is None:
            ev = evs[name] = torch.cuda.Event()

        def rec(ev=ev):
            ev.record(torch.cuda.current_stream())
            self._dp_point = ev
        self._py(rec)

    @contextlib.contextmanager
    def _comm_launch(self):
        """Context under which dist.py launches a bucket's collective.  torch.distributed orders a collective behind the CURRENT stream at the call; the
        gradient kernels of the bucket's layers may sit on the main, head, tail-filter-gradient or filter-gradient stream.  The call is therefore made from
        the tail-filter-gradient stream after it has been told to wait for the others at this point: its own work (the extras' / heads' filter gradients)
        is what the bucket is waiting for anyway, nothing on it is latency-critical (it joins in front of the optimizer), and the main chain itself waits
        for nothing.  No extra stream: the HIP runtime deals streams onto four hardware queues and a fifth would alias the main stream's.
        (Round 3 dropped the side streams under data parallel instead: +5.3 % / +1.1 % on the step before a byte crossed xGMI.)"""
        if (self._tail is None and self._twg is None and self.wgrad_stream is None) or not self._dp_multi_stream():
            yield                                         # one stream carries everything: the current stream is the right one
            return
        main = torch.cuda.current_stream()
        c = self._twg if self._twg is not None else (self.wgrad_stream if self.wgrad_stream is not None else main)
        point, self._dp_point = self._dp_point, None
        for st in (main, self._tail, self._twg, self.wgrad_stream):
            if st is not None and st != c:
                if st is main and point is not None:
                    c.wait_event(point)                   # the main chain up to the closing layer's last gradient launch, not its input gradient behind it
                else:
                    c.wait_stream(st)
        with torch.cuda.stream(c):
            yield

    # ------------------------------------------------------------------ public: training
    def _set_batch_engine(self, images, ground_truth):
        images_in = images
        images = torch.as_tensor(images, dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert tuple(images.shape) == (self.batch_size, self.INPUT_SIZE, self.INPUT_SIZE, 3), images.shape
        if self._img_prefetched is not None and self._img_prefetched is images_in:
            self._img_prefetched = None                       # train_one_epoch: these pixels crossed under the previous step (_prefetch_issue)
            if self._img_ready is not None:
                self._img_ready.synchronize()                 # (long done; strict: the iterator may refill its host buffer in place from here on)
        else:
            if self._img_ready is not None:                   # a prefetched batch that is not the one being set: the new copy goes behind it
                torch.cuda.current_stream().wait_event(self._img_ready)
                self._img_ready = self._img_prefetched = None
            self.images.copy_(images, non_blocking=True)
        gt = torch.as_tensor(ground_truth, dtype=torch.float32)
        if self.gt is None or self.gt.shape != gt.shape:
            self.gt = torch.zeros(gt.shape, device=self.dev)
            self._graphs_invalidate()                 # captured launches hold the old pointer / pad length
        self.gt.copy_(gt, non_blocking=True)

    def _step_front(self):
        if self.m_best is None or self.m_best.shape[1] != self.gt.shape[1]:
            self.m_best = torch.zeros(self.batch_size, self.gt.shape[1], dtype=torch.int32, device=self.dev)
        side = self._side if self.config.get('side_front', True) else None
        if side is None:
            ops.zero(self.G)
            self._forward(True)
            self._loss(1.0 / self.loss_divisor_batch)
            return
        # Two pieces of the step that do not depend on the network run on a side stream UNDER the forward pass instead of in front of / behind it (round 3):
        # the flat gradient buffer is cleared (105 MB; the first filter gradient is a whole forward pass away) and the priors are matched to the ground truth
        # (SSD300.py:347-426 needs the boxes only).  The join is in front of the loss.
        main = torch.cuda.current_stream()
        self._py(lambda: side.wait_stream(main))          # behind the previous step's optimizer (it read G) and set_batch's copies
        with torch.cuda.stream(side):
            ops.zero(self.G)
            self._match()
        # One join instead of two in front of the loss: every cross-queue wait that the main queue actually has to honour costs it ~12-25 us (trace: 37 us
        # between the last head kernel and the first loss kernel with two joins, 12 with one).  The head stream waits for the side stream at its fork --
        # the side's work ended two milliseconds earlier -- and the forward pass's join with the head stream covers both.
        via_tail = self._tail is not None and self.sync_bn is None and self.config.get('front_join_via_tail', True)
        if via_tail:
            self._py(lambda: self._tail.wait_stream(side))
        self._forward(True)
        self._prefetch_issue(side)                            # (after the forward pass is enqueued: a pageable source blocks the host for the copy's duration)
        if not via_tail:
            self._py(lambda: main.wait_stream(side))
        self._wt_pending = False
        self._loss(1.0 / self.loss_divisor_batch, matched=True)

    def _graphs_invalidate(self):
        self._cmds = None
        self._g_front = self._g_back = None
        self._g_back_segs = None
        self._eager_steps = 0

    def _graphs_build_safe(self):
        """Capture; on any capture error fall back to eager launches for the rest of the run (loudly)."""
        try:
            self._graphs_build()
            return True
        except Exception as e:                                     # noqa: BLE001
            import warnings
            warnings.warn(f'odtk: HIP-graph capture failed ({type(e).__name__}: {e}); continuing with eager launches')
            self._graphs_invalidate()
            self.use_graph = False
            torch.cuda.synchronize()
            return False

    def _graphs_build(self):
        """Capture the launch sequence of a step into HIP graphs (about 210 kernel launches per step; the host
        cannot issue the ~100 short box/BN/small-conv launches as fast as the GPU retires them).  Forward+loss is
        one graph; backward is one graph, or with data parallelism one graph per gradient bucket (the all-reduces are
        launched eagerly between the replays).  The optimizer stays eager: `lr` changes per call."""
        torch.cuda.synchronize()
        # with a process group attached RCCL's watchdog thread keeps calling the HIP runtime; only THIS thread's calls
        # have to be capture-safe
        self._capture_mode = 'thread_local' if self.dist is not None else 'global'
        self._g_front = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_front, capture_error_mode=self._capture_mode):
            self._step_front()
        self._g_back = None
        self._g_back_segs = None
        if self.dist is None:
            self._g_back = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_back, capture_error_mode=self._capture_mode):
                self._backward()
        else:
            # data parallel: one graph per gradient bucket; the bucket's all-reduce is launched (eagerly, on RCCL's
            # stream) between two replays, exactly where the eager backward would have launched it
            boundaries = set(self.dist.boundary_layers())
            it = self._backward_iter()
            segs, done = [], False
            while not done:
                g, names = torch.cuda.CUDAGraph(), []
                with torch.cuda.graph(g, capture_error_mode=self._capture_mode):
                    while True:
                        try:
                            n = next(it)
                        except StopIteration:
                            done = True
                            break
                        names.append(n)
                        if n in boundaries:
                            break
                if names:
                    segs.append((g, names))
            self._g_back_segs = segs

    AUTO_STEPS = 5

    @property
    def launch_mode_pending(self):
        """True while use_graph='auto' has not decided yet (bench.py keeps warming up until it has)"""
        return self._auto is not None and self.use_graph

    def _auto_step(self, lr):
        """One calibration step of use_graph='auto': steps AUTO_STEPS.. replay the graphs, the last AUTO_STEPS launch eagerly."""
        au = self._auto
        mode = 'graph' if au['left'] > self.AUTO_STEPS else 'eager'
        torch.cuda.synchronize()
        t0 = __import__('time').perf_counter()
        saved = self.use_graph
        self.use_graph = mode == 'graph'
        self._auto = None
        try:
            loss = self._train_step_engine(lr)
        finally:
            self._auto, self.use_graph = au, saved
        torch.cuda.synchronize()
        dt = __import__('time').perf_counter() - t0
        if self.dist is not None and self.dist.world > 1:
            # every rank must pick the SAME launch mode (the bucket graphs and the eager backward issue their collectives at different points
            # of the host timeline): decide on the slowest rank's timing
            import torch.distributed as tdist
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX, group=self.dist.red.group)
            dt = float(t.item())
        au['t'].setdefault(mode, []).append(dt)
        au['left'] -= 1
        if au['left'] == 0:
            best = {m: min(v[1:]) for m, v in au['t'].items()}        # first step of a mode: one-off costs
            self.use_graph = best['graph'] <= best['eager']
            self.launch_mode = {'picked': 'graph' if self.use_graph else 'eager', 'ms': {m: round(v * 1e3, 3) for m, v in best.items()}}
            self._auto = None
            if self.verbose:
                print('[odtk] launch mode:', self.launch_mode)
        return loss

    launch_mode = None

    def _train_step_engine(self, lr):
        """One optimizer step on the batch loaded by set_batch(); returns the loss (data + L2)
        as a 1-element device tensor without synchronising."""
        use_graph = self.use_graph and self._eager_steps >= 2
        if use_graph and self._g_front is None:
            use_graph = self._graphs_build_safe()
        if use_graph and self._auto is not None:
            return self._auto_step(lr)
        if self.dist is not None:
            self.dist.begin_step()
        if self.use_list and self._eager_steps >= 2 and self.sync_bn is None:
            if self._cmds is None:                               # third step: run it through the Python wrappers once more, recording
                _lib.record_begin()
                try:
                    self._step_front()
                    self._backward()
                finally:
                    cmds = _lib.record_end()
                self._cmds = cmds
            else:
                for f, a in self._cmds:
                    rc = f(*a)
                    if rc:
                        _lib.check(rc)
            return self._finish_step(lr)
        if use_graph:
            if getattr(self, '_wt_pending', False):
                # the last EAGER step (use_graph='auto' calibrates with eager ones) queued the refresh of the dgrad filter copies on the side stream;
                # nothing inside a captured graph orders its nodes behind work outside it: join here, once
                torch.cuda.current_stream().wait_stream(self._side)
                self._wt_pending = False
            self._g_front.replay()
        else:
            self._step_front()
            self._eager_steps += 1
        if use_graph and self._g_back is not None:
            self._g_back.replay()
        elif use_graph and self._g_back_segs is not None:
            for g, names in self._g_back_segs:
                g.replay()
                for n in names:
                    self.dist.layer_ready(n)
        else:
            self._backward()
        return self._finish_step(lr)

    def _finish_step(self, lr):
        if self.dist is not None:
            self.dist.finish_step()
        ops.sgd_momentum(self.P, self.Mom, self.G, lr, 0.9, self.weight_decay, 1.0, self.l2_partial,
                         self.Pc if self.DT == BF16 else None)
        # reference SSD300.py:148-152: sum_i loss_i / batch + wd * sum_v ||v||^2 / 2 (pre-update weights) -- one launch; the result goes to a ring of
        # eight scalars, so a loss tensor the caller has not read yet survives the next seven steps
        out = self.loss_ring[self.global_step % 8:self.global_step % 8 + 1]
        ops.loss_total(self.loss_parts[:, 3], self.batch_size, 4, self.l2_partial, 1.0 / self.batch_size, self.weight_decay,
                       self.data_loss, self.l2_sum, out)
        side = self._side if (self.config.get('side_front', True) and self.use_graph in (False, 'list')) else None
        if side is None:
            self.refresh_wt()
        else:
            # the dgrad-layout filter copies are first read a whole forward pass into the next step: refreshed on the side stream, which the next step's
            # front joins before its loss (a graph replay has no such join with work outside the graph: there the refresh stays on the main stream)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.refresh_wt()
            self._wt_pending = True
        self.global_step += 1
        return out

    def train_one_epoch(self, lr):
        if callable(self.train_initializer):
            self.train_initializer()
        mean_loss = []
        num_iters = self.num_train // self.batch_size
        it = iter(self.train_iterator)

        def fetch():
            nonlocal it
            try:
                return next(it)
            except StopIteration:
                it = iter(self.train_iterator)
                return next(it)
        # eager launches on a GPU with the side stream on: the next batch's pixels are copied under the running step (_prefetch_issue)
        overlap = (self.dev.type == 'cuda' and self._side is not None and self.use_graph is False and not getattr(self, 'use_list', False)
                   and self.config.get('side_front', True) and self.config.get('prefetch_images', True) and self.data_format != 'channels_first')
        nxt = fetch() if num_iters > 0 else None
        for i in range(num_iters):
            images, gt = nxt
            self.set_batch(images, gt)
            nxt = fetch() if i + 1 < num_iters else None
            if overlap and nxt is not None:
                host = torch.as_tensor(nxt[0], dtype=torch.float32)
                if tuple(host.shape) == tuple(self.images.shape):
                    self._prefetch_next = (nxt[0], host)
            loss = float(self.train_step(lr).item())
            self._prefetch_next = None
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
        assert self.batch_size == 1 and tuple(images.shape) == (1, self.INPUT_SIZE, self.INPUT_SIZE, 3)
        self.images.copy_(images)
        # Reference quirk, reproduced by default: test mode re-binds self.images to `placeholder - mean`
        # and feeds THAT tensor (SSD300.py:65-66,487), so fed pixels bypass the mean subtraction.
        self._forward(False, subtract_mean=bool(self.config.get('test_subtract_mean', False)))
        nc = self.num_classes - 1
        pri = self.pri
        ops.ssd_decode(self.pred[0], self.num_classes, pri[2], pri[3], self.nms_score_threshold, self.d_conf,
                       self.d_boxes, self.d_keep, self.d_cand)
        cap = max(int(self.nms_max_boxes), 1)
        out_idx = torch.zeros(nc, cap, dtype=torch.int32, device=self.dev)
        out_cnt = torch.zeros(nc, dtype=torch.int32, device=self.dev)
        ops.nms_batched(self.d_boxes, 0, self.d_conf, 1, nc, self.d_cand, 1, nc, 1, self.NUM_PRIORS, nc, None, 0,
                        int(self.nms_max_boxes), self.nms_iou_threshold, out_idx, cap, out_cnt)
        cnt = out_cnt.cpu().tolist()
        idx = out_idx.cpu()
        conf, boxes = self.d_conf.cpu(), self.d_boxes.cpu()
        scores, bbox, cid = [], [], []
        for c in range(nc):                           # ascending class id, NMS pick order inside
            ids = idx[c, : cnt[c]].long()
            scores.append(conf[ids, c]); bbox.append(boxes[ids])
            cid.append(torch.full((cnt[c],), c, dtype=torch.int32))
        return [torch.cat(scores).numpy(), torch.cat(bbox, 0).numpy().reshape(-1, 4), torch.cat(cid).numpy()]

    # ------------------------------------------------------------------ checkpoints
    def tf_variable_map(self):
        return reference_variable_map(self.EXTRA_SEQ, self.NH)

    def _logical(self, name, buf):
        """parameter `name` out of a flat buffer (P or Mom) in TensorFlow's layout: kernels HWIO, un-padded"""
        v = self.param(name, buf).detach().cpu()
        if name.endswith('.w'):
            v = v[..., : self.convs[name[:-2]].cin].permute(1, 2, 3, 0)
        return np.ascontiguousarray(v.numpy())

    def export_tf_variables(self):
        """what the reference's `tf.train.Saver()` (SSD300.py:464-466) would write: every global variable -- weights, BN
        moving statistics, global_step and the MomentumOptimizer slots, which are created inside the 'inference' scope
        (:104, :149) and therefore named inference/<variable>/Momentum."""
        self._sync_from_twin()
        out = OrderedDict()
        for tfname, ours in self.tf_variable_map().items():
            if ours in self.pinfo:
                out[tfname] = self._logical(ours, self.P)
                out[f'inference/{tfname}/Momentum'] = self._logical(ours, self.Mom)
            else:
                out[tfname] = self.stat(ours).detach().cpu().numpy().copy()
        out['global_step'] = np.asarray(self.global_step, dtype=np.int32)
        return out

    def load_tf_checkpoint(self, path):
        """`saver.restore(sess, path)` (SSD300.py:502-504) from the files of a reference-trained model (or ours)."""
        from .tf_checkpoint import NewCheckpointReader
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()
        reader = NewCheckpointReader(str(path))
        names = reader.get_variable_to_shape_map()
        for tfname, ours in self.tf_variable_map().items():
            if ours in self.pinfo:
                v = torch.from_numpy(reader.get_tensor(tfname))                 # KeyError = Saver's NotFoundError
                self.set_param(ours, v.permute(3, 0, 1, 2).contiguous() if ours.endswith('.w') else v)
                slot = [k for k in names if k.endswith(tfname + '/Momentum')]
                if slot:
                    mv = torch.from_numpy(reader.get_tensor(slot[0]))
                    dst = self.param(ours, self.Mom)
                    if ours.endswith('.w'):
                        dst.zero_()
                        dst[..., : mv.shape[2]] = mv.permute(3, 0, 1, 2).to(self.dev)
                    else:
                        dst.copy_(mv.to(self.dev).view(dst.shape))
            else:
                self.stat(ours).copy_(torch.from_numpy(reader.get_tensor(tfname)).to(self.dev))
        if reader.has_tensor('global_step'):
            self.global_step = int(reader.get_tensor('global_step'))
        self._refresh_operand_copies()

    def _save_weight_engine(self, mode, path):
        """SSD300.py:490-500.  config['checkpoint_format'] = 'tf' writes the reference's own files
        (`<path>-<step>.index` + `.data-00000-of-00001` + `checkpoint`, readable by its `load_weight`); the default
        'torch' keeps one torch file `<path>-<step>`."""
        assert (mode in ['latest', 'best'])
        dirname = os.path.dirname(path)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
            print(dirname, 'does not exist, create it done')
        if self.checkpoint_format == 'tf':
            from . import tf_checkpoint
            prefix = path + '-' + str(self.global_step)
            tf_checkpoint.write_bundle(prefix, self.export_tf_variables())
            tf_checkpoint.update_checkpoint_state(prefix)
            print('save', mode, 'model in', path, 'successfully')
            return
        blob = {'params': self.export_params(), 'momentum': self.Mom.detach().cpu(),
                'global_step': self.global_step, 'layout': {k: (int(o), tuple(int(x) for x in shp)) for k, (o, shp) in self.pinfo.items()}}
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

    # ------------------------------------------------------------------ data parallel
    def attach_data_parallel(self, group=None, bucket_mb=25, sync_bn=False, grad_dtype='f32', force_collectives=False, collective='torch'):
        """Shard images over ranks (one process per GPU); gradients are summed with bucketed
        RCCL all-reduce overlapped with backward.  The loss divisor becomes the GLOBAL batch.
        grad_dtype 'bf16': the buckets travel as bf16 copies (half the xGMI bytes); force_collectives: issue them in a world of one rank too;
        collective 'odtk': the sums go through the C-ABI's own collective (odtk_comm_allreduce) instead of torch.distributed's."""
        from .dist import GradAllReducer
        self.dist = GradAllReducer(self, group, bucket_mb, grad_dtype, force_collectives, collective)
        self.dist.red.launch_ctx = self._comm_launch
        self._graphs_invalidate()
        self.loss_divisor_batch = self.batch_size * self.dist.world
        if sync_bn:
            # SURVEY.md 8e option B: batch statistics over all replicas -- W ranks x B images compute exactly what one device
            # computes on W*B (strong scaling with reference semantics).  Two small collectives per BN layer and pass sit inside
            # the step, so the launches stay eager (no HIP-graph replay).
            self.sync_bn = ops.SyncBN(group)
            self.use_graph = False
        return self.dist
