"""Light-Head R-CNN behind the reference's class surface, on libodtk.

Reference: /root/reference/LH_RCNN.py (class LHRCNN; driver testlhrcnn.py: 700 x 1100, batch 32, the three-phase schedule keys)
  * constructor, config keys ............. :11-54
  * input ................................ :58-72    images / 127.5 - 1; test mode feeds the tensor AFTER that line (:467) -- reproduced, 'test_normalise' opts out
  * backbone 'feature_extractor' ......... :203-231  conv(bias) | separable(no bias) -> batch norm -> ReLU; 3x3 / s2 SAME max pool after conv1; stride 32
  * RPN 'rpn' ............................ :77-97    rpn_conv + two 3x3 heads whose batch norm writes straight into the f32 tensors rpn_conf [N, A, 2] / rpn_bbox [N, A, 4]
  * light head 'rcnn' .................... :99-104   separable 1x15 -> 15x1, twice, summed (490 channels)
  * RPN loss / R-CNN stage ............... :106-170  odtk_lhrcnn_match -> 2 x odtk_nms_batched -> odtk_lhrcnn_rpn_loss -> odtk_crop_and_resize_fwd -> three dense layers
                                                      (1x1 convolutions over the 256-row slots of every image) -> odtk_lhrcnn_rcnn_loss, and back
  * optimizer ............................ :171-201  see below
  * inference ............................ :134-138, :153-164, :203-236
  * train / test / checkpoints ........... :459-514
The graph engine is refinedet.RefineDet320's with two more plan entries: 'dw' (the depthwise half of a separable layer; its pointwise half is an ordinary 1x1 `bn`
layer) and 'sink' (an activation handed to the R-CNN stage, which lives outside the plan because its row count changes per step).

What the reference's training graph does, as TensorFlow executes it (oracle/lhrcnn_ref.py header has the reasoning; tests/golden/lhrcnn_train.npz pins it on the
reference's own class): BOTH optimizer ops run on every step -- backbone + RPN move by rpn_loss, the light head by rcnn_loss, and the light head's loss does not
reach the backbone (its variable list stops at 'rcnn', :190) -- global_step advances every step, and the schedule keys only choose which of the two losses
train_one_epoch REPORTS.  One fused momentum launch per variable group covers that.  The label of a box's best anchor is tf.gather's GPU result (0 when the
anchor index is out of range, :337); the R-CNN centre target is divided by the proposal's centre (:430).
f32 engine by default; `compute_dtype='bf16'` is an opt-in that has not run on the GPU as a whole yet (see __init__).  Data parallel: `attach_data_parallel` (bucketed all-reduce of the flat gradient buffer, the
dense layers' bucket first); each replica normalises the R-CNN loss by its own row counts.
"""
from __future__ import annotations

import sys
from collections import OrderedDict

import numpy as np
import torch

from . import heads, ops
from ._lib import BF16, F32, F32X3
from .refinedet import RefineDet320

ANCHOR_SCALES = [32, 64, 128, 256, 512]
ANCHOR_RATIOS = [0.5, 1.0, 2.0]
NA = len(ANCHOR_SCALES) * len(ANCHOR_RATIOS)
STRIDE = 32.0
CROP, HEAD_CH, ROIS = 7, 490, 256
RCNN_FIRST = 'state5_conv1_1'              # first layer (creation order) of the 'rcnn' variable scope: the flat parameter buffer splits here


def layer_table(num_classes):
    """[(name, kind, cin, cout, kh, kw, stride, relu)] in TensorFlow's creation order; kind 'conv' | 'sep' (both + batch norm) | 'dense'"""
    s = [('conv1', 'conv', 3, 24, 3, 3, 2, True)]
    c = 24
    for stage, ch, nsep in ((2, 144, 3), (3, 288, 7), (4, 576, 3)):
        s.append((f'stage{stage}_sconv1', 'conv', c, ch, 3, 3, 2, True))
        s += [(f'stage{stage}_sconv{j}', 'sep', ch, ch, 3, 3, 1, True) for j in range(2, nsep + 2)]
        c = ch
    s += [('rpn_conv', 'conv', 576, 256, 3, 3, 1, True), ('rpn_conf', 'conv', 256, NA * 2, 3, 3, 1, False), ('rpn_pbbox', 'conv', 256, NA * 4, 3, 3, 1, False)]
    for b in (1, 2):
        s += [(f'state5_conv{b}_1', 'sep', 576, 256, 1, 15, 1, True), (f'state5_conv{b}_2', 'sep', 256, HEAD_CH, 15, 1, 1, True)]
    s += [('roi_feat_dense', 'dense', CROP * CROP * HEAD_CH, 2048, 1, 1, 1, True), ('rcnn_pconf', 'dense', 2048, num_classes, 1, 1, 1, False),
          ('rcnn_pbbox', 'dense', 2048, 4, 1, 1, 1, False)]
    return s


def make_anchors(fh, fw, img_h, img_w):
    """LH_RCNN.py:240-261 and the inside-the-picture mask of :87-97, in float32 as the graph computes them.  -> (y1x1, y2x2, yx, hw [A, 2] of the kept
    anchors, row [A] int32 = their index among all fh * fw * 15 anchors)"""
    f32 = torch.float32
    cy = (torch.arange(fh, dtype=f32) + 0.5).view(fh, 1, 1, 1).expand(fh, fw, NA, 1)
    cx = (torch.arange(fw, dtype=f32) + 0.5).view(1, fw, 1, 1).expand(fh, fw, NA, 1)
    centre = torch.cat([cy, cx], -1) * STRIDE
    sizes = torch.tensor([[s * (r ** 0.5), s / (r ** 0.5)] for s in ANCHOR_SCALES for r in ANCHOR_RATIOS], dtype=f32).view(1, 1, NA, 2)
    y1x1, y2x2 = (centre - sizes / 2.).reshape(-1, 2), (centre + sizes / 2.).reshape(-1, 2)
    yx, hw = y1x1 / 2. + y2x2 / 2., y2x2 - y1x1
    h, w = float(img_h - 1), float(img_w - 1)                   # self.h, self.w (:36-39); the mask subtracts one more (:88-89)
    keep = (y1x1[:, 0] >= 0.) & (y1x1[:, 1] >= 0.) & (y2x2[:, 0] <= h - 1) & (y2x2[:, 1] <= w - 1)
    row = torch.nonzero(keep).flatten().to(torch.int32)
    return y1x1[keep].contiguous(), y2x2[keep].contiguous(), yx[keep].contiguous(), hw[keep].contiguous(), row


class _Stage:
    """buffers of the RPN loss and of the R-CNN stage for `rows` crop rows (training: 256 per image; inference: post_nms_proposal)"""

    def __init__(self, m, rows, pad):
        dev, dt = m.dev, m.tdt
        N = m.batch_size
        A = m.anc['yx'].shape[0]
        self.rows = rows
        if pad is not None:
            self.ws = ops.lhrcnn_workspace(N, A, pad, dev)
            self.d_rpn_conf = torch.zeros(N, m.A, 2, device=dev)
            self.d_rpn_bbox = torch.zeros(N, m.A, 4, device=dev)
        ch = m.chunk
        self.ldr = ops.pad_to(CROP * CROP * HEAD_CH, ch)
        self.ldl, self.ldb = ops.pad_to(m.num_classes, ch), ops.pad_to(4, ch)
        self.roi = torch.zeros(rows, self.ldr, dtype=dt, device=dev)
        self.fc1 = torch.zeros(rows, 2048, dtype=dt, device=dev)
        self.logits = torch.zeros(rows, self.ldl, dtype=dt, device=dev)
        self.pbbox = torch.zeros(rows, self.ldb, dtype=dt, device=dev)
        self.desc = {'roi_feat_dense': ops.conv_desc(rows, 1, 1, self.ldr, self.ldr, 2048, 2048, 1, 1, 1, m.CDT, m.CDT),
                     'rcnn_pconf': ops.conv_desc(rows, 1, 1, 2048, 2048, m.num_classes, self.ldl, 1, 1, 1, m.CDT, m.CDT),
                     'rcnn_pbbox': ops.conv_desc(rows, 1, 1, 2048, 2048, 4, self.ldb, 1, 1, 1, m.CDT, m.CDT)}
        if pad is not None:
            self.d_logits = torch.zeros(rows, self.ldl, dtype=dt, device=dev)
            self.d_pbbox = torch.zeros(rows, self.ldb, dtype=dt, device=dev)
            self.d_fc1 = torch.zeros(rows, 2048, dtype=dt, device=dev)
            self.d_roi = torch.zeros(rows, self.ldr, dtype=dt, device=dev)
        # the loss / decode kernels read and write f32: on the bf16 engine the head's outputs and their gradients pass through f32 twins
        f32 = dt == torch.float32
        self.logits32 = self.logits if f32 else torch.zeros(rows, self.ldl, device=dev)
        self.pbbox32 = self.pbbox if f32 else torch.zeros(rows, self.ldb, device=dev)
        if pad is not None:
            self.d_logits32 = self.d_logits if f32 else torch.zeros(rows, self.ldl, device=dev)
            self.d_pbbox32 = self.d_pbbox if f32 else torch.zeros(rows, self.ldb, device=dev)
            self.d_feat32 = None if f32 else torch.zeros(m.feat.M, m.feat.ld, device=dev)


class LHRCNN(RefineDet320):
    NAME = 'LHRCNN'
    # Training default: the EXACT f32 engine (round 5).  'f32x3' (compute_dtype='f32x3': f32 tensors, convolutions / dense layers as three bf16 MFMA products, 440 -> 595
    # images/s) stays opt-in for this class: it is the one class the gradient-direction gate (tools/bf16_after_training.py, profiles/r04x_gate_f32x3_all_classes.md) was
    # never run for, and its second training step sits 6-9 % off the oracle's loss where the exact engine holds 5e-2 (tests/test_gpu_lhrcnn.py).
    DEFAULT_ENGINE = 'f32'
    L2_AFTER = None
    MOMENTUM_SLOT_SCOPE = 'rcnn/'           # the optimizer is created inside `with tf.variable_scope('rcnn')` (LH_RCNN.py:98, :171)

    def __init__(self, config, data_provider):
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.data_shape = config['data_shape']
        self.num_classes = config['num_classes'] + 1          # background = LAST index
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.rpn_first_step = config['rpn_first_step']
        self.rcnn_first_step = config['rcnn_first_step']
        self.rpn_second_step = config['rpn_second_step']
        self.post_nms_proposal = config['post_nms_proposal']
        self.anchor_scales, self.anchor_ratios, self.num_anchors = ANCHOR_SCALES, ANCHOR_RATIOS, NA
        h, w, c = self.data_shape if self.data_format == 'channels_last' else (self.data_shape[1], self.data_shape[2], self.data_shape[0])
        assert c == 3
        self._hw = (int(h), int(w))
        self.h, self.w = float(h - 1), float(w - 1)
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
        # 'bf16' (opt-in, end of round 3): the engine's bf16 kernels everywhere, the head's outputs widened to f32 in front of the loss / decode kernels.  Built from
        # launches that are each verified on MI355X (bf16 storage of the depthwise / crop kernels, the bf16 convolutions, the casts) but NOT yet run as a whole on
        # the GPU, and not put through the gradient-direction gate of DESIGN.md 5: f32 stays the default.
        engine = config.get('compute_dtype', self.DEFAULT_ENGINE if config['mode'] == 'train' else 'f32')
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
        self.table = layer_table(self.num_classes)
        # what the engine sees: a separable layer is its pointwise 1x1 convolution (+ batch norm) behind a 'dw' plan entry; a dense layer is a 1x1 convolution with bias
        self.specs = [(n, 'vgg' if kind == 'dense' else 'conv', cin, cout, 1 if kind != 'conv' else kh, stride, 1, relu) for n, kind, cin, cout, kh, kw, stride, relu in self.table]
        self._sep = {n: (kh, kw, cin) for n, kind, cin, _, kh, kw, _, _ in self.table if kind == 'sep'}
        self._init_parameters(int(config.get('seed', 0)))
        g = torch.Generator().manual_seed(int(config.get('seed', 0)) + 1)
        for n, (kh, kw, cin) in self._sep.items():
            self.set_param(n + '.dw', torch.randn(kh, kw, cin, generator=g) * (2.0 / (kh * kw)) ** 0.5)
        self._build()
        self._warmup_setup(dict(config, f32_warmup_steps=config.get('f32_warmup_steps', 0)), data_provider, True)
        self._infer = None

    # ------------------------------------------------------------------ parameters
    def _extra_layer_params(self, spec):
        if spec[0] in self._sep:
            kh, kw, cin = self._sep[spec[0]]
            return ((spec[0] + '.dw', (kh, kw, cin)),)
        return ()

    def load_oracle_params(self, p):
        super().load_oracle_params({k: (v.reshape(v.shape[0], 1, 1, v.shape[1]) if (k.endswith('.w') and v.dim() == 2) else v) for k, v in p.items()})

    def export_params(self):
        """oracle layout: dense kernels [units, in]; the separable layers' inert zero bias (the engine's 1x1 layer carries one, TensorFlow's has none) is dropped"""
        out = OrderedDict()
        for k, v in super().export_params().items():
            if k.endswith('.b') and k[:-2] in self._sep:
                continue
            out[k] = v.reshape(v.shape[0], v.shape[3]) if (k.endswith('.w') and k[:-2] in ('roi_feat_dense', 'rcnn_pconf', 'rcnn_pbbox')) else v
        return out

    def _load_pretraining_weight(self):
        pass

    def _input_hw(self):
        return self._hw

    def _preprocess_input(self, normalise):
        if normalise:
            ops.preprocess_norm(self.images, 127.5, (1., 1., 1.), (1., 1., 1.), self.input.ld, self.DT, self.input.t)     # images / 127.5 - 1
        else:
            ops.preprocess(self.images, (0., 0., 0.), self.input.ld, self.DT, self.input.t)

    # ------------------------------------------------------------------ the graph
    def _build_model(self, h):
        N, dev = self.batch_size, self.dev
        t = {s[0]: s for s in self.table}

        def layer(name, x, stop_grad=False):
            if name in self._sep:
                kh, kw, _ = self._sep[name]
                x = h.dw(name, x, kh, kw, stop_grad)
            return h.bn(name, x)
        x = layer('conv1', self.input)
        x = h.pool('pool1', x, 3, 2)
        for s in self.table[1:]:
            if s[0].startswith('stage'):
                x = layer(s[0], x)
        c4 = x
        self.fh, self.fw = c4.H, c4.W
        self.A = c4.H * c4.W * NA                               # rows per image of the two RPN prediction tensors (all anchors)
        y1x1, y2x2, yx, hw, row = make_anchors(c4.H, c4.W, *self._hw)
        assert row.numel() > 0, f"no anchor lies inside a {self._hw[0]} x {self._hw[1]} picture"
        self.anc = dict(y1x1=y1x1.to(dev), y2x2=y2x2.to(dev), yx=yx.to(dev), hw=hw.to(dev), row=row.to(dev), A_full=self.A)
        self.rpn_conf = torch.zeros(N, self.A, 2, device=dev)
        self.rpn_bbox = torch.zeros(N, self.A, 4, device=dev)
        r = layer('rpn_conv', c4)
        h.bn('rpn_conf', r, ('rpn_conf', 0, 2))
        h.bn('rpn_pbbox', r, ('rpn_bbox', 0, 4))
        # the light head reads c4 but never sends a gradient into it: its loss only trains the 'rcnn' variables (LH_RCNN.py:190-191)
        b1 = layer('state5_conv1_2', layer('state5_conv1_1', c4, stop_grad=True))
        b2 = layer('state5_conv2_2', layer('state5_conv2_1', c4, stop_grad=True))
        self.feat = h.add('rcnn_feat', b1, b2)
        h.sink(self.feat)
        assert t['roi_feat_dense'][2] == CROP * CROP * self.feat.C

    def _make_loss(self, pad):
        return _Stage(self, ROIS * self.batch_size, pad)

    # ------------------------------------------------------------------ R-CNN stage
    def _dense_fwd(self, st):
        P = self.Pc
        ops.conv2d_fwd(st.desc['roi_feat_dense'], st.roi, self._flat('roi_feat_dense.w', P), self.param('roi_feat_dense.b'), st.fc1, True)
        ops.conv2d_fwd(st.desc['rcnn_pconf'], st.fc1, self._flat('rcnn_pconf.w', P), self.param('rcnn_pconf.b'), st.logits, False)
        ops.conv2d_fwd(st.desc['rcnn_pbbox'], st.fc1, self._flat('rcnn_pbbox.w', P), self.param('rcnn_pbbox.b'), st.pbbox, False)
        if self.DT == BF16:
            ops.cast_to_f32(st.logits, st.logits32)
            ops.cast_to_f32(st.pbbox, st.pbbox32)

    def _loss_step(self):
        """RPN loss (gradients into the prediction tensors' gradient buffers), then the whole R-CNN stage forward AND backward down to d(rcnn_feat);
        -> (rpn data loss, rcnn data loss) as device scalars"""
        st, ws, N = self.loss, self.loss.ws, self.batch_size
        H, W = self._hw
        ops.lhrcnn_match(self.anc, self.rpn_conf, self.gt, ws)
        cap = ws['cap']
        counts = ws['counts'].view(-1)
        ops.nms_batched(ws['pos_box'], cap * 4, ws['pos_score'], cap, 1, ws['pos_valid'], cap, 1, 1, cap, N, counts[3:], 8, 0, 0.7, ws['sel_pos'], 128, ws['cnt_pos'])
        ops.nms_batched(ws['neg_box'], cap * 4, ws['neg_score'], cap, 1, ws['neg_valid'], cap, 1, 1, cap, N, counts[4:], 8, 0, 0.7, ws['sel_neg'], 256, ws['cnt_neg'])
        ops.lhrcnn_rpn_loss(self.anc, self.rpn_conf, self.rpn_bbox, self.gt, ws, self.num_classes, 1.0 / self.loss_divisor_batch, H, W, st.d_rpn_conf, st.d_rpn_bbox)
        f = self.feat
        ops.crop_and_resize_fwd(f.t, f.ld, N, f.H, f.W, f.C, ws['roi_box'], ws['roi_img'], CROP, st.roi, st.ldr)
        self._dense_fwd(st)
        # data parallel: every replica takes the R-CNN means over ITS rows (the row counts live on the device; as with the batch-norm statistics each replica is
        # the reference computation on its own images) and the summed gradients are divided by the number of replicas; the RPN loss is a per-image mean
        ops.lhrcnn_rcnn_loss(st.logits32, st.ldl, st.pbbox32, st.ldb, N, self.num_classes, ws, float(self.batch_size) / self.loss_divisor_batch, st.d_logits32, st.d_pbbox32)
        if self.DT == BF16:
            ops.cast_from_f32(st.d_logits32, st.d_logits)
            ops.cast_from_f32(st.d_pbbox32, st.d_pbbox)
        # backward of the three dense layers: filter / bias gradients, then d(fc1) through its ReLU, then d(roi rows), then the image gradient of the crop
        G = self.G
        ops.conv2d_wgrad(st.desc['rcnn_pconf'], st.fc1, st.d_logits, st.ldl, self._flat('rcnn_pconf.w', G), self._flat('rcnn_pconf.b', G))
        ops.conv2d_wgrad(st.desc['rcnn_pbbox'], st.fc1, st.d_pbbox, st.ldb, self._flat('rcnn_pbbox.w', G), self._flat('rcnn_pbbox.b', G))
        ops.conv2d_dgrad(st.desc['rcnn_pconf'], st.d_logits, st.ldl, self.wt['rcnn_pconf'], st.fc1, st.d_fc1, False)
        ops.conv2d_dgrad(st.desc['rcnn_pbbox'], st.d_pbbox, st.ldb, self.wt['rcnn_pbbox'], st.fc1, st.d_fc1, True)
        ops.conv2d_wgrad(st.desc['roi_feat_dense'], st.roi, st.d_fc1, 2048, self._flat('roi_feat_dense.w', G), self._flat('roi_feat_dense.b', G))
        ops.conv2d_dgrad(st.desc['roi_feat_dense'], st.d_fc1, 2048, self.wt['roi_feat_dense'], None, st.d_roi, False)
        if self.DT == BF16:
            ops.crop_and_resize_bwd(st.d_roi, st.ldr, N, f.H, f.W, f.C, ws['roi_box'], ws['roi_img'], CROP, st.d_feat32, f.ld)
            ops.cast_from_f32(st.d_feat32, f.g)
        else:
            ops.crop_and_resize_bwd(st.d_roi, st.ldr, N, f.H, f.W, f.C, ws['roi_box'], ws['roi_img'], CROP, f.g, f.ld)
        return ws['rpn_parts'][:, 3].sum() / self.batch_size, ws['rcnn_parts'].sum()

    def _step_body(self):
        self.G.zero_()
        self._forward(True)
        self._rpn_loss, self._rcnn_loss = self._loss_step()
        if self.dist is not None:
            self.dist.layer_ready('roi_feat_dense')            # the three dense layers are the last segments of the flat gradient buffer
        for name in self._backward_iter():
            # a separable layer is complete when its depthwise filter gradient has been launched (the 'dw' entry follows its 1x1 layer in the backward plan)
            if self.dist is not None:
                base = name[:-3] if name.endswith('.dw') else name
                if name.endswith('.dw') or base not in self._sep:
                    self.dist.layer_ready(base)

    def _reported(self, step):
        """the loss tf.case selects for the report (LH_RCNN.py:198-203, :471-478)"""
        if step < self.rpn_first_step:
            return 'rpn_loss'
        if step < self.rcnn_first_step:
            return 'rcnn_loss'
        return 'rpn_loss' if step < self.rpn_second_step else 'rcnn_loss'

    def _train_step_engine(self, lr):
        """one step of BOTH optimizer ops (module docstring); returns the loss the schedule reports, data term + weight decay * l2 of that op's variables"""
        if self.dist is not None:
            self.dist.begin_step()
        self._step_body()
        self._eager_steps += 1
        if self.dist is not None:
            self.dist.finish_step()
        b = self.pinfo[RCNN_FIRST + '.dw'][0]                   # backbone + RPN variables | 'rcnn' variables
        bf = self.DT == BF16
        nb = ops.sgd_blocks(b)
        ops.sgd_momentum(self.P[:b], self.Mom[:b], self.G[:b], lr, 0.9, self.weight_decay, 1.0, self.l2_partial[:nb], self.Pc[:b] if bf else None)
        ops.sgd_momentum(self.P[b:], self.Mom[b:], self.G[b:], lr, 0.9, self.weight_decay, 1.0, self.l2_partial[nb:], self.Pc[b:] if bf else None)
        ops.sum_f32(self.l2_partial[:nb], self.l2_sum)
        ops.sum_f32(self.l2_partial[nb:], self.l2_sum2)
        self._fp_batch.run()
        which = self._reported(self.global_step)
        self.global_step += 1
        self.last_losses = (self._rpn_loss + self.weight_decay * self.l2_sum, self._rcnn_loss + self.weight_decay * self.l2_sum2)
        return self.last_losses[0 if which == 'rpn_loss' else 1]

    def _init_parameters(self, seed):
        super()._init_parameters(seed)
        b = self.pinfo[RCNN_FIRST + '.dw'][0]
        self.l2_partial = torch.zeros(ops.sgd_blocks(b) + ops.sgd_blocks(self.nparam - b), device=self.dev)
        self.l2_sum2 = torch.zeros(1, device=self.dev)

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
            step = self.global_step
            loss = float(self.train_step(lr).item())
            if self.verbose:
                print('iters ', str(i + 1) + str('/') + str(num_iters), self._reported(step), loss, 'global_step', step)
                sys.stdout.flush()
            mean_loss.append(loss)
        return np.mean(mean_loss)

    # ------------------------------------------------------------------ inference
    def test_one_image(self, images):
        images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert self.batch_size == 1 and tuple(images.shape) == tuple(self.images.shape), images.shape
        self.images.copy_(images)
        # reference quirk: `self.images` names the tensor after `/ 127.5 - 1` when test_one_image feeds it (LH_RCNN.py:68-69, :467): the fed pixels are used as they are
        self._forward(False, bool(self.config.get('test_normalise', False)))
        if self._infer is None:
            st = _Stage(self, int(self.post_nms_proposal), None)
            A, dev, R = self.anc['yx'].shape[0], self.dev, int(self.post_nms_proposal)
            st.prop, st.score = torch.zeros(A, 4, device=dev), torch.zeros(A, device=dev)
            st.sel, st.cnt = torch.zeros(1, R, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
            st.roi_box, st.roi_prop = torch.zeros(R, 4, device=dev), torch.zeros(R, 4, device=dev)
            st.roi_img = torch.zeros(R, dtype=torch.int32, device=dev)
            nc = self.num_classes - 1
            st.conf, st.boxes = torch.zeros(R, nc, device=dev), torch.zeros(R, 4, device=dev)
            st.cand = torch.zeros(R, nc, dtype=torch.uint8, device=dev)
            self._infer = st
        st = self._infer
        H, W = self._hw
        A, R = self.anc['yx'].shape[0], st.rows
        ops.lhrcnn_rpn_decode(self.anc, self.rpn_conf[0], self.rpn_bbox[0], H, W, st.prop, st.score)
        ops.nms_batched(st.prop, 0, st.score, 0, 1, None, 0, 0, 1, A, 1, None, 0, R, 0.7, st.sel, R, st.cnt)
        ops.lhrcnn_gather_rois(st.prop, st.sel, st.cnt, H, W, st.roi_box, st.roi_prop, st.roi_img)
        f = self.feat
        ops.crop_and_resize_fwd(f.t, f.ld, 1, f.H, f.W, f.C, st.roi_box, st.roi_img, CROP, st.roi, st.ldr)
        self._dense_fwd(st)
        ops.lhrcnn_rcnn_decode(st.logits32, st.ldl, st.pbbox32, st.ldb, st.roi_prop, st.roi_img, self.num_classes, self.nms_score_threshold, st.conf, st.boxes, st.cand)
        scores, bbox, cid = heads._per_class_nms(st.conf, st.boxes, st.cand, self.num_classes - 1, self.nms_max_boxes, self.nms_iou_threshold)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    # ------------------------------------------------------------------ the reference's variable names
    def reference_variable_map(self):
        """our parameter / statistic name -> the variable name in the reference's graph (tests/golden/lhrcnn_names.json)"""
        out, count = OrderedDict(), {}

        def bn_of(scope):
            k = count.get(scope, 0)
            count[scope] = k + 1
            return f'{scope}/batch_normalization' + (f'_{k}' if k else '')
        for name, kind, *_ in self.table:
            if name == 'conv1':
                scope = 'feature_extractor/stage1'
            elif name.startswith('stage'):
                scope = 'feature_extractor/' + name.split('_')[0]
            else:
                scope = 'rpn' if name.startswith('rpn') else 'rcnn'
            layer = f'{scope}/{name}'
            if kind == 'sep':
                out[name + '.dw'], out[name + '.w'] = layer + '/depthwise_kernel', layer + '/pointwise_kernel'
            else:
                out[name + '.w'], out[name + '.b'] = layer + '/kernel', layer + '/bias'
            if kind != 'dense':
                b = bn_of(scope)
                for a, t in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
                    out[f'{name}.{a}'] = f'{b}/{t}'
        return out

    def _logical(self, name, buf):
        v = self.get_param(name, buf)
        if name.endswith('.dw'):
            return np.ascontiguousarray(v.unsqueeze(-1).numpy())                      # [kh, kw, C, 1]
        if name.endswith('.w') and name[:-2] in ('roi_feat_dense', 'rcnn_pconf', 'rcnn_pbbox'):
            return np.ascontiguousarray(v.reshape(v.shape[0], v.shape[3]).t().numpy())   # [in, units]
        return np.ascontiguousarray((v.permute(1, 2, 3, 0) if name.endswith('.w') else v).numpy())

    @staticmethod
    def _from_tf(ours, arr):
        v = torch.from_numpy(np.asarray(arr))
        if ours.endswith('.dw'):
            return v.squeeze(-1)                                                      # [kh, kw, C, 1]
        if ours.endswith('.w'):
            return v.permute(3, 0, 1, 2).contiguous() if v.dim() == 4 else v.t().contiguous().reshape(v.shape[1], 1, 1, v.shape[0])
        return v

    def load_tf_checkpoint(self, path):
        """`saver.restore(sess, path)` from the files of a reference-trained model (or ours): weights, moving statistics, Momentum slots, global_step"""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()                                   # weights are loaded: the run does not start from random initialisation
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = reader.get_variable_to_shape_map()
        for ours, tfname in self.reference_variable_map().items():
            if ours in self.pinfo:
                self.set_param(ours, self._from_tf(ours, reader.get_tensor(tfname)))      # KeyError = Saver's NotFoundError
                slot = [k for k in names if k.endswith(tfname + '/Momentum')]
                if slot:
                    mv = self._from_tf(ours, reader.get_tensor(slot[0]))
                    dst = self.param(ours, self.Mom)
                    if ours.endswith('.w'):
                        dst.zero_()
                        dst[..., : mv.shape[-1]] = mv.to(self.dev)
                    else:
                        dst.copy_(mv.to(self.dev).view(dst.shape))
            else:
                self.stat(ours).copy_(torch.from_numpy(reader.get_tensor(tfname)).to(self.dev))
        if reader.has_tensor('global_step'):
            self.global_step = int(reader.get_tensor('global_step'))
        self._refresh_operand_copies()

    def load_pretraining_weight(self, path):
        """`self.pretraining_weight_saver.restore` (LH_RCNN.py:448-449, :511-513): the trainables of scope 'feature_extractor' from a tf.train.Saver checkpoint"""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()                                   # weights are loaded: the run does not start from random initialisation
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        for ours, tfname in self.reference_variable_map().items():
            if tfname.startswith('feature_extractor/') and ours in self.pinfo:
                v = torch.from_numpy(reader.get_tensor(tfname))                       # KeyError = Saver's NotFoundError
                if ours.endswith('.dw'):
                    v = v.squeeze(-1)
                elif ours.endswith('.w'):
                    v = v.permute(3, 0, 1, 2).contiguous()
                self.set_param(ours, v)
        self._refresh_operand_copies()
        print('>> load pretraining weight', path, 'successfully')
