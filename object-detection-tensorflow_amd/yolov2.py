"""YOLOv2 (Darknet-19 + passthrough + one 5-prior head) behind the reference's class surface, on libodtk.

Reference: /root/reference/YOLOv2.py (class YOLOv2; driver testYOLOv2.py: 480 x 480, five priors in cell units, scales 1 / 1 / 5 / 1)
  * constructor, config keys ............. :11-53
  * input ................................ :55-72   (images - mean; test mode feeds the tensor after the subtraction -- reproduced, 'test_subtract_mean' opts out)
  * backbone 'backone' ................... :261-312  18 x [conv(bias) + batch norm + leaky_relu(0.1)], five 2x2 / s2 SAME max pools; features lrelu18 (1024),
                                                      "passthrough" lrelu17 (512, same resolution), stride 32
  * head ................................. :75-99    5 x [conv + BN + leaky], concat(passthrough, lrelu5), 1x1 conv + BN to (classes + 5) * priors, no activation;
                                                      the batch norm writes straight into the f32 prediction tensor [N, H*W*priors, classes + 5]
  * loss, optimizer ...................... :101-175  odtk_yolov2_loss (heads.YOLOv2Loss), mean over the batch + weight_decay * l2, Momentum 0.9
  * inference ............................ :177-201  odtk_yolov2_decode_candidates + per-class NMS (heads.yolov2_detect)
  * train / test / checkpoints ........... :313-360  (load_pretraining_weight: the backbone's trainables from a tf.train.Saver checkpoint)
The graph engine is refinedet.RefineDet320's (per-activation gradient buffers, transposed-free here: 'bn' layers with activation code 2 = leaky, 'pool', 'concat').
"""
from __future__ import annotations

import numpy as np
import torch

from . import heads, ops
from ._lib import BF16, F32, F32X3
from .refinedet import RefineDet320

BACKBONE = [(32, 3), 'P', (64, 3), 'P', (128, 3), (64, 1), (128, 3), 'P', (256, 3), (128, 1), (256, 3), 'P',
            (512, 3), (256, 1), (512, 3), (256, 1), (512, 3), 'P', (1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]
HEAD = [(1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]
STRIDE = 32.0
LEAKY = 2


def layer_specs(num_classes, num_priors):
    """[(name, kind, cin, cout, k, stride, dil, activation)] in TensorFlow's creation order: b1..b18 (backbone), h1..h5, pred"""
    s, c, i = [], 3, 0
    for l in BACKBONE:
        if l != 'P':
            i += 1
            s.append((f'b{i}', 'conv', c, l[0], l[1], 1, 1, LEAKY)); c = l[0]
    for j, (co, k) in enumerate(HEAD):
        s.append((f'h{j + 1}', 'conv', c, co, k, 1, 1, LEAKY)); c = co
    s.append(('pred', 'conv', 512 + 1024, (num_classes + 5) * num_priors, 1, 1, 1, 0))
    return s


def reference_variable_map(num_layers_backbone=18):
    """our parameter / statistic name -> the reference graph's variable name (default layer names are numbered per enclosing variable scope)"""
    m = {}

    def bn(scope, k):
        return f'{scope}/batch_normalization' + (f'_{k}' if k else '')
    for i in range(1, num_layers_backbone + 1):
        m[f'b{i}'] = (f'backone/conv{i}', bn('backone', i - 1))
    for j in range(1, 6):
        m[f'h{j}'] = (f'head/conv{j}', bn('head', j - 1))
    m['pred'] = ('head/predictions', bn('head', 5))
    out = {}
    for ours, (conv, b) in m.items():
        out[ours + '.w'], out[ours + '.b'] = conv + '/kernel', conv + '/bias'
        for a, t in (('gamma', 'gamma'), ('beta', 'beta'), ('mmean', 'moving_mean'), ('mvar', 'moving_variance')):
            out[f'{ours}.{a}'] = f'{b}/{t}'
    return out


class YOLOv2(RefineDet320):
    L2_AFTER = None
    NAME = 'YOLOv2'
    DEFAULT_ENGINE = 'bf16'                 # passes the gate (filter-gradient cosine vs f32 after 300 f32 steps: input side 0.93, minimum 0.91)

    def __init__(self, config, data_provider):
        assert len(config['data_shape']) == 3
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.data_shape = config['data_shape']
        self.num_classes = config['num_classes']
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1
        self.coord_sacle = config['coord_scale']                # (sic, YOLOv2.py:25)
        self.noobj_scale = config['noobj_scale']
        self.obj_scale = config['obj_scale']
        self.class_scale = config['class_scale']
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.rescore_confidence = config['rescore_confidence']
        self.num_priors = len(config['priors'])
        self.priors_flat = [float(v) for hw in config['priors'] for v in hw]
        self.final_units = (self.num_classes + 5) * self.num_priors
        h, w, c = self.data_shape if self.data_format == 'channels_last' else (self.data_shape[1], self.data_shape[2], self.data_shape[0])
        assert c == 3 and h % 32 == 0 and w % 32 == 0, "YOLOv2 needs an input that is a multiple of 32 (five 2x2 pools)"
        self._hw = (h, w)
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
        self.specs = layer_specs(self.num_classes, self.num_priors)
        self._init_parameters(int(config.get('seed', 0)))
        self._build()
        self._warmup_setup(config, data_provider, 'compute_dtype' in config)

    MOMENTUM_SLOT_SCOPE = ''                # the optimizer is created outside every variable scope (YOLOv2.py:168)

    def reference_variable_map(self):
        return reference_variable_map()

    def _input_hw(self):
        return self._hw

    def _load_pretraining_weight(self):
        pass

    def _build_model(self, h):
        N, dev = self.batch_size, self.dev
        H, W = self._hw[0] // 32, self._hw[1] // 32
        self.grid = (H, W)
        self.A = H * W * self.num_priors                        # prediction rows per image
        self.pred = torch.zeros(N, self.A, self.num_classes + 5, device=dev)
        x, i, passthrough = self.input, 0, None
        for l in BACKBONE:
            if l == 'P':
                x = h.pool(f'pool{i}', x, 2, 2)
            else:
                i += 1
                x = h.bn(f'b{i}', x)
                if i == 17:
                    passthrough = x
        for j in range(1, 6):
            x = h.bn(f'h{j}', x)
        x = h.concat('cat', [passthrough, x])
        h.bn('pred', x, ('pred', 0, self.num_classes + 5))

    def _make_loss(self, pad):
        H, W = self.grid
        return heads.YOLOv2Loss(self.batch_size, H, W, self.num_priors, self.num_classes, self.priors_flat,
                                (self.coord_sacle, self.noobj_scale, self.obj_scale, self.class_scale), self.dev)

    def _loss_step(self):
        parts = self.loss(self.pred, self.gt, 1.0 / self.loss_divisor_batch, STRIDE)
        return parts[:, 4].sum() / self.batch_size

    def test_one_image(self, images):
        images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        if self.data_format == 'channels_first' and images.shape[1] == 3:
            images = images.permute(0, 2, 3, 1)
        assert self.batch_size == 1 and tuple(images.shape) == tuple(self.images.shape), images.shape
        self.images.copy_(images)
        self._forward(False, subtract_mean=bool(self.config.get('test_subtract_mean', False)))      # reference quirk: the fed tensor is `images - mean`
        H, W = self.grid
        scores, bbox, cid = heads.yolov2_detect(self.pred[0].view(H, W, self.num_priors, self.num_classes + 5), self.priors_flat,
                                                self.nms_score_threshold, self.nms_max_boxes, self.nms_iou_threshold, STRIDE)
        return [scores.cpu().numpy(), bbox.cpu().numpy().reshape(-1, 4), cid.cpu().numpy()]

    def load_pretraining_weight(self, path):
        """`self.pretraining_weight_saver.restore` (YOLOv2.py:206-208, :355-357): the trainables of scope 'backone' from a tf.train.Saver checkpoint"""
        if getattr(self, 'f32_warmup_steps', 0):
            self.cancel_warmup()                                   # weights are loaded: the run does not start from random initialisation
        from .tf_checkpoint import NewCheckpointReader
        reader = NewCheckpointReader(str(path))
        names = reference_variable_map()
        for ours in self.pinfo:
            if ours.startswith('b'):
                v = torch.from_numpy(reader.get_tensor(names[ours]))          # KeyError = Saver's NotFoundError
                self.set_param(ours, v.permute(3, 0, 1, 2).contiguous() if ours.endswith('.w') else v)
        self._refresh_operand_copies()
        print('>> load pretraining weight', path, 'successfully')
