#!/usr/bin/env python
"""The reference's driver flow (testSSD300.py / testYOLOv3.py: config dict -> data provider -> model -> train_one_epoch ->
save_weight -> test_one_image) on synthetic VOC-shaped pictures, with the GPU augmentor in front of the model.
Needs an MI355X:   python examples/train_synthetic.py [ssd300|yolov3|retinanet] [epochs]

What changes for a user of the reference:
    import SSD300 as net; model = net.SSD300(config, data_provider)        ->   from odtk import SSD300
    utils.tfrecord_voc_utils.get_generator(..., image_augmentor_config)     ->   any iterable of (images, ground_truth) batches;
                                                                                 odtk.augment.Augmentor takes the same config dict
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import odtk                                   # noqa: E402
from odtk.augment import Augmentor            # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'ssd300'
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
size = {'ssd300': 300, 'yolov3': 416, 'retinanet': 512}[which]
batch_size = 8
dev = torch.device('cuda:0')

image_augmentor_config = {                    # testSSD300.py:34-46
    'data_format': 'channels_last', 'output_shape': [size, size], 'crop_method': 'random', 'flip_prob': [0., 0.5], 'fill_mode': 'BILINEAR',
    'keep_aspect_ratios': False, 'constant_values': 0., 'color_jitter_prob': 0.5, 'rotate': [0.5, -5., -5.], 'pad_truth_to': 60,
}


class SyntheticVOC:
    """decoded pictures of different sizes + [ymin, ymax, xmin, xmax, class] boxes, augmented on the GPU per batch"""

    def __init__(self, batches, seed=0):
        self.batches, self.rng, self.aug = batches, np.random.default_rng(seed), Augmentor(seed=seed, **image_augmentor_config)

    def __iter__(self):
        for _ in range(self.batches):
            imgs, gts = [], []
            for _ in range(batch_size):
                h, w = int(self.rng.integers(300, 500)), int(self.rng.integers(300, 500))
                imgs.append(torch.from_numpy(self.rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).to(dev))
                k = int(self.rng.integers(1, 5))
                yc, xc = self.rng.uniform(0.3, 0.7, k) * h, self.rng.uniform(0.3, 0.7, k) * w
                bh, bw = self.rng.uniform(0.1, 0.5, k) * h, self.rng.uniform(0.1, 0.5, k) * w
                gts.append(torch.tensor(np.stack([yc - bh / 2, yc + bh / 2, xc - bw / 2, xc + bw / 2, self.rng.integers(0, 20, k)], 1), dtype=torch.float32))
            yield self.aug(imgs, gts)


data_provider = {'data_shape': [size, size, 3], 'num_train': 4 * batch_size, 'num_val': 0, 'train_generator': SyntheticVOC(4), 'val_generator': None}
if which == 'ssd300':
    config = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': batch_size,
              'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': './vgg_16.ckpt'}   # testSSD300.py:15-32
    model = odtk.SSD300(config, data_provider)
elif which == 'retinanet':
    config = {'is_bottleneck': True, 'residual_block_list': [3, 4, 6, 3], 'init_conv_filters': 16, 'mode': 'train', 'is_pretraining': False,
              'data_shape': [size, size, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
              'batch_size': batch_size, 'gamma': 2.0, 'alpha': 0.25, 'nms_score_threshold': 0.8, 'nms_max_boxes': 10,
              'nms_iou_threshold': 0.45}                                                                # testretinanet.py:22-41
    model = odtk.RetinaNet(config, data_provider)
else:
    config = {'mode': 'train', 'data_shape': [size, size, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
              'batch_size': batch_size, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3,
              'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.5,
              'priors': [[[10., 13.], [16, 30.], [33., 23.]], [[30., 61.], [62., 45.], [59., 119.]], [[116., 90.], [156., 198.], [373., 326.]]]}  # testYOLOv3.py:17-41
    model = odtk.YOLOv3(config, data_provider)
for epoch in range(epochs):
    print('-' * 20, 'epoch', epoch, '-' * 20)
    mean_loss = model.train_one_epoch(0.001)
    print('>> mean loss', mean_loss)
    model.save_weight('latest', os.path.join('/tmp', which, 'test'))
test = type(model)(dict(config, mode='test'), None)
test.load_weight(os.path.join('/tmp', which, 'test') + '-' + str(model.global_step))
scores, bbox, class_id = test.test_one_image(np.random.default_rng(1).integers(0, 256, (1, size, size, 3)).astype(np.float32))
print('detections:', len(scores))
