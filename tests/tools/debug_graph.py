import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, odtk
from oracle import ssd300_ref as R
CONFIG = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
          'batch_size': 2, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5,
          'pretraining_weight': './vgg_16.ckpt', 'verbose': False}
imgs, gt = R.synthetic_batch(2, 31)
for tag, ug in (('eagerA', False), ('eagerB', False), ('graph', True), ('graph2', True)):
    cfg = dict(CONFIG, compute_dtype='f32', batch_size=2, use_graph=ug, seed=4)
    m = odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': 2, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    m.set_batch(imgs, gt)
    print(tag, [round(float(m.train_step(0.005).item()), 6) for _ in range(6)])
