#!/usr/bin/env python
"""Generates tests/golden/optimizer_scopes.json: for every reference detector class, the tf.variable_scope that is OPEN when the class
calls `optimizer.minimize(...)` -- constructed from the reference's own file on the eager TF-1.x shim (oracle/tf_shim records the scope
stack at that call).  TensorFlow creates the optimizer's slot variables (`<scope>/<variable>/Momentum`, `/Adam`, `/Adam_1`) and AdamOptimizer's
non-slot accumulators (`<scope>/beta1_power`, `beta2_power`) under that scope, so it is the prefix a tf.train.Saver file of the reference
carries -- what `export_tf_variables` / `load_tf_checkpoint` of every class here must write and accept
(tests/test_tf_checkpoint_cpu.py::test_optimizer_slot_scopes_follow_the_reference).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_optimizer_scopes.py
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import tf_shim    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
# class name -> (reference file, generator module whose CONFIG / batches() build the class at a small shape)
CLASSES = {
    'CenterNet': ('CenterNet.py', 'make_golden_centernet_net'),
    'FCOS': ('FCOS.py', 'make_golden_fcos_net'),
    'LHRCNN': ('LH_RCNN.py', 'make_golden_lhrcnn'),
    'PFPNetR': ('PFPNetR.py', 'make_golden_pfpnet'),
    'RefineDet320': ('RefineDet.py', 'make_golden_refinedet_net'),
    'RetinaNet': ('RetinaNet.py', 'make_golden_retinanet_net'),
    'YOLOv2': ('YOLOv2.py', 'make_golden_yolov2'),
    'YOLOv3': ('YOLOv3.py', 'make_golden_yolov3_train'),
}


def _vgg():
    import make_golden as MG
    from oracle import ssd300_ref as R
    return MG.vgg_tensors(R.init_params(7))          # what NewCheckpointReader('vgg_16.ckpt') hands the VGG-based classes


def build(cls, ref_file, gen):
    tf_shim.install(_vgg())
    tf_shim.S.reset()
    if cls == 'CenterNet':
        tf_shim.TRACE_DEAD_COND_BRANCHES = True
        sys.modules['tensorflow'].cond = tf_shim.cond
    tf_shim.GATHER_OOB_ZERO = cls == 'LHRCNN'          # LH_RCNN.py:337 only executes with tf.gather's GPU behaviour (oracle/lhrcnn_ref.py header)
    ref = tf_shim.load_reference_module('/root/reference/' + ref_file, 'reference_scope_' + cls)
    mod = importlib.import_module(gen)
    data = mod.batches()

    class It:
        def get_next(self):
            im, g = data[0]
            return tf_shim.wrap(im.clone()), tf_shim.wrap(g.clone())
    prov = {'num_train': 4, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
    if 'data_shape' in mod.CONFIG:
        prov['data_shape'] = mod.CONFIG['data_shape']
    getattr(ref, cls)(dict(mod.CONFIG), prov)
    tf_shim.TRACE_DEAD_COND_BRANCHES = False
    tf_shim.GATHER_OOB_ZERO = False
    return tf_shim.S.optimizer_scope


def main():
    out = {}
    only = sys.argv[1:]                       # `make_golden_optimizer_scopes.py LHRCNN`: rebuild these classes only, keep the other entries of the file
    if only:
        out = json.load(open(os.path.join(OUT, 'optimizer_scopes.json')))
        for cls in only:
            out[cls] = build(cls, *CLASSES[cls])
            print(cls, repr(out[cls]))
        with open(os.path.join(OUT, 'optimizer_scopes.json'), 'w') as f:
            json.dump(out, f, indent=0, sort_keys=True)
        return
    for cls, (ref_file, gen) in CLASSES.items():
        out[cls] = build(cls, ref_file, gen)
        print(cls, repr(out[cls]))
    # SSD300 / SSD512: make_golden.py's configuration (the empty `else:` of the shipped files is repaired in memory by the loader)
    import make_golden as MG
    from oracle import ssd300_ref as R
    for cls, ref_file in (('SSD300', 'SSD300.py'), ('SSD512', 'SSD512.py')):
        tf_shim.install(_vgg())
        tf_shim.S.reset()
        ref = tf_shim.load_reference_ssd300('/root/reference/' + ref_file)
        imgs, gt = R.synthetic_batch(1, 300)

        class It:
            def get_next(self):
                return tf_shim.wrap(imgs.clone()), tf_shim.wrap(gt.clone())
        size = 300 if cls == 'SSD300' else 512
        prov = {'data_shape': [size, size, 3], 'num_train': 1, 'num_val': 0, 'train_generator': (lambda: None, It()), 'val_generator': None}
        getattr(ref, cls)(dict(MG.CONFIG, mode='train', batch_size=1), prov)
        out[cls] = tf_shim.S.optimizer_scope
        print(cls, repr(out[cls]))
    with open(os.path.join(OUT, 'optimizer_scopes.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
