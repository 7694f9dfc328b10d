#!/usr/bin/env python
"""Generates tests/golden/ssd300_variables.json: name, shape and dtype of every variable the REFERENCE's SSD300 graph
creates (SSD300.py:43, :77, :193-313, :85-90), collected from the eager TF-1.x shim after constructing the reference
class in train mode.  It pins odtk.ssd300.reference_variable_map (the names a tf.train.Saver checkpoint of the
reference holds) on the reference's own code; the tf.layers default-name counting is the shim's restatement of TF.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_variables.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ssd300_ref as R          # noqa: E402
from oracle import tf_shim                  # noqa: E402
import make_golden as MG                    # noqa: E402


def main():
    p = R.init_params(7)
    tf_shim.install(MG.vgg_tensors(p))
    ref = tf_shim.load_reference_ssd300()

    class _It:
        def get_next(self):
            im, g = R.synthetic_batch(2, 100)
            return tf_shim.wrap(im), tf_shim.wrap(g)
    prov = {'data_shape': [300, 300, 3], 'num_train': 2, 'num_val': 0, 'train_generator': (lambda: None, _It()), 'val_generator': None}
    ref.SSD300(dict(MG.CONFIG, mode='train'), prov)
    out = {n: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''), trainable=n in tf_shim.S.trainable)
           for n, v in tf_shim.S.variables.items()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ssd300_variables.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out), 'variables;', sum(v['trainable'] for v in out.values()), 'trainable')
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
