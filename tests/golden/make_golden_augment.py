#!/usr/bin/env python
"""Generates tests/golden/augment.npz by calling the REFERENCE's own utils/image_augmentor.image_augmentor on the
eager TF-1.x shim (oracle/tf_shim), with tf.random_uniform scripted (tf_shim.RANDOM_QUEUE) so that every case is
reproducible.  Each case is run twice, as the reference allows: with ground truth (returns the boxes) and without
(returns the augmented image, image_augmentor.py:233).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_augment.py [all|base|methods]
('methods' writes tests/golden/augment_zoom_methods.npz: the NEAREST_NEIGHBOR / BICUBIC fill modes, added in round 3)
tests/test_oracle_golden.py checks oracle/augment_ref.py against the fixture; tests/test_gpu_augment.py the kernels.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tf_shim                 # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (name, input hw, kwargs, draws) -- draws in the reference's order (see oracle/augment_ref.py)
CASES = [
    ('ssd_driver', (37, 50), dict(output_shape=[30, 30], crop_method='random', flip_prob=[0., 0.5], fill_mode='BILINEAR',
                                  keep_aspect_ratios=False, constant_values=0., color_jitter_prob=0.5, rotate=[0.5, -5., -5.]),
     [0.3, 0.2,  0.1, 0.9, 0.4, 0.12, -0.07,  0.2, -5.0]),          # flips | bcs, brightness, hue | rotate p, angle
    ('zoom_crop_flip', (40, 50), dict(output_shape=[32, 32], zoom_size=[40, 44], crop_method='random', flip_prob=[0.5, 0.5],
                                      fill_mode='BILINEAR', keep_aspect_ratios=False),
     [4, 6, 0.2, 0.9]),
    ('zoom_center_both_flips', (33, 47), dict(output_shape=[24, 28], zoom_size=[30, 36], crop_method='center', flip_prob=[1., 1.],
                                              fill_mode='BILINEAR', keep_aspect_ratios=False),
     [0.5, 0.5]),
    ('keep_aspect_h', (60, 40), dict(output_shape=[32, 32], zoom_size=[36, 40], crop_method='random', fill_mode='BILINEAR',
                                     keep_aspect_ratios=True, constant_values=127.),
     [2, 3]),
    ('keep_aspect_w', (30, 64), dict(output_shape=[32, 32], zoom_size=[36, 40], crop_method='center', fill_mode='BILINEAR',
                                     keep_aspect_ratios=True, constant_values=5.),
     []),
    ('contrast_rotate', (28, 28), dict(output_shape=[28, 28], fill_mode='BILINEAR', color_jitter_prob=1.0, rotate=[1.0, -5., 5.]),
     [0.0, 0.0, 0.0, 0.25, 1.15, 0.05,  0.0, 3.5]),
    ('channels_first', (20, 26), dict(output_shape=[16, 16], zoom_size=[18, 20], crop_method='random', flip_prob=[0.5, 0.5],
                                      fill_mode='BILINEAR', data_format='channels_first'),
     [1, 2, 0.7, 0.1]),
]


# the two fill modes no driver script uses (image_augmentor.py:72-76): tf.image.resize_images with NEAREST_NEIGHBOR / BICUBIC,
# align_corners=True, through the shim's restatement of the TF 1.13 kernels -> tests/golden/augment_zoom_methods.npz
CASES_METHODS = [
    ('nearest_zoom_crop_flip', (40, 50), dict(output_shape=[32, 32], zoom_size=[40, 44], crop_method='random', flip_prob=[0.5, 0.5],
                                              fill_mode='NEAREST_NEIGHBOR', keep_aspect_ratios=False),
     [4, 6, 0.2, 0.9]),
    ('nearest_up_keep_aspect', (21, 17), dict(output_shape=[40, 40], zoom_size=[48, 44], crop_method='center', fill_mode='NEAREST_NEIGHBOR',
                                              keep_aspect_ratios=True, constant_values=9.),
     []),
    ('nearest_plain', (37, 50), dict(output_shape=[30, 30], fill_mode='NEAREST_NEIGHBOR'), []),
    ('bicubic_zoom_crop_flip', (40, 50), dict(output_shape=[32, 32], zoom_size=[40, 44], crop_method='random', flip_prob=[0.5, 0.5],
                                              fill_mode='BICUBIC', keep_aspect_ratios=False),
     [3, 5, 0.9, 0.2]),
    ('bicubic_up_keep_aspect', (21, 17), dict(output_shape=[40, 40], zoom_size=[48, 44], crop_method='random', fill_mode='BICUBIC',
                                              keep_aspect_ratios=True, constant_values=127.),
     [5, 1]),
    ('bicubic_down_colour', (61, 47), dict(output_shape=[28, 24], fill_mode='BICUBIC', color_jitter_prob=1.0),
     [0.0, 0.0, 0.0, 0.2, 0.9, -0.05]),
    ('bicubic_channels_first', (20, 26), dict(output_shape=[16, 16], zoom_size=[18, 20], crop_method='random', flip_prob=[0.5, 0.5],
                                              fill_mode='BICUBIC', data_format='channels_first'),
     [1, 2, 0.7, 0.1]),
]


def boxes_for(h, w, g, n):
    """n boxes well inside the image so that no centre is lost (the only inputs the reference survives, :217)"""
    yc = (0.35 + 0.3 * torch.rand(n, generator=g)) * h
    xc = (0.35 + 0.3 * torch.rand(n, generator=g)) * w
    bh = (0.1 + 0.3 * torch.rand(n, generator=g)) * h
    bw = (0.1 + 0.3 * torch.rand(n, generator=g)) * w
    cls = torch.randint(0, 20, (n,), generator=g).float()
    return torch.stack([yc - bh / 2, yc + bh / 2, xc - bw / 2, xc + bw / 2, cls], -1)


def run_cases(ref, cases, seed, fname):
    g = torch.Generator().manual_seed(seed)
    out = {}
    meta = []
    for name, (h, w), kw, draws in cases:
        kw = dict(kw)
        fmt = kw.pop('data_format', 'channels_last')
        img = (torch.rand(h, w, 3, generator=g) * 255).round()
        src = img.permute(2, 0, 1).contiguous() if fmt == 'channels_first' else img
        gt = boxes_for(h, w, g, 3)
        tf_shim.RANDOM_QUEUE[:] = list(draws)
        ret_img, ret_gt = ref.image_augmentor(src, [h, w, 3], fmt, ground_truth=gt, pad_truth_to=6, **kw)
        assert not tf_shim.RANDOM_QUEUE, (name, tf_shim.RANDOM_QUEUE)
        assert ret_img is src                               # the reference hands back image_copy (:231)
        if kw.get('rotate') is not None:
            # the reference cannot rotate without ground truth (NameError on ymin, :191) and with it returns image_copy;
            # the rotated image is the last thing it computes, so take it from the shim's rotate trace
            aug = tf_shim.ROTATE_TRACE[-1]
        else:
            tf_shim.RANDOM_QUEUE[:] = list(draws)
            aug = ref.image_augmentor(src, [h, w, 3], fmt, **kw)
            assert not tf_shim.RANDOM_QUEUE
        out[f'{name}_image'] = src.numpy().astype(np.uint8)
        out[f'{name}_gt_in'] = gt.numpy()
        out[f'{name}_gt_out'] = ret_gt.numpy()
        out[f'{name}_aug'] = aug.numpy().astype(np.float32)
        meta.append(dict(name=name, hw=[h, w], data_format=fmt, kwargs=kw, draws=list(draws)))
        print(name, tuple(aug.shape), ret_gt[:3].tolist())
    out['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, fname), **out)


def main():
    tf_shim.install()
    ref = tf_shim.load_reference_module('/root/reference/utils/image_augmentor.py', 'reference_image_augmentor')
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'base'):
        run_cases(ref, CASES, 77, 'augment.npz')
    if which in ('all', 'methods'):
        run_cases(ref, CASES_METHODS, 78, 'augment_zoom_methods.npz')
    tf_shim.uninstall()


if __name__ == '__main__':
    main()
