"""CPU fp32 restatement of the reference's SSD512 (TEST INFRASTRUCTURE ONLY): oracle/ssd300_ref.py run with the 512 x 512 variant's tables.

/root/reference/SSD512.py differs from SSD300.py only in: input size 512 (:17-21); conv12_1 / conv12_2 and a seventh head (:320-322, :91);
six anchors on pred5 (:89, :123); the scale list 0.07, 0.15 ... 0.9 (:116-118) -> 24 912 priors.  Every function of ssd300_ref is re-exported
here bound to those tables (the module's table globals are swapped for the duration of a call).
Pinned against the reference's own SSD512 class run on oracle/tf_shim: tests/golden/ssd512.npz (tests/golden/make_golden_ssd512.py).
"""
from __future__ import annotations

import contextlib
import functools

from . import ssd300_ref as R

INPUT_SIZE = 512
EXTRA_LAYERS = list(R.EXTRA_LAYERS) + [("conv12_1", 256, 128, 1, 1, 1), ("conv12_2", 128, 256, 3, 2, 1)]
FEATS = list(R.FEATS) + ["conv12_2"]
FEAT_CH = [512, 1024, 512, 256, 256, 256, 256]
ANCHORS_PER_CELL = [4, 6, 6, 6, 6, 4, 4]
ASPECTS = [[2, 1 / 2], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2], [2, 1 / 2]]


def prior_scales():
    """SSD512.py:116-118"""
    s = [0.07 * INPUT_SIZE]
    s = s + [(0.15 + (0.9 - 0.15) / 5 * (i - 1)) * INPUT_SIZE for i in range(1, 8)]
    return [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 7)]


_TABLES = dict(INPUT_SIZE=INPUT_SIZE, EXTRA_LAYERS=EXTRA_LAYERS, FEATS=FEATS, FEAT_CH=FEAT_CH, ANCHORS_PER_CELL=ANCHORS_PER_CELL, ASPECTS=ASPECTS,
               prior_scales=prior_scales)


@contextlib.contextmanager
def tables():
    old = {k: getattr(R, k) for k in _TABLES}
    try:
        for k, v in _TABLES.items():
            setattr(R, k, v)
        yield
    finally:
        for k, v in old.items():
            setattr(R, k, v)


def _bound(fn):
    @functools.wraps(fn)
    def f(*a, **k):
        with tables():
            return fn(*a, **k)
    return f


conv_specs, init_params, calibrate_bn, forward, feature_sizes, priors = (_bound(f) for f in (R.conv_specs, R.init_params, R.calibrate_bn, R.forward,
                                                                                             R.feature_sizes, R.priors))
batch_loss, train_step, test_one_image, synthetic_batch = (_bound(f) for f in (R.batch_loss, R.train_step, R.test_one_image, R.synthetic_batch))
trainable_names, one_image_loss, match, nms, decode, detect = R.trainable_names, R.one_image_loss, R.match, R.nms, R.decode, R.detect
BN_EPS, MEAN_RGB = R.BN_EPS, R.MEAN_RGB
