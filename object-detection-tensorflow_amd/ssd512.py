"""SSD512 (VGG-16) behind the reference class surface: the SSD300 class (ssd300.py) with the 512 x 512 variant's tables.

Reference: /root/reference/SSD512.py -- identical to SSD300.py except for
  * the input size 512 (:17-21), hence feature maps 64 / 32 / 16 / 8 / 8 / 4 / 2 (conv10_2 keeps stride 1 as in SSD300, :311-313),
  * one more extra block conv12_1 (1x1, 128) / conv12_2 (3x3, stride 2, 256) and a seventh head pred7 (:320-322, :91),
  * six anchors on pred5 (aspects 2, 1/2, 3, 1/3; :89, :123),
  * the scale list 0.07, then 0.15 ... 0.9 in five steps, times the input size (:116-118)
-> 24 912 priors.  Every kernel is the SSD300 path's; the prior matching (one workgroup per image) and the hard-negative-mining NMS
take up to 32 768 priors per image (the NMS's rarely used whole-problem fallback sorts through global memory above 16 384).
Driver: testSSD512.py (same keys as testSSD300.py).
"""
from __future__ import annotations

from .ssd300 import EXTRA_SEQ as _EXTRA300
from .ssd300 import SSD300, reference_variable_map as _map300

INPUT_SIZE = 512
FEATURE_SIZES = [64, 32, 16, 8, 8, 4, 2]
ANCHORS_PER_CELL = [4, 6, 6, 6, 6, 4, 4]
ASPECTS = [[2, 1 / 2], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2], [2, 1 / 2]]
EXTRA_SEQ = list(_EXTRA300) + [("conv12_1", 256, 128, 1, 1, 1), ("conv12_2", 128, 256, 3, 2, 1)]     # SSD512.py:320-321
FEAT_SRC = ["feat1", "conv7", "conv8_2", "conv9_2", "conv10_2", "conv11_2", "conv12_2"]               # :322
NUM_PRIORS = sum(f * f * a for f, a in zip(FEATURE_SIZES, ANCHORS_PER_CELL))                          # 24912


def prior_scales(input_size=INPUT_SIZE):
    """SSD512.py:116-118"""
    s = [0.07 * input_size]
    s = s + [(0.15 + (0.9 - 0.15) / 5 * (i - 1)) * input_size for i in range(1, 8)]
    return [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 7)]


def reference_variable_map():
    return _map300(EXTRA_SEQ, 7)


class SSD512(SSD300):
    INPUT_SIZE = INPUT_SIZE
    FEATURE_SIZES = FEATURE_SIZES
    ANCHORS_PER_CELL = ANCHORS_PER_CELL
    ASPECTS = ASPECTS
    EXTRA_SEQ = EXTRA_SEQ
    FEAT_SRC = FEAT_SRC
    FEAT_CH = [512, 1024, 512, 256, 256, 256, 256]

    @classmethod
    def prior_scales(cls):
        return prior_scales(cls.INPUT_SIZE)
