"""SSD300 (VGG-16) behind the reference class surface -- see SSD300 below."""
from __future__ import annotations

INPUT_SIZE = 300
FEATURE_SIZES = [38, 19, 10, 5, 5, 3]           # conv10_2 has stride 1 (reference SSD300.py:311)
ANCHORS_PER_CELL = [4, 6, 6, 6, 4, 4]
ASPECTS = [[2, 1 / 2], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2, 3, 1 / 3], [2, 1 / 2], [2, 1 / 2]]


def prior_spec(input_size=INPUT_SIZE):
    """Host part of SSD300._get_abbox (reference SSD300.py:112-119, 333-336): the python-double
    (h, w) list per level, flattened; the per-cell arithmetic runs in odtk_ssd_priors."""
    s = [(0.2 + (0.9 - 0.2) / 5 * (i - 1)) * input_size for i in range(1, 8)]
    s = [[s[i], (s[i] * s[i + 1]) ** 0.5] for i in range(0, 6)]
    flat = []
    for size, ar in zip(s, ASPECTS):
        pr = [[size[0], size[0]], [size[1], size[1]]]
        for a in ar:
            pr.append([size[0] * (a ** 0.5), size[0] / (a ** 0.5)])
        for h, w in pr:
            flat += [h, w]
    return FEATURE_SIZES, ANCHORS_PER_CELL, flat
