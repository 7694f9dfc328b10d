"""Synthetic inputs for the tools/ benchmarks (self-contained: nothing under tools/ imports oracle/).
Ground truth rows are [yc, xc, h, w, class] in pixels, padded with -1 (utils/image_augmentor.py:24-27 of the reference)."""
import torch


def synthetic_gt(batch, input_size, seed, pad=60, max_obj=6, lo=0.1, hi=0.8):
    g = torch.Generator().manual_seed(seed)
    gt = torch.full((batch, pad, 5), -1.0)
    for i in range(batch):
        n = int(torch.randint(1, max_obj + 1, (1,), generator=g))
        h = torch.rand(n, generator=g) * (input_size * (hi - lo)) + input_size * lo
        w = torch.rand(n, generator=g) * (input_size * (hi - lo)) + input_size * lo
        yc = h / 2 + torch.rand(n, generator=g) * (input_size - h)
        xc = w / 2 + torch.rand(n, generator=g) * (input_size - w)
        cls = torch.randint(0, 20, (n,), generator=g).float()
        gt[i, :n] = torch.stack([yc, xc, h, w, cls], 1)
    return gt


def pyramid_shapes(input_h, input_w, levels=5):
    """p3.. feature-map sizes of the reference's ResNet + FPN detectors (SAME convs: ceil halving; RetinaNet.py:258-285)."""
    c = lambda v: -(-v // 2)
    h, w = c(c(input_h)), c(c(input_w))
    out = []
    for _ in range(levels):
        h, w = c(h), c(w)
        out.append((h, w))
    return out


RETINA_ANCHOR_SIZES = [32, 64, 128, 256, 512]                 # RetinaNet.py:39-41
RETINA_RATIOS = [1, 1 / 2, 2]
RETINA_SCALES = [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]


def retina_priors_flat():
    flat = []
    for size in RETINA_ANCHOR_SIZES:
        for r in RETINA_RATIOS:
            for s in RETINA_SCALES:
                flat += [s * size * (r ** 0.5), s * size / (r ** 0.5)]
    return flat


YOLO_PRIORS_PX = [[[10., 13.], [16, 30.], [33., 23.]], [[30., 61.], [62., 45.], [59., 119.]], [[116., 90.], [156., 198.], [373., 326.]]]
YOLO_STRIDE = [8., 16., 32.]                                  # testYOLOv3.py:36-38, YOLOv3.py:38


def yolo_priors_flat():
    return [v / YOLO_STRIDE[i] for i in range(3) for hw in YOLO_PRIORS_PX[i] for v in hw]
