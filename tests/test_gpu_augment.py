"""GPU parity of the image augmentor (csrc/augment.hip) through the C-ABI:
  * the golden cases produced by the reference's own image_augmentor with scripted draws (tests/golden/augment.npz);
  * batches of pictures of different sizes / dtypes against oracle/augment_ref.py on the same draws;
  * size-independent properties at the driver scripts' full sizes (flip = mirror bit-exact, rotate keeps boxes on blobs).
Tolerances: boxes 1e-4 px; pixels 2e-3 on the 0..255 scale (bilinear / bicubic / colour arithmetic order), 2e-2 for rotated noise;
nearest-neighbour pictures without colour jitter are copies of source pixels and compared exactly."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment_ref as AR  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _aug():
    import odtk  # noqa: F401
    from odtk import augment
    return augment


def _golden(fname='augment.npz'):
    g = np.load(os.path.join(GOLD, fname))
    return g, json.loads(bytes(g['meta']).decode())


@pytest.mark.parametrize('fname', ['augment.npz', 'augment_zoom_methods.npz'], ids=['driver-modes', 'nearest-bicubic'])
@pytest.mark.parametrize('u8', [False, True], ids=['f32-src', 'u8-src'])
def test_golden_cases_vs_reference(u8, fname):
    A = _aug()
    dev = torch.device('cuda:0')
    g, meta = _golden(fname)
    for m in meta:
        n = m['name']
        src = torch.from_numpy(g[f'{n}_image'])
        img = (src if u8 else src.float()).to(dev)
        h, w = m['hw']
        out, gt = A.image_augmentor(img, [h, w, 3], m['data_format'], ground_truth=torch.from_numpy(g[f'{n}_gt_in']).to(dev),
                                    pad_truth_to=6, draws=m['draws'], **m['kwargs'])
        torch.cuda.synchronize()
        np.testing.assert_allclose(gt.cpu().numpy(), g[f'{n}_gt_out'], rtol=0, atol=1e-4, err_msg=n)
        want = g[f'{n}_aug']
        if m['kwargs'].get('rotate') is not None and m['data_format'] == 'channels_first':
            want = want.transpose(2, 0, 1)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-3, err_msg=n)
        if m['kwargs']['fill_mode'] == 'NEAREST_NEIGHBOR' and m['kwargs'].get('color_jitter_prob') is None:
            assert np.array_equal(out.cpu().numpy(), want), n      # source pixels (and the pad constant) copied, nothing computed
        only = A.image_augmentor(img, [h, w, 3], m['data_format'], draws=m['draws'], **{k: v for k, v in m['kwargs'].items() if k != 'rotate'}) \
            if m['kwargs'].get('rotate') is None else None
        if only is not None:
            assert torch.equal(only, out)                       # with and without ground truth: the same picture
        quirk, _ = A.image_augmentor(img, [h, w, 3], m['data_format'], ground_truth=torch.from_numpy(g[f'{n}_gt_in']).to(dev),
                                     pad_truth_to=6, draws=m['draws'], image_quirk=True, **m['kwargs'])
        assert quirk is img                                     # image_augmentor.py:231


CONFIGS = [
    dict(output_shape=[300, 300], crop_method='random', flip_prob=[0., 0.5], fill_mode='BILINEAR', keep_aspect_ratios=False,
         constant_values=0., color_jitter_prob=0.5, rotate=[0.5, -5., -5.]),                                 # testSSD300.py:34-46
    dict(output_shape=[64, 96], zoom_size=[80, 120], crop_method='random', flip_prob=[0.5, 0.5], fill_mode='BILINEAR',
         keep_aspect_ratios=True, constant_values=114., color_jitter_prob=0.7, rotate=[0.6, -5., 5.]),
    dict(output_shape=[48, 48], zoom_size=[56, 60], crop_method='center', fill_mode='BILINEAR'),
    dict(output_shape=[72, 80], fill_mode='CONSTANT', constant_values=3., flip_prob=[0.5, 0.5]),               # pad only
    dict(output_shape=[64, 96], zoom_size=[80, 120], crop_method='random', flip_prob=[0.5, 0.5], fill_mode='NEAREST_NEIGHBOR',
         keep_aspect_ratios=True, constant_values=114., color_jitter_prob=0.7),
    dict(output_shape=[300, 300], zoom_size=[330, 340], crop_method='random', flip_prob=[0.5, 0.5], fill_mode='BICUBIC',
         keep_aspect_ratios=False, color_jitter_prob=0.5),
]


def _random_batch(seed, n, lo, hi, max_hw=None):
    g = torch.Generator().manual_seed(seed)
    imgs, gts = [], []
    for i in range(n):
        h = int(torch.randint(lo, hi, (1,), generator=g)); w = int(torch.randint(lo, hi, (1,), generator=g))
        if max_hw is not None:
            h, w = min(h, max_hw[0]), min(w, max_hw[1])
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        k = int(torch.randint(1, 7, (1,), generator=g))
        yc, xc = torch.rand(k, generator=g) * h, torch.rand(k, generator=g) * w            # some centres near the border: lost
        bh, bw = (0.05 + 0.4 * torch.rand(k, generator=g)) * h, (0.05 + 0.4 * torch.rand(k, generator=g)) * w
        gt = torch.stack([(yc - bh / 2).clamp(min=0), (yc + bh / 2).clamp(max=h - 1.), (xc - bw / 2).clamp(min=0),
                          (xc + bw / 2).clamp(max=w - 1.), torch.randint(0, 20, (k,), generator=g).float()], -1)
        imgs.append(img); gts.append(gt)
    return imgs, gts


@pytest.mark.parametrize('ci', range(len(CONFIGS)))
def test_batches_vs_oracle(ci):
    A = _aug()
    dev = torch.device('cuda:0')
    cfg = CONFIGS[ci]
    pad_only = cfg['fill_mode'] == 'CONSTANT'
    imgs, gts = _random_batch(100 + ci, 9, 40, 140, max_hw=(72, 80) if pad_only else None)
    gts[3] = torch.tensor([[0., 0., 0., 0., 4.]])          # its centre sits on the border under every transform: all lost -> fallback
    gts[5] = torch.cat([gts[5][:1], torch.tensor([[0., 0., 0., 0., 9.]]), gts[5][1:]])   # one lost among others
    aug = A.Augmentor('channels_last', pad_truth_to=8, seed=ci, **cfg)
    rng = np.random.default_rng(ci)
    draws = []
    for img in imgs:                               # record what a numpy stream would draw, then replay it on both sides
        rec = _Recorder(rng)
        aug.plan(img.shape[0], img.shape[1], rec)
        draws.append(rec.log)
    mixed = [im.to(dev) if i % 2 else im.float().to(dev) for i, im in enumerate(imgs)]
    out, gt = aug(mixed, [t.to(dev) for t in gts], draws=draws)
    torch.cuda.synchronize()
    lost = 0
    for i, (im, t) in enumerate(zip(imgs, gts)):
        want_img, want_gt = AR.image_augmentor(im.float(), list(im.shape), 'channels_last', ground_truth=t, pad_truth_to=8, draws=draws[i], **cfg)
        np.testing.assert_allclose(gt[i].cpu().numpy(), want_gt.numpy(), rtol=0, atol=1e-4, err_msg=f'boxes of image {i}')
        # rotated pictures: the f32 source coordinate of a 300-px picture carries ~3e-5 px of rounding, times up to 255 / px of
        # gradient in these noise pictures
        tol = 2e-2 if cfg.get('rotate') is not None else 2e-3
        np.testing.assert_allclose(out[i].cpu().numpy(), want_img.numpy(), rtol=0, atol=tol, err_msg=f'image {i}')
        lost += int((want_gt[:, 4] >= 0).sum()) < t.shape[0]
    assert lost >= 1                               # the situation the reference aborts in (:217) is exercised


class _Recorder:
    def __init__(self, rng):
        self.rng, self.log = rng, []

    def uniform(self, lo, hi):
        self.log.append(float(self.rng.uniform(lo, hi)))
        return self.log[-1]

    def integer(self, lo, hi):
        self.log.append(int(self.rng.integers(lo, hi)) if hi > lo else lo)
        return self.log[-1]


def test_flip_is_exact_mirror_full_size():
    """batch 32 of VOC-sized pictures to 300 x 300 (testSSD300.py:34-46 without the random colour / rotate part):
    the flipped output is the bit-exact mirror of the unflipped one, boxes mirror as out - x - 1"""
    A = _aug()
    dev = torch.device('cuda:0')
    imgs, gts = _random_batch(7, 32, 330, 500)
    imgs = [im.to(dev) for im in imgs]
    gts = [torch.stack([t[:, 0] * .5 + 40, t[:, 1] * .5 + 60, t[:, 2] * .5 + 40, t[:, 3] * .5 + 60, t[:, 4]], -1).to(dev) for t in gts]
    aug = A.Augmentor('channels_last', [300, 300], flip_prob=[0.5, 0.5], pad_truth_to=60)
    a, ga = aug(imgs, gts, draws=[[0.9, 0.9]] * 32)
    b, gb = aug(imgs, gts, draws=[[0.1, 0.1]] * 32)
    torch.cuda.synchronize()
    assert torch.equal(torch.flip(a, [1, 2]), b)
    valid = ga[..., 4] >= 0
    assert torch.equal(valid, gb[..., 4] >= 0) and int(valid.sum()) == sum(t.shape[0] for t in gts)
    torch.testing.assert_close(gb[..., 0][valid], 299. - ga[..., 0][valid], atol=1e-4, rtol=0)
    torch.testing.assert_close(gb[..., 1][valid], 299. - ga[..., 1][valid], atol=1e-4, rtol=0)
    torch.testing.assert_close(gb[..., 2:4][valid], ga[..., 2:4][valid], atol=1e-4, rtol=0)
    assert (ga[~valid] == -1).all()


def test_rotate_keeps_boxes_on_their_blobs():
    """a bright rectangle and its box go through zoom + crop + flip + rotate: the rotated box contains the blob and is tight
    to ~1.5 px + the growth of an axis-aligned box under a 5 degree turn"""
    A = _aug()
    dev = torch.device('cuda:0')
    img = torch.zeros(240, 320, 3)
    y0, y1, x0, x1 = 60, 150, 90, 250
    img[y0:y1 + 1, x0:x1 + 1] = 200.
    gt = torch.tensor([[float(y0), float(y1), float(x0), float(x1), 5.]])
    for ang in (-5., 5.):
        out, g = A.image_augmentor(img.to(dev), [240, 320, 3], 'channels_last', [200, 200], zoom_size=[220, 230], crop_method='random',
                                   flip_prob=[0.5, 0.5], rotate=[1., -5., 5.], ground_truth=gt.to(dev), pad_truth_to=3,
                                   draws=[7, 11, 0.1, 0.8, 0.0, ang])
        torch.cuda.synchronize()
        ys, xs = torch.nonzero(out[..., 0].cpu() > 100., as_tuple=True)
        yc, xc, h, w = g[0, :4].cpu().tolist()
        by0, by1, bx0, bx1 = yc - h / 2, yc + h / 2, xc - w / 2, xc + w / 2
        assert by0 - 1.5 <= ys.min() and ys.max() <= by1 + 1.5 and bx0 - 1.5 <= xs.min() and xs.max() <= bx1 + 1.5
        assert ys.min() - by0 < 3. and by1 - ys.max() < 3. and xs.min() - bx0 < 3. and bx1 - xs.max() < 3.


def test_argument_errors_match_reference_messages():
    A = _aug()
    img = torch.zeros(20, 30, 3, device='cuda:0')
    with pytest.raises(Exception, match="data_format must in"):
        A.image_augmentor(img, [20, 30, 3], 'NHWC', [10, 10])
    with pytest.raises(Exception, match="crop_method must in"):
        A.image_augmentor(img, [20, 30, 3], 'channels_last', [10, 10], zoom_size=[12, 12], crop_method='corner')
    with pytest.raises(Exception, match="rotate\\[1\\] can't  grater than rotate\\[2\\]"):
        A.image_augmentor(img, [20, 30, 3], 'channels_last', [10, 10], rotate=[.5, 3., -3.])
    with pytest.raises(Exception, match="fill_mode must in"):
        A.image_augmentor(img, [20, 30, 3], 'channels_last', [10, 10], fill_mode='LANCZOS')
