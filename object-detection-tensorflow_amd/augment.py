"""GPU image augmentor: the reference's `utils/image_augmentor.image_augmentor` behind the same arguments.

Reference: /root/reference/utils/image_augmentor.py
  * signature and argument meaning ....... :7-27   (`image_augmentor` below keeps names, order and defaults)
  * argument checks and their messages ... :29-59  (`_check_args`, including the operator-precedence slips of :39,
                                            :49 and :56, which only ever let MORE configurations through)
  * random draws and their order ......... :135-138 crop, :149-150 flips, :174-188 colour, :192 / :236 rotate
  * everything per pixel and per box ..... libodtk (csrc/augment.hip) through odtk_augment_boxes / odtk_augment_images
`Augmentor` is the batch form a loader calls (tfrecord_voc_utils.parse_function :95-112 maps the function over one
example at a time; here a whole batch of decoded pictures of different sizes goes through three launches).

Where this differs from the reference, on purpose (SURVEY.md 8f.2 asks for the decision):
  - with ground truth the reference returns `image_copy`, the UN-augmented picture (:231); the default here returns
    the augmented picture (what :233 returns without ground truth), `image_quirk=True` gives the reference's value;
  - boxes whose centre leaves the picture are dropped from every column (the reference drops them from four of five
    and TensorFlow aborts at :217); if all are lost the picture and boxes fall back to the plain resize (:221-226);
Randomness comes from a numpy Generator (or explicit `draws`), consumed in the reference's order; TF's own random
stream cannot be reproduced, the transforms given the draws are what the tests pin.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AugPlan, call

PI_REF = 3.1415926                        # image_augmentor.py:236
RESIZE_CODE = {'CONSTANT': 0, 'BILINEAR': 1, 'NEAREST_NEIGHBOR': 2, 'BICUBIC': 3}      # odtk_aug_plan.resize; image_augmentor.py:72-76


def _check_args(data_format, fill_mode, zoom_size, output_shape, crop_method, keep_aspect_ratios, constant_values, color_jitter_prob,
                flip_prob, rotate, has_gt):
    if data_format not in ['channels_first', 'channels_last']:
        raise Exception("data_format must in ['channels_first', 'channels_last']!")
    if fill_mode not in ['CONSTANT', 'NEAREST_NEIGHBOR', 'BILINEAR', 'BICUBIC']:
        raise Exception("fill_mode must in ['CONSTANT', 'NEAREST_NEIGHBOR', 'BILINEAR', 'BICUBIC']!")
    if fill_mode == 'CONSTANT' and zoom_size is not None:
        raise Exception("if fill_mode is 'CONSTANT', zoom_size can't be None!")
    if zoom_size is not None:
        if keep_aspect_ratios and constant_values is None:
            raise Exception('please provide constant_values!')
        if not zoom_size[0] >= output_shape[0] and zoom_size[1] >= output_shape[1]:
            raise Exception("output_shape can't greater that zoom_size!")
        if crop_method not in ['random', 'center']:
            raise Exception("crop_method must in ['random', 'center']!")
    if color_jitter_prob is not None and not 0. <= color_jitter_prob <= 1.:
        raise Exception("color_jitter_prob can't less that 0.0, and can't grater that 1.0")
    if flip_prob is not None and (not 0. <= flip_prob[0] <= 1. and 0. <= flip_prob[1] <= 1.):
        raise Exception("flip_prob can't less than 0.0, and can't grater than 1.0")
    if rotate is not None:
        if len(rotate) != 3:
            raise Exception('please provide "rotate" parameter as [rotate_prob, min_angle, max_angle]!')
        if not 0. <= rotate[0] <= 1.:
            raise Exception("rotate prob can't less that 0.0, and can't grater that 1.0")
        if has_gt and (not -5. <= rotate[1] <= 5. and -5. <= rotate[2] <= 5.):
            raise Exception('rotate range must be -5 to 5, otherwise coordinate mapping become imprecise!')
        if not rotate[1] <= rotate[2]:
            raise Exception("rotate[1] can't  grater than rotate[2]")


class _Draws:
    """the reference's tf.random_uniform calls, in its order: scripted (a list) or from a numpy Generator"""

    def __init__(self, scripted=None, rng=None):
        self.q = list(scripted) if scripted is not None else None
        self.rng = rng if rng is not None else np.random.default_rng()

    def uniform(self, lo, hi):
        if self.q is not None:
            return float(self.q.pop(0))
        return float(self.rng.uniform(lo, hi))

    def integer(self, lo, hi):
        """tf.random_uniform([], lo, hi, tf.int32): integers in [lo, hi)"""
        if self.q is not None:
            return int(self.q.pop(0))
        return int(self.rng.integers(lo, hi)) if hi > lo else lo


class Augmentor:
    """One configuration (the `image_augmentor_config` dict of the reference's driver scripts, testSSD300.py:34-46),
    applied to batches: `images` a list of device tensors (u8 or f32, HWC or CHW per `data_format`, any sizes),
    `ground_truths` a list of [G,5] ymin,ymax,xmin,xmax,class tensors (or None).  Returns the f32 batch
    [N,out_h,out_w,C] (or [N,C,out_h,out_w]) and, with ground truth, [N,pad_truth_to,5] = yc,xc,h,w,class padded with -1."""

    def __init__(self, data_format, output_shape, zoom_size=None, crop_method=None, flip_prob=None, fill_mode='BILINEAR',
                 keep_aspect_ratios=False, constant_values=0., color_jitter_prob=None, rotate=None, pad_truth_to=None, seed=None):
        self.cfg = dict(data_format=data_format, output_shape=[int(v) for v in output_shape],
                        zoom_size=None if zoom_size is None else [int(v) for v in zoom_size], crop_method=crop_method,
                        flip_prob=flip_prob, fill_mode=fill_mode, keep_aspect_ratios=keep_aspect_ratios,
                        constant_values=constant_values, color_jitter_prob=color_jitter_prob, rotate=rotate)
        self.pad_truth_to = pad_truth_to
        self.rng = np.random.default_rng(seed)
        self._ws = None

    # ---- host logic: one plan per image (image_augmentor.py:87-146 sizes, :148-197 draws)
    def plan(self, in_h, in_w, draws: _Draws) -> dict:
        c = self.cfg
        f32 = np.float32
        out_h, out_w = c['output_shape']
        zoom_h, zoom_w = c['zoom_size'] if c['zoom_size'] is not None else (out_h, out_w)
        keep = c['keep_aspect_ratios'] or c['fill_mode'] == 'CONSTANT'
        p = dict(in_h=int(in_h), in_w=int(in_w), resize=RESIZE_CODE[c['fill_mode']], crop_h=0, crop_w=0, flip_td=0, flip_lr=0,
                 has_brightness=0, has_contrast=0, has_hue=0, has_rotate=0, brightness=0., contrast=1., hue=0., angle=0.)
        if not keep:
            p.update(resize_h=zoom_h, resize_w=zoom_w, ratio_y=float(f32(zoom_h / in_h)), ratio_x=float(f32(zoom_w / in_w)))
        elif not p['resize']:
            p.update(resize_h=int(in_h), resize_w=int(in_w), ratio_y=1., ratio_x=1.)
        else:
            tall = zoom_h / in_h < zoom_w / in_w
            r = f32(zoom_h / in_h) if tall else f32(zoom_w / in_w)
            p.update(resize_h=zoom_h if tall else int(f32(in_h) * r), resize_w=int(f32(in_w) * r) if tall else zoom_w,
                     ratio_y=float(r), ratio_x=float(r))
        if c['zoom_size'] is not None:
            if c['crop_method'] == 'random':
                p['crop_h'] = draws.integer(0, zoom_h - out_h)
                p['crop_w'] = draws.integer(0, zoom_w - out_w)
            else:
                p['crop_h'], p['crop_w'] = (zoom_h - out_h) // 2, (zoom_w - out_w) // 2
        if c['flip_prob'] is not None:
            td, lr = draws.uniform(0., 1.), draws.uniform(0., 1.)
            p['flip_td'], p['flip_lr'] = int(td < c['flip_prob'][0]), int(lr < c['flip_prob'][1])
        if c['color_jitter_prob'] is not None:
            bcs = [draws.uniform(0., 1.) for _ in range(3)]
            if bcs[0] < c['color_jitter_prob']:
                p['has_brightness'], p['brightness'] = 1, draws.uniform(0., 0.3)
            if bcs[1] < c['color_jitter_prob']:
                p['has_contrast'], p['contrast'] = 1, draws.uniform(0.8, 1.2)
            if bcs[2] < c['color_jitter_prob']:
                p['has_hue'], p['hue'] = 1, draws.uniform(-0.1, 0.1)
        if c['rotate'] is not None:
            if draws.uniform(0., 1.) < c['rotate'][0]:
                p['has_rotate'] = 1
                p['angle'] = float(f32(f32(draws.uniform(c['rotate'][1], c['rotate'][2])) * f32(PI_REF) / f32(180.)))
        return p

    def __call__(self, images, ground_truths=None, draws=None, image_quirk=False):
        c = self.cfg
        _check_args(c['data_format'], c['fill_mode'], c['zoom_size'], c['output_shape'], c['crop_method'], c['keep_aspect_ratios'],
                    c['constant_values'], c['color_jitter_prob'], c['flip_prob'], c['rotate'], ground_truths is not None)
        _lib.load()
        N = len(images)
        assert N > 0 and (draws is None or len(draws) == N)
        chw = c['data_format'] == 'channels_first'
        dev = images[0].device
        out_h, out_w = c['output_shape']
        zoom_h, zoom_w = c['zoom_size'] if c['zoom_size'] is not None else (out_h, out_w)
        plans = (AugPlan * N)()
        keepalive = []
        nch = None
        for n, img in enumerate(images):
            assert img.is_cuda and img.dim() == 3 and img.dtype in (torch.uint8, torch.float32), "pictures: device u8 / f32 rank-3 tensors"
            img = img.contiguous()
            keepalive.append(img)
            ch, h, w = (img.shape[0], img.shape[1], img.shape[2]) if chw else (img.shape[2], img.shape[0], img.shape[1])
            nch = ch if nch is None else nch
            assert ch == nch, "all pictures of a batch need the same channel count"
            p = self.plan(h, w, _Draws(draws[n] if draws is not None else None, self.rng))
            pl = plans[n]
            pl.src, pl.src_u8, pl.src_chw = img.data_ptr(), int(img.dtype == torch.uint8), int(chw)
            for k, v in p.items():
                setattr(pl, k, v)
        if c['color_jitter_prob'] is not None and nch != 3:
            raise Exception('colour jitter needs 3-channel pictures')
        plans_dev = torch.frombuffer(bytearray(bytes(plans)), dtype=torch.uint8).to(dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        gt_out = fallback = None
        if ground_truths is not None:
            if self.pad_truth_to is None:
                raise Exception('ground_truth needs pad_truth_to')          # the reference would return only the picture (:233)
            P = max(1, max(int(g.shape[0]) for g in ground_truths))
            gt_host = torch.zeros(N, P, 5)
            cnt = torch.zeros(N, dtype=torch.int32)
            for n, g in enumerate(ground_truths):
                gt_host[n, :g.shape[0]] = g.detach().float().cpu()
                cnt[n] = g.shape[0]
            gt_in, cnt = gt_host.to(dev), cnt.to(dev)
            gt_out = torch.empty(N, self.pad_truth_to, 5, device=dev)
            fallback = torch.zeros(N, dtype=torch.int32, device=dev)
            call("odtk_augment_boxes", plans_dev.data_ptr(), gt_in.data_ptr(), cnt.data_ptr(), N, P, out_h, out_w, self.pad_truth_to,
                 gt_out.data_ptr(), fallback.data_ptr(), stream)
        need = call_ll("odtk_augment_workspace_bytes", N, nch, out_h, out_w)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((N, nch, out_h, out_w) if chw else (N, out_h, out_w, nch), device=dev)
        call("odtk_augment_images", plans_dev.data_ptr(), None if fallback is None else fallback.data_ptr(), N, nch, zoom_h, zoom_w,
             out_h, out_w, float(c['constant_values'] or 0.), int(chw), out.data_ptr(), self._ws.data_ptr(), stream)
        for t in keepalive + [plans_dev]:
            t.record_stream(torch.cuda.current_stream())
        if ground_truths is None:
            return out
        return (list(images) if image_quirk else out), gt_out


def call_ll(name, *args):
    return int(getattr(_lib.load(), name)(*args))


def image_augmentor(image, input_shape, data_format, output_shape, zoom_size=None, crop_method=None, flip_prob=None,
                    fill_mode='BILINEAR', keep_aspect_ratios=False, constant_values=0., color_jitter_prob=None, rotate=None,
                    ground_truth=None, pad_truth_to=None, draws=None, seed=None, image_quirk=False):
    """The reference's function for ONE picture (same arguments, image_augmentor.py:7-27); `input_shape` is checked
    against the tensor.  Returns the picture, or (picture, ground_truth [pad_truth_to,5]) when ground truth is given."""
    chw = data_format == 'channels_first'
    if data_format in ('channels_first', 'channels_last'):
        h, w = (image.shape[1], image.shape[2]) if chw else (image.shape[0], image.shape[1])
        assert (int(input_shape[0]), int(input_shape[1])) == (h, w), "input_shape does not describe `image`"
    aug = Augmentor(data_format, output_shape, zoom_size, crop_method, flip_prob, fill_mode, keep_aspect_ratios, constant_values,
                    color_jitter_prob, rotate, pad_truth_to, seed)
    if ground_truth is None or pad_truth_to is None:
        # the reference ignores ground truth without pad_truth_to on return (:228-233) -- and cannot rotate without it (:191)
        return aug([image], None, None if draws is None else [draws])[0]
    out, gt = aug([image], [ground_truth], None if draws is None else [draws], image_quirk)
    return out[0], gt[0]
