"""CPU fp32 restatement of the reference's image augmentor (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/utils/image_augmentor.py:
  * argument checks ...................... :29-59
  * zoom / keep-aspect resize + pad ...... :87-129   (bilinear / nearest / bicubic per fill_mode :72-76, align_corners=True;
                                           boxes scaled by the same ratios)
  * crop (only with zoom_size) ........... :131-146
  * flips ................................ :148-172   (box: max' = out - min - 1, min' = out - max - 1)
  * colour jitter ........................ :173-188   (brightness +U(0,.3), contrast U(.8,1.2), hue U(-.1,.1))
  * rotate ............................... :190-197, rotate_helper :235-260  (image by +ang, box corners by -ang about
                                           ((w-1)/2, (h-1)/2), new box = min/max of the 4 corners)
  * clip, centre filter, [yc,xc,h,w,cls] . :201-219
  * all-boxes-lost fallback .............. :221-226, gt_checker_helper :263-267
  * pad to pad_truth_to with -1 .......... :228-232
Randomness: the reference draws with tf.random_uniform in this ORDER (a draw inside a tf.cond branch only happens when
the branch is taken): crop_h, crop_w | flip_td, flip_lr | bcs[3], brightness delta, contrast factor, hue delta |
rotate p, angle.  `draws` is that sequence, supplied by the caller (TF's generator cannot be matched, so parity
is checked with scripted draws, see tests/golden/make_golden_augment.py).

Reference behaviour that is NOT reproduced, on purpose (SURVEY.md section 8f item 2 leaves the decision open):
  - with ground_truth and pad_truth_to the reference returns `image_copy`, the UN-augmented input image (:231), next to
    the augmented boxes.  restated here as `image_quirk=True`; the default returns the augmented image, which is also what
    the reference itself returns when ground_truth is None (:233);
  - when SOME (not all) boxes lose their centre, :217 concatenates the unfiltered centres [G,1] with the filtered
    sizes [G',1] and TensorFlow aborts with a shape error; here the centres are filtered too (the evident intent).
    So is the all-lost fallback (:221-226), which the reference can never reach for the same reason.
The TF image ops are restated from their kernels: resize_bilinear (align_corners), adjust_contrastv2 (per-channel
mean), adjust_hue (RGB -> hue/min/max -> RGB), contrib.image.rotate (projective transform, bilinear, zero outside).
The colour and rotate IMAGE ops are "parity unpinned" (no TensorFlow here to produce vectors); the geometry and every
box computation are pinned against the reference's own code run on oracle/tf_shim (tests/golden/augment.npz).
Only tests/ and the smoke/bench checkers may import this file.
"""
from __future__ import annotations

import math

import numpy as np
import torch

PI_REF = 3.1415926                                   # image_augmentor.py:236


def check_args(data_format, fill_mode, zoom_size, output_shape, crop_method, keep_aspect_ratios, constant_values,
               color_jitter_prob, flip_prob, rotate, ground_truth):
    """the reference's argument checks with its messages (:29-59), including the two it gets wrong on purpose-or-not:
    'CONSTANT' with a zoom_size raises, and the flip / zoom range checks bind as `(not a) and b`."""
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
    if flip_prob is not None:
        if not 0. <= flip_prob[0] <= 1. and 0. <= flip_prob[1] <= 1.:
            raise Exception("flip_prob can't less than 0.0, and can't grater than 1.0")
    if rotate is not None:
        if len(rotate) != 3:
            raise Exception('please provide "rotate" parameter as [rotate_prob, min_angle, max_angle]!')
        if not 0. <= rotate[0] <= 1.:
            raise Exception("rotate prob can't less that 0.0, and can't grater that 1.0")
        if ground_truth is not None:
            if not -5. <= rotate[1] <= 5. and -5. <= rotate[2] <= 5.:
                raise Exception('rotate range must be -5 to 5, otherwise coordinate mapping become imprecise!')
        if not rotate[1] <= rotate[2]:
            raise Exception("rotate[1] can't  grater than rotate[2]")


def plan(input_shape, output_shape, zoom_size, crop_method, flip_prob, fill_mode, keep_aspect_ratios,
         color_jitter_prob, rotate, draws):
    """consume `draws` in the reference's order and return the per-image plan (all plain numbers)."""
    f32 = np.float32
    d = list(draws)
    in_h, in_w = int(input_shape[0]), int(input_shape[1])
    out_h, out_w = int(output_shape[0]), int(output_shape[1])
    zh, zw = (int(zoom_size[0]), int(zoom_size[1])) if zoom_size is not None else (out_h, out_w)
    if fill_mode == 'CONSTANT':
        keep_aspect_ratios = True
    p = dict(in_h=in_h, in_w=in_w, out_h=out_h, out_w=out_w, zoom_h=zh, zoom_w=zw, resize=fill_mode != 'CONSTANT', method=fill_mode)
    if keep_aspect_ratios:
        if fill_mode != 'CONSTANT':
            if zh / in_h < zw / in_w:
                r = f32(zh / in_h)
                p['resize_h'], p['resize_w'] = zh, int(f32(in_w) * r)
            else:
                r = f32(zw / in_w)
                p['resize_h'], p['resize_w'] = int(f32(in_h) * r), zw
            p['ratio_y'] = p['ratio_x'] = float(r)
        else:                                           # pad only, boxes untouched (:119-123)
            p['resize_h'], p['resize_w'], p['ratio_y'], p['ratio_x'] = in_h, in_w, 1.0, 1.0
    else:
        p['resize_h'], p['resize_w'] = zh, zw
        p['ratio_y'], p['ratio_x'] = float(f32(zh / in_h)), float(f32(zw / in_w))
    p['crop_h'] = p['crop_w'] = 0
    if zoom_size is not None:
        if crop_method == 'random':
            p['crop_h'], p['crop_w'] = int(d.pop(0)), int(d.pop(0))
        else:
            p['crop_h'], p['crop_w'] = (zh - out_h) // 2, (zw - out_w) // 2
    p['flip_td'] = p['flip_lr'] = False
    if flip_prob is not None:
        a, b = float(d.pop(0)), float(d.pop(0))
        p['flip_td'], p['flip_lr'] = a < flip_prob[0], b < flip_prob[1]
    p['brightness'] = p['contrast'] = p['hue'] = None
    if color_jitter_prob is not None:
        bcs = [float(d.pop(0)) for _ in range(3)]
        if bcs[0] < color_jitter_prob:
            p['brightness'] = float(d.pop(0))
        if bcs[1] < color_jitter_prob:
            p['contrast'] = float(d.pop(0))
        if bcs[2] < color_jitter_prob:
            p['hue'] = float(d.pop(0))
    p['angle'] = None
    if rotate is not None:
        if float(d.pop(0)) < rotate[0]:
            p['angle'] = float(f32(f32(d.pop(0)) * f32(PI_REF) / f32(180.)))
    p['unused_draws'] = d
    return p


def resize_bilinear_align(img, oh, ow):
    """TF ResizeBilinear(align_corners=True) on HWC f32: scale = (in-1)/(out-1), top + (bottom - top) * lerp."""
    H, W, _ = img.shape
    sy = np.float32((H - 1) / (oh - 1)) if oh > 1 else np.float32(0)
    sx = np.float32((W - 1) / (ow - 1)) if ow > 1 else np.float32(0)
    iy = (np.arange(oh, dtype=np.float32) * sy).astype(np.float32)
    ix = (np.arange(ow, dtype=np.float32) * sx).astype(np.float32)
    y0 = np.floor(iy).astype(np.int64); y1 = np.minimum(np.ceil(iy).astype(np.int64), H - 1)
    x0 = np.floor(ix).astype(np.int64); x1 = np.minimum(np.ceil(ix).astype(np.int64), W - 1)
    ly = torch.from_numpy(iy - y0.astype(np.float32)).view(oh, 1, 1)
    lx = torch.from_numpy(ix - x0.astype(np.float32)).view(1, ow, 1)
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return top + (bot - top) * ly


def _resize_scale(n_in, n_out, align_corners=True):
    """CalculateResizeScale (tensorflow/core/kernels/image_resizer_state.h)"""
    return np.float32((n_in - 1) / np.float32(n_out - 1)) if (align_corners and n_out > 1) else np.float32(n_in / np.float32(n_out))


def resize_nearest_align(img, oh, ow, align_corners=True):
    """TF 1.13 ResizeNearestNeighbor (resize_nearest_neighbor_op.cc): in = min(roundf(out * scale), in_size - 1) with
    align_corners (floorf without); image_augmentor.py:72,98-101 / :114-117 call it with align_corners=True."""
    H, W, _ = img.shape

    def src(n_in, n_out):
        pos = np.arange(n_out, dtype=np.float32) * _resize_scale(n_in, n_out, align_corners)
        idx = np.where(pos - np.floor(pos) >= np.float32(0.5), np.floor(pos) + 1, np.floor(pos)) if align_corners else np.floor(pos)
        return np.minimum(idx.astype(np.int64), n_in - 1)
    return img[src(H, oh)][:, src(W, ow)]


_BICUBIC_TABLE = None


def _bicubic_table():
    """InitCoeffsTable of resize_bicubic_op.cc: 1 025 pairs, [2i] the kernel on |x| <= 1, [2i+1] on 1 <= |x| <= 2, A = -0.75;
    x is a float, the polynomial is evaluated in double and stored as float"""
    global _BICUBIC_TABLE
    if _BICUBIC_TABLE is None:
        A = -0.75
        t = np.zeros((1025, 2), np.float32)
        for i in range(1025):
            x = float(np.float32(i * 1.0 / 1024))
            t[i, 0] = ((A + 2) * x - (A + 3)) * x * x + 1
            x = float(np.float32(x + 1.0))
            t[i, 1] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
        _BICUBIC_TABLE = t
    return _BICUBIC_TABLE


def _bicubic_taps(n_in, n_out, align_corners=True):
    """GetWeightsAndIndices of resize_bicubic_op.cc for every output position: ([n_out,4] f32 weights, [n_out,4] indices)"""
    tab = _bicubic_table()
    scale = _resize_scale(n_in, n_out, align_corners)
    pos = (np.arange(n_out, dtype=np.float32) * scale).astype(np.float32)        # scale * out_loc, a float product
    loc = pos.astype(np.int64)                                                   # const int64 in_loc = scale * out_loc
    delta = (pos - loc.astype(np.float32)).astype(np.float32)
    off = np.rint((delta * np.float32(1024)).astype(np.float32)).astype(np.int64)   # lrintf: round half to even
    w = np.stack([tab[off, 1], tab[off, 0], tab[1024 - off, 0], tab[1024 - off, 1]], -1)
    idx = np.clip(np.stack([loc - 1, loc, loc + 1, loc + 2], -1), 0, n_in - 1)
    return w, idx


def resize_bicubic_align(img, oh, ow, align_corners=True):
    """TF 1.13 ResizeBicubic: per output row the four source rows are interpolated along x (v0 w0 + v1 w1 + v2 w2 + v3 w3 in
    float), then the four results along y with the same expression."""
    H, W, _ = img.shape
    wy, iy = _bicubic_taps(H, oh, align_corners)
    wx, ix = _bicubic_taps(W, ow, align_corners)
    wx_t = torch.from_numpy(wx).view(1, ow, 4, 1)
    rows = img[:, torch.from_numpy(ix)]                                   # [H, ow, 4, C]
    horiz = rows[:, :, 0] * wx_t[:, :, 0] + rows[:, :, 1] * wx_t[:, :, 1] + rows[:, :, 2] * wx_t[:, :, 2] + rows[:, :, 3] * wx_t[:, :, 3]
    wy_t = torch.from_numpy(wy).view(oh, 4, 1, 1)
    col = horiz[torch.from_numpy(iy)]                                      # [oh, 4, ow, C]
    return col[:, 0] * wy_t[:, 0] + col[:, 1] * wy_t[:, 1] + col[:, 2] * wy_t[:, 2] + col[:, 3] * wy_t[:, 3]


def resize_bilinear_legacy(img, oh, ow):
    """TF-1.x tf.image.resize default (align_corners=False, no half-pixel centres): src = dst * in/out."""
    H, W, _ = img.shape
    iy = np.arange(oh, dtype=np.float32) * np.float32(H / oh)
    ix = np.arange(ow, dtype=np.float32) * np.float32(W / ow)
    y0 = np.floor(iy).astype(np.int64); y1 = np.minimum(y0 + 1, H - 1)
    x0 = np.floor(ix).astype(np.int64); x1 = np.minimum(x0 + 1, W - 1)
    ly = torch.from_numpy(iy - y0.astype(np.float32)).view(oh, 1, 1)
    lx = torch.from_numpy(ix - x0.astype(np.float32)).view(1, ow, 1)
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * lx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * lx
    return top + (bot - top) * ly


def adjust_hue(img, delta):
    """TF AdjustHue: RGB -> (hue, min, max) -> hue + delta (mod 1) -> RGB; scale-free in the value range."""
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    vmax = torch.maximum(torch.maximum(r, g), b)
    vmin = torch.minimum(torch.minimum(r, g), b)
    rng = vmax - vmin
    norm = torch.where(rng > 0, 1. / (6. * rng), torch.zeros_like(rng))
    h = torch.where(r == vmax, norm * (g - b), torch.where(g == vmax, norm * (b - r) + 2. / 6., norm * (r - g) + 4. / 6.))
    h = torch.where(rng > 0, h, torch.zeros_like(h))
    h = torch.where(h < 0, h + 1., h)
    h = torch.remainder(h + delta, 1.0)
    dh = h * 6.
    i = torch.floor(dh).clamp(max=5)
    f = dh - i
    up, dn = vmin + rng * f, vmin + rng * (1. - f)
    i = i.long()
    rr = torch.stack([vmax, dn, vmin, vmin, up, vmax], -1).gather(-1, i.unsqueeze(-1)).squeeze(-1)
    gg = torch.stack([up, vmax, vmax, dn, vmin, vmin], -1).gather(-1, i.unsqueeze(-1)).squeeze(-1)
    bb = torch.stack([vmin, vmin, up, vmax, vmax, dn], -1).gather(-1, i.unsqueeze(-1)).squeeze(-1)
    return torch.stack([rr, gg, bb], -1)


def rotate_bilinear(img, ang):
    """tf.contrib.image.rotate(img, ang, 'BILINEAR'): out(x, y) = in(cos x - sin y + ox, sin x + cos y + oy), 0 outside."""
    H, W, _ = img.shape
    c, s = math.cos(ang), math.sin(ang)
    ox = ((W - 1) - (c * (W - 1) - s * (H - 1))) / 2.
    oy = ((H - 1) - (s * (W - 1) + c * (H - 1))) / 2.
    ys = torch.arange(H, dtype=torch.float32).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, W)
    sx = c * xs - s * ys + ox
    sy = s * xs + c * ys + oy
    x0, y0 = torch.floor(sx), torch.floor(sy)
    fx, fy = (sx - x0).unsqueeze(-1), (sy - y0).unsqueeze(-1)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        v = img[yy.clamp(0, H - 1).long(), xx.clamp(0, W - 1).long()]
        return v * ok.unsqueeze(-1).float()
    top = tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx
    bot = tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx
    return top * (1 - fy) + bot * fy


def augment_image(image, p, constant_values=0.):
    """the image chain of one image (HWC f32) under plan p."""
    img = image
    if p['resize']:
        img = {'NEAREST_NEIGHBOR': resize_nearest_align, 'BICUBIC': resize_bicubic_align}.get(p.get('method'), resize_bilinear_align)(
            img, p['resize_h'], p['resize_w'])
    canvas = torch.full((p['zoom_h'], p['zoom_w'], img.shape[2]), float(constant_values))
    canvas[:img.shape[0], :img.shape[1]] = img[:p['zoom_h'], :p['zoom_w']]
    img = canvas[p['crop_h']:p['crop_h'] + p['out_h'], p['crop_w']:p['crop_w'] + p['out_w']]
    if p['flip_td']:
        img = torch.flip(img, [0])
    if p['flip_lr']:
        img = torch.flip(img, [1])
    if p['brightness'] is not None:
        img = img + p['brightness']
    if p['contrast'] is not None:
        mean = img.mean(dim=(0, 1), keepdim=True)
        img = (img - mean) * p['contrast'] + mean
    if p['hue'] is not None:
        img = adjust_hue(img, p['hue'])
    if p['angle'] is not None:
        img = rotate_bilinear(img, p['angle'])
    return img.contiguous()


def augment_boxes(ground_truth, p, pad_truth_to):
    """ground_truth [G,5] = ymin, ymax, xmin, xmax, class -> ([pad_truth_to,5] = yc, xc, h, w, class; fallback flag)."""
    g = ground_truth.float()
    ymin, ymax, xmin, xmax, cls = (g[:, k] for k in range(5))
    copy = torch.stack([ymin / 2. + ymax / 2., xmin / 2. + xmax / 2., ymax - ymin, xmax - xmin, cls], -1)
    oh, ow = float(p['out_h']), float(p['out_w'])
    ymin, ymax = ymin * p['ratio_y'], ymax * p['ratio_y']
    xmin, xmax = xmin * p['ratio_x'], xmax * p['ratio_x']
    ymin, ymax = ymin - float(p['crop_h']), ymax - float(p['crop_h'])
    xmin, xmax = xmin - float(p['crop_w']), xmax - float(p['crop_w'])
    if p['flip_td']:
        ymax, ymin = oh - ymin - 1., oh - ymax - 1.
    if p['flip_lr']:
        xmax, xmin = ow - xmin - 1., ow - xmax - 1.
    if p['angle'] is not None:
        ang = -p['angle']
        c, s = math.cos(ang), math.sin(ang)
        cx, cy = (ow - 1.) / 2., (oh - 1.) / 2.
        offx = cx * (1 - c) + cy * s
        offy = cy * (1 - c) - cx * s
        xs = torch.stack([xmin * c - ymin * s, xmax * c - ymax * s, xmin * c - ymax * s, xmax * c - ymin * s], -1) + offx
        ys = torch.stack([xmin * s + ymin * c, xmax * s + ymax * c, xmin * s + ymax * c, xmax * s + ymin * c], -1) + offy
        xmin, xmax = xs.min(-1).values, xs.max(-1).values
        ymin, ymax = ys.min(-1).values, ys.max(-1).values
    ymin, ymax = ymin.clamp(0., oh - 1.), ymax.clamp(0., oh - 1.)
    xmin, xmax = xmin.clamp(0., ow - 1.), xmax.clamp(0., ow - 1.)
    yc, xc = (ymin + ymax) / 2., (xmin + xmax) / 2.
    keep = (yc > 0.) & (yc < oh - 1.) & (xc > 0.) & (xc < ow - 1.)
    out = torch.stack([yc, xc, ymax - ymin, xmax - xmin, cls], -1)[keep]
    fallback = out.shape[0] == 0
    if fallback:
        fact = torch.tensor([oh / p['in_h'], ow / p['in_w'], oh / p['in_h'], ow / p['in_w'], 1.])
        out = copy * fact
    pad = torch.full((pad_truth_to, 5), -1.0)
    pad[:out.shape[0]] = out[:pad_truth_to]
    return pad, fallback


def image_augmentor(image, input_shape, data_format, output_shape, zoom_size=None, crop_method=None, flip_prob=None,
                    fill_mode='BILINEAR', keep_aspect_ratios=False, constant_values=0., color_jitter_prob=None, rotate=None,
                    ground_truth=None, pad_truth_to=None, draws=(), image_quirk=False):
    """same signature as the reference's function (:7-9) plus the scripted draws."""
    check_args(data_format, fill_mode, zoom_size, output_shape, crop_method, keep_aspect_ratios, constant_values,
               color_jitter_prob, flip_prob, rotate, ground_truth)
    p = plan(input_shape, output_shape, zoom_size, crop_method, flip_prob, fill_mode, keep_aspect_ratios, color_jitter_prob,
             rotate, draws)
    hwc = image.permute(1, 2, 0) if data_format == 'channels_first' else image
    out = augment_image(hwc.float(), p, constant_values)
    if ground_truth is None:
        return out.permute(2, 0, 1).contiguous() if data_format == 'channels_first' else out
    gt, fallback = augment_boxes(ground_truth, p, pad_truth_to)
    if fallback:
        out = resize_bilinear_legacy(hwc.float(), p['out_h'], p['out_w'])
    if data_format == 'channels_first':
        out = out.permute(2, 0, 1).contiguous()
    return (image if image_quirk else out), gt
