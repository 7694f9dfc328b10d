"""A torch-CPU stand-in for the launches of odtk.ops (TEST INFRASTRUCTURE ONLY): every function has the signature and the memory
semantics (rows x pitch operands, zero pad columns, accumulate flags, in-place outputs) of the libodtk wrapper it replaces, computed with
plain torch ops.  With it the model classes' HOST LOGIC -- buffer planning, launch order, gradient bookkeeping, optimizer glue -- runs
and is checked against the oracles without a GPU (`-m "not gpu"`); the kernels themselves are checked on the GPU, one by one and
through the same classes.  Usage: `with mock_ops.installed(): model = odtk.YOLOv3(dict(config, device='cpu'), provider)`."""
import contextlib
import inspect

import torch
import torch.nn.functional as F

BN_EPS, BN_MOM = 1e-3, 0.99
_POOL_ARGMAX = {}            # index buffer -> torch's arg-max of the forward pooling (the recorded routing of the 2x2 pools)


def _storage(y):
    """the kernels get a raw pointer (here: into the middle of the [N][A][width] prediction tensor) and address past the view they were
    handed: (whole storage as a flat tensor, element offset of y in it)"""
    return torch.empty(0, dtype=y.dtype, device=y.device).set_(y.untyped_storage()), y.storage_offset()


def _nhwc(rows, d_or_shape, C):
    N, H, W = d_or_shape
    return rows[:, :C].float().reshape(N, H, W, C)


def _conv_weights(w_flat, d):
    return w_flat.float().reshape(d.K, d.R, d.S, d.C)


def _same_pads(in_size, out_size, k, stride, dil, pad_before):
    total = max((out_size - 1) * stride + (k - 1) * dil + 1 - in_size, 0)
    return pad_before, total - pad_before


def _conv(x_nhwc, w_krsc, d):
    pt, pb = _same_pads(d.H, d.Ho, d.R, d.stride, d.dil, d.pad_t)
    pl, pr = _same_pads(d.W, d.Wo, d.S, d.stride, d.dil, d.pad_l)
    xp = F.pad(x_nhwc.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, w_krsc.permute(0, 3, 1, 2), None, stride=d.stride, dilation=d.dil).permute(0, 2, 3, 1)


def conv2d_fwd(d, x, w, bias, y, relu):
    out = _conv(_nhwc(x, (d.N, d.H, d.W), d.C), _conv_weights(w, d), d)
    if bias is not None:
        out = out + bias.float()
    if relu:
        out = torch.relu(out)
    y[:, :d.K] = out.reshape(-1, d.K).to(y.dtype)


def conv2d_fwd_pool2x2(d, x, w, bias, y, relu, y_pool, idx):
    out = _conv(_nhwc(x, (d.N, d.H, d.W), d.C), _conv_weights(w, d), d)
    if bias is not None:
        out = out + bias.float()
    if relu:
        out = torch.relu(out)
    if y is not None:
        y[:, :d.K] = out.reshape(-1, d.K).to(y.dtype)
    Ho, Wo = (d.Ho + 1) // 2, (d.Wo + 1) // 2
    v = out.to(y_pool.dtype).float()                             # the pool sees the STORED (rounded) activations
    vp = F.pad(v.permute(0, 3, 1, 2), (0, 2 * Wo - d.Wo, 0, 2 * Ho - d.Ho), value=float('-inf'))
    pooled, arg = F.max_pool2d(vp, 2, 2, return_indices=True)
    if idx is not None:
        _POOL_ARGMAX[idx.data_ptr()] = (arg, vp.shape)
    y_pool[:, :d.K] = pooled.permute(0, 2, 3, 1).reshape(-1, d.K).to(y_pool.dtype)


def conv2d_fwd_pool2x2_fused(d):
    return True


def conv2d_relu_bits_supported(producer, consumer, consumer_lddy):
    return producer.K == 64 and consumer.C == 64 and consumer.K == 64 and consumer.R == 3 and producer.C <= 8


def _pack_bits(y):
    """[M][ld] activation -> [M][ld / 8] bytes, bit e = element 8 j + e > 0"""
    m = (y.float() > 0).reshape(y.shape[0], -1, 8).to(torch.int32)
    return (m * (2 ** torch.arange(8, dtype=torch.int32, device=y.device))).sum(-1).to(torch.uint8).reshape(-1)


def conv2d_fwd_bits(d, x, w, bias, y, relu, relu_bits):
    conv2d_fwd(d, x, w, bias, y, relu)
    relu_bits.copy_(_pack_bits(y))


def conv2d_dgrad_bits(d, dy, lddy, w_t, relu_bits, dx, accumulate):
    b = relu_bits.reshape(dx.shape[0], -1).to(torch.int32)
    src = ((b.unsqueeze(-1) >> torch.arange(8, dtype=torch.int32, device=b.device)) & 1).reshape(dx.shape[0], -1).to(dx.dtype)
    conv2d_dgrad(d, dy, lddy, w_t, src, dx, accumulate)


def conv2d_dgrad(d, dy, lddy, w_t, relu_src, dx, accumulate):
    Kp = lddy
    wt = w_t.float().reshape(d.C, d.R * d.S, Kp)
    w = wt.flip(1)[:, :, :d.K].permute(2, 1, 0).reshape(d.K, d.R, d.S, d.C)              # undo the flip / transpose of filter_prepare
    x = torch.zeros(d.N, d.H, d.W, d.C, requires_grad=True, device=dy.device)
    out = _conv(x, w, d)
    g, = torch.autograd.grad(out, x, dy[:, :d.K].float().reshape(out.shape))
    g = g.reshape(-1, d.C)
    if accumulate:
        g = g + dx[:, :d.C].float()
    if relu_src is not None:
        g = g * (relu_src[:, :d.C].float() > 0)
    dx[:, :d.C] = g.to(dx.dtype)


def conv2d_wgrad(d, x, dy, lddy, dw, dbias=None):
    w = torch.zeros(d.K, d.R, d.S, d.C, requires_grad=True, device=x.device)
    out = _conv(_nhwc(x, (d.N, d.H, d.W), d.C), w, d)
    g, = torch.autograd.grad(out, w, dy[:, :d.K].float().reshape(out.shape))
    dw.view(-1)[: g.numel()] += g.reshape(-1)
    if dbias is not None:
        dbias[: d.K] += dy[:, :d.K].float().sum(0)


# (ODTK_F32X3 descriptors -- f32 tensors, split bf16 arithmetic inside the library -- are restated as what they approximate: the plain f32 convolution)
def conv2d_x3_supported(d):
    return 7 if d.dtype == 2 and d.C * d.K * d.R * d.S >= 20000 else 0


def colsum(dy, M, C_, ld, out, accumulate, ws):
    v = dy[:M, :C_].float().sum(0)
    out[:C_] = out[:C_] + v if accumulate else v


class FilterPrepareBatch:
    def __init__(self, entries, dtype, device):
        self.entries = entries

    def run(self):
        for (w, wt, K, R, S, C_, Kp) in self.entries:
            full = torch.zeros(C_, R * S, Kp, device=w.device)
            full[:, :, :K] = w.float().reshape(K, R * S, C_).permute(2, 1, 0).flip(1)
            wt.copy_(full.reshape(-1).to(wt.dtype))


def preprocess(images, mean3, ldx, dtype, x):
    x.zero_()
    x[:, :3] = (images.float() - torch.tensor(mean3, dtype=torch.float32, device=images.device)).reshape(-1, 3).to(x.dtype)


def _act(v, relu):
    return torch.relu(v) if relu == 1 else (torch.where(v > 0, v, 0.1 * v) if relu == 2 else v)


def _strided_index(M, C_, ldy, rows_per_img, y_img_stride, base, device=None):
    m = torch.arange(M, device=device)
    return (base + (m // rows_per_img) * y_img_stride + (m % rows_per_img) * ldy).view(-1, 1) + torch.arange(C_, device=device).view(1, -1)


def _read_rows(y, M, C_, ldy, rows_per_img, y_img_stride):
    if y_img_stride == 0 and rows_per_img == M and y.dim() == 2:
        return y[:M, :C_].float()
    flat, base = _storage(y)
    return flat[_strided_index(M, C_, ldy, rows_per_img, y_img_stride, base, y.device).reshape(-1)].reshape(M, C_).float()


def _write_rows(y, vals, M, C_, ldy, rows_per_img, y_img_stride):
    if y_img_stride == 0 and rows_per_img == M and y.dim() == 2:
        y[:M, :C_] = vals.to(y.dtype)
        return
    flat, base = _storage(y)
    flat[_strided_index(M, C_, ldy, rows_per_img, y_img_stride, base, y.device).reshape(-1)] = vals.reshape(-1).to(flat.dtype)


def bn_fwd(z, M, C_, ldz, gamma, beta, mmean, mvar, save_mean, save_invstd, training, relu, y, ldy, rows_per_img, y_img_stride, ws):
    v = z[:M, :C_].float()
    if training:
        mean = v.mean(0)
        var = ((v - mean) ** 2).mean(0)
        save_mean.copy_(mean); save_invstd.copy_(torch.rsqrt(var + BN_EPS))
        mmean.mul_(BN_MOM).add_((1 - BN_MOM) * mean)
        mvar.mul_(BN_MOM).add_((1 - BN_MOM) * var * (M / max(M - 1, 1)))
    else:
        mean, var = mmean.float(), mvar.float()
    out = (v - mean) * (torch.rsqrt(var + BN_EPS) * gamma.float()) + beta.float()
    _write_rows(y, _act(out, relu), M, C_, ldy, rows_per_img, y_img_stride)


def bn_bwd(z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, dz, dgamma, dbeta, ws):
    d = _read_rows(dy, M, C_, ldy, rows_per_img, y_img_stride)
    if relu:
        yv = _read_rows(y, M, C_, ldy, rows_per_img, y_img_stride)
        d = d * (yv > 0) if relu == 1 else torch.where(yv > 0, d, 0.1 * d)
    xh = (z[:M, :C_].float() - save_mean) * save_invstd
    dbeta.copy_(d.sum(0)); dgamma.copy_((d * xh).sum(0))
    out = gamma.float() * save_invstd * (d - d.mean(0) - xh * (d * xh).mean(0))
    dz.zero_()
    dz[:M, :C_] = out.to(dz.dtype)


def add2d(a, lda, b, ldb, y, ldy, M, C_):
    v = a[:M, :C_].float() + (b[:M, :C_].float() if b is not None else 0.)
    y[:M, :C_] = v.to(y.dtype)


def upsample2x_fwd(x, ldx, y, ldy, N, H, W, C_):
    v = x[:, :C_].reshape(N, H, W, C_).repeat_interleave(2, 1).repeat_interleave(2, 2)
    y[:, :C_] = v.reshape(-1, C_)


def upsample2x_bwd(dy, lddy, dx, lddx, N, H, W, C_, accumulate=False):
    v = dy[:, :C_].float().reshape(N, H, 2, W, 2, C_).sum(dim=(2, 4)).reshape(-1, C_)
    dx[:, :C_] = (v + (dx[:, :C_].float() if accumulate else 0.)).to(dx.dtype)


def _bilinear(x_nhwc, Ho, Wo):
    n, h, w, c = x_nhwc.shape
    fy = torch.arange(Ho, dtype=torch.float32, device=x_nhwc.device) * (h / Ho)
    fx = torch.arange(Wo, dtype=torch.float32, device=x_nhwc.device) * (w / Wo)
    y0, x0 = torch.floor(fy).long(), torch.floor(fx).long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    ly, lx = (fy - y0.float()).view(1, Ho, 1, 1), (fx - x0.float()).view(1, 1, Wo, 1)
    top = x_nhwc[:, y0][:, :, x0] + (x_nhwc[:, y0][:, :, x1] - x_nhwc[:, y0][:, :, x0]) * lx
    bot = x_nhwc[:, y1][:, :, x0] + (x_nhwc[:, y1][:, :, x1] - x_nhwc[:, y1][:, :, x0]) * lx
    return top + (bot - top) * ly


def resize_bilinear_fwd(x, ldx, y, ldy, N, H, W, Ho, Wo, C_, accumulate=False):
    v = _bilinear(x[:, :C_].float().reshape(N, H, W, C_), Ho, Wo).reshape(-1, C_)
    y[:, :C_] = (v + (y[:, :C_].float() if accumulate else 0.)).to(y.dtype)


def resize_bilinear_bwd(dy, lddy, dx, lddx, N, H, W, Ho, Wo, C_, accumulate=False):
    x = torch.zeros(N, H, W, C_, requires_grad=True, device=dy.device)
    g, = torch.autograd.grad(_bilinear(x, Ho, Wo), x, dy[:, :C_].float().reshape(N, Ho, Wo, C_))
    dx[:, :C_] = (g.reshape(-1, C_) + (dx[:, :C_].float() if accumulate else 0.)).to(dx.dtype)


def _bilinear2(x_nhwc, Ho, Wo, align_corners):
    n, h, w, c = x_nhwc.shape
    sy = (h - 1) / (Ho - 1) if (align_corners and Ho > 1) else h / Ho
    sx = (w - 1) / (Wo - 1) if (align_corners and Wo > 1) else w / Wo
    fy = torch.arange(Ho, dtype=torch.float32, device=x_nhwc.device) * torch.tensor(sy, dtype=torch.float32)
    fx = torch.arange(Wo, dtype=torch.float32, device=x_nhwc.device) * torch.tensor(sx, dtype=torch.float32)
    y0, x0 = torch.floor(fy).long(), torch.floor(fx).long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    ly, lx = (fy - y0.float()).view(1, Ho, 1, 1), (fx - x0.float()).view(1, 1, Wo, 1)
    top = x_nhwc[:, y0][:, :, x0] + (x_nhwc[:, y0][:, :, x1] - x_nhwc[:, y0][:, :, x0]) * lx
    bot = x_nhwc[:, y1][:, :, x0] + (x_nhwc[:, y1][:, :, x1] - x_nhwc[:, y1][:, :, x0]) * lx
    return top + (bot - top) * ly


def resize_bilinear2_fwd(x, ldx, y, ldy, N, H, W, Ho, Wo, C_, align_corners, accumulate=False):
    v = _bilinear2(x[:, :C_].float().reshape(N, H, W, C_), Ho, Wo, align_corners).reshape(-1, C_)
    y[:, :C_] = (v + (y[:, :C_].float() if accumulate else 0.)).to(y.dtype)


def resize_bilinear2_bwd(dy, lddy, dx, lddx, N, H, W, Ho, Wo, C_, align_corners, accumulate=False, relu_src=None):
    x = torch.zeros(N, H, W, C_, requires_grad=True, device=dy.device)
    g, = torch.autograd.grad(_bilinear2(x, Ho, Wo, align_corners), x, dy[:, :C_].float().reshape(N, Ho, Wo, C_))
    g = g.reshape(-1, C_)
    if relu_src is not None:
        g = g * (relu_src[:, :C_].float() > 0)
    dx[:, :C_] = (g + (dx[:, :C_].float() if accumulate else 0.)).to(dx.dtype)


def copy_channels(src, lds, src_off, dst, ldd, dst_off, M, C_, accumulate=False, relu_src=None):
    v = src[:, src_off: src_off + C_].float()
    if relu_src is not None:
        v = v * (relu_src[:, dst_off: dst_off + C_].float() > 0)
    dst[:, dst_off: dst_off + C_] = (v + (dst[:, dst_off: dst_off + C_].float() if accumulate else 0.)).to(dst.dtype)


def _pool(x_nhwc, k, stride, pt, pl, Ho, Wo):
    n, h, w, c = x_nhwc.shape
    pb = max((Ho - 1) * stride + k - h - pt, 0)
    pr = max((Wo - 1) * stride + k - w - pl, 0)
    xp = F.pad(x_nhwc.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf'))
    return F.max_pool2d(xp, k, stride).permute(0, 2, 3, 1)


def maxpool_fwd(x, y, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    y[:, :C_] = _pool(x[:, :C_].float().reshape(N, H, W, C_), k, stride, pad_t, pad_l, Ho, Wo).reshape(-1, C_).to(y.dtype)


def maxpool_bwd(x, y, dy, dx, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    xr = x[:, :C_].float().reshape(N, H, W, C_).clone().requires_grad_(True)
    g, = torch.autograd.grad(_pool(xr, k, stride, pad_t, pad_l, Ho, Wo), xr, dy[:, :C_].float().reshape(N, Ho, Wo, C_))
    dx[:, :C_] = g.reshape(-1, C_).to(dx.dtype)


def rows_to_f32(x, ldx, y, ldy, rows_per_img, y_img_stride, M, C_):
    flat, base = _storage(y)
    for n in range(M // rows_per_img):
        blk = x[n * rows_per_img:(n + 1) * rows_per_img, :C_].float()
        idx = base + n * y_img_stride + torch.arange(rows_per_img, device=y.device).view(-1, 1) * ldy + torch.arange(C_, device=y.device).view(1, -1)
        flat[idx.reshape(-1)] = blk.reshape(-1)


def rows_from_f32(y, ldy, rows_per_img, y_img_stride, x, ldx, M, C_):
    flat, base = _storage(y)
    x.zero_()
    for n in range(M // rows_per_img):
        idx = base + n * y_img_stride + torch.arange(rows_per_img, device=y.device).view(-1, 1) * ldy + torch.arange(C_, device=y.device).view(1, -1)
        x[n * rows_per_img:(n + 1) * rows_per_img, :C_] = flat[idx.reshape(-1)].reshape(rows_per_img, C_).to(x.dtype)


def gn_workspace(N, C_, device):
    return torch.zeros(4, dtype=torch.uint8)


def _gn_xhat(x, N, HW, C_, groups, mean, rstd):
    v = x[:, :C_].float().reshape(N, HW, groups, C_ // groups)
    return (v - mean.view(N, 1, groups, 1)) * rstd.view(N, 1, groups, 1)


def gn_fwd(x, ldx, y, ldy, N, HW, C_, groups, gamma, beta, relu, save):
    v = x[:, :C_].float().reshape(N, HW, groups, C_ // groups)
    mean = v.mean(dim=(1, 3))
    var = ((v - mean.view(N, 1, groups, 1)) ** 2).mean(dim=(1, 3))
    rstd = torch.rsqrt(var + 1e-6)
    if save is not None:
        save[:, :, 0], save[:, :, 1] = mean, rstd
    out = _gn_xhat(x, N, HW, C_, groups, mean, rstd).reshape(N * HW, C_) * gamma.float() + beta.float()
    y[:, :C_] = _act(out, relu).to(y.dtype)


def gn_bwd(x, ldx, y, dy, ldy, dx, lddx, N, HW, C_, groups, gamma, save, relu, accumulate, dgamma, dbeta, ws):
    d = dy[:, :C_].float()
    if relu:
        d = d * (y[:, :C_].float() > 0)
    xh = _gn_xhat(x, N, HW, C_, groups, save[:, :, 0], save[:, :, 1])
    db_, dg_ = d.sum(0), (d * xh.reshape(N * HW, C_)).sum(0)
    if int(accumulate) & 2:
        db_, dg_ = db_ + dbeta, dg_ + dgamma
    dbeta.copy_(db_); dgamma.copy_(dg_)
    dg = (d * gamma.float()).reshape(N, HW, groups, C_ // groups)
    m1 = dg.mean(dim=(1, 3), keepdim=True)
    m2 = (dg * xh).mean(dim=(1, 3), keepdim=True)
    out = (save[:, :, 1].view(N, 1, groups, 1) * (dg - m1 - xh * m2)).reshape(N * HW, C_)
    if int(accumulate) & 1:
        out = out + dx[:, :C_].float()
    dx[:, :C_] = out.to(dx.dtype)


def exp_rows_to_f32(x, ldx, y, M, C_):
    y.reshape(M, C_).copy_(torch.exp(x[:M, :C_].float()))


def exp_rows_bwd(dy, y, dx, lddx, M, C_):
    dx.zero_()
    dx[:M, :C_] = (dy.reshape(M, C_) * y.reshape(M, C_)).to(dx.dtype)


def fcos_workspace(conf, N, device):
    return torch.zeros(4, dtype=torch.uint8)


def fcos_loss(conf, reg, center, gt, grad_scale, loss, d_conf, d_reg, d_center, ws):
    from oracle import fcos_ref as FR
    N = conf[0].shape[0]
    cs = [t.detach().clone().requires_grad_(True) for t in conf]
    rs = [t.detach().clone().requires_grad_(True) for t in reg]
    zs = [t.detach().clone().requires_grad_(True) for t in center]
    tot = 0.
    for i in range(N):
        li = FR.one_image_loss([c[i] for c in cs], [r[i] for r in rs], [z[i] for z in zs], gt[i])
        loss[i] = li.detach()
        tot = tot + li
    grads = torch.autograd.grad(tot * grad_scale, cs + rs + zs, allow_unused=True)
    for dst, g in zip(list(d_conf) + list(d_reg) + list(d_center), grads):
        dst.copy_(g if g is not None else torch.zeros_like(dst))


# ---- inference tails: decode kernels and the batched per-class NMS
def nms_batched(boxes, box_stride, scores, score_bstride, score_estride, valid, valid_bstride, valid_estride, valid_value, n, B,
                max_out_dev, max_out_stride, max_out_const, iou_thr, out_idx, cap, out_cnt):
    """tf.image.non_max_suppression for B problems with the strided operand addressing of odtk_nms_batched (oracle NMS inside);
    called with no-op intent by the mocked SSD loss path (max_out_dev given): then nothing is done"""
    if max_out_dev is not None:
        return
    from oracle import ssd300_ref as R
    bx, sc = boxes.reshape(-1), scores.reshape(-1)
    vd = valid.reshape(-1) if valid is not None else None
    idx = torch.arange(n)
    for b in range(B):
        bb = bx[b * box_stride: b * box_stride + 4 * n].reshape(n, 4)
        ss = sc[b * score_bstride + idx * score_estride]
        ok = torch.ones(n, dtype=torch.bool) if vd is None else (vd[b * valid_bstride + idx * valid_estride] == valid_value)
        rows = torch.nonzero(ok).flatten()
        sel = torch.from_numpy(R.nms(bb[rows].numpy(), ss[rows].numpy(), int(max_out_const), float(iou_thr)).astype('int64'))
        k = min(sel.numel(), cap)
        out_idx[b, :k] = rows[sel[:k]].to(out_idx.dtype)
        out_cnt[b] = k


def yolov3_decode_candidates(preds, priors_flat, decode_scale):
    from oracle import yolov3_ref as YR
    assert [float(v) for v in decode_scale] == [32., 32., 16.]
    pri = [[[priors_flat[(l * 3 + a) * 2] * YR.STRIDE[l], priors_flat[(l * 3 + a) * 2 + 1] * YR.STRIDE[l]] for a in range(3)] for l in range(3)]
    return YR.decode_candidates([p.float() for p in preds], num_classes=preds[0].shape[-1] - 5, priors_px=pri)


def fcos_decode_candidates(conf, reg, center):
    from oracle import fcos_ref as FR
    return FR.decode_candidates([c.float() for c in conf], [r.float() for r in reg], [z.float() for z in center])


def retina_decode(pconf, pbox, yx, hw, thr):
    from oracle import retinanet_ref as RR
    conf, boxes, keep, cand = RR.decode_candidates(pbox[:, :2], pbox[:, 2:], pconf, (None, None, yx, hw), thr, num_classes=pconf.shape[1])
    return conf.contiguous(), boxes.contiguous(), keep.to(torch.uint8), cand.to(torch.uint8)


# ---- SSD300-specific launches


def maxpool_fwd_argmax(x, y, arg, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    xr = x[:, :C_].float().reshape(N, H, W, C_)
    y[:, :C_] = _pool(xr, k, stride, pad_t, pad_l, Ho, Wo).reshape(-1, C_).to(y.dtype)
    _POOL_ARGMAX[arg.data_ptr()] = xr                         # the routing is re-derived from the stored input by autograd (first maximum, as TF)


def maxpool_bwd_argmax(arg, dy, dx, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    xr = _POOL_ARGMAX[arg.data_ptr()].clone().requires_grad_(True)
    g, = torch.autograd.grad(_pool(xr, k, stride, pad_t, pad_l, Ho, Wo), xr, dy[:, :C_].float().reshape(N, Ho, Wo, C_))
    dx[:, :C_] = g.reshape(-1, C_).to(dx.dtype)


def maxpool2x2_fwd_idx(x, y, idx, N, H, W, C_, ld, Ho, Wo):
    v = x[:, :C_].float().reshape(N, H, W, C_)
    vp = F.pad(v.permute(0, 3, 1, 2), (0, 2 * Wo - W, 0, 2 * Ho - H), value=float('-inf'))
    out, arg = F.max_pool2d(vp, 2, 2, return_indices=True)
    _POOL_ARGMAX[idx.data_ptr()] = (arg, vp.shape)
    y[:, :C_] = out.permute(0, 2, 3, 1).reshape(-1, C_).to(y.dtype)


def maxpool2x2_bwd_idx(idx, dy, dx, N, H, W, C_, ld, Ho, Wo):
    arg, shp = _POOL_ARGMAX[idx.data_ptr()]
    g = F.max_unpool2d(dy[:, :C_].float().reshape(N, Ho, Wo, C_).permute(0, 3, 1, 2).contiguous(), arg, 2, 2, output_size=shp[2:])
    dx[:, :C_] = g[:, :, :H, :W].permute(0, 2, 3, 1).reshape(-1, C_).to(dx.dtype)


def l2norm_fwd(x, y, M, C_, ld, gamma):
    v = x[:M, :C_].float()
    y[:M, :C_] = (v / torch.sqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12)) * gamma.float()).to(y.dtype)


def l2norm_bwd(x, dy, dx, M, C_, ld, gamma, dgamma, accumulate, relu_src):
    v = x[:M, :C_].float().clone().requires_grad_(True)
    gm = gamma.float().clone().requires_grad_(True)
    out = v / torch.sqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12)) * gm
    gv, gg = torch.autograd.grad(out, [v, gm], dy[:M, :C_].float())
    dgamma += gg
    if accumulate:
        gv = gv + dx[:M, :C_].float()
    if relu_src is not None:
        gv = gv * (relu_src[:M, :C_].float() > 0)
    dx[:M, :C_] = gv.to(dx.dtype)


def ssd_priors(input_size, fsizes, nas, prior_hw_flat, device):
    """the prior kernel's arithmetic in numpy float32, in the op order of the reference's _get_abbox (SSD300.py:323-343)"""
    import numpy as np
    f32 = np.float32
    outs, k = [[], [], [], []], 0
    for f, na in zip(fsizes, nas):
        pr = np.asarray(prior_hw_flat[2 * k: 2 * (k + na)], dtype=np.float64).astype(f32).reshape(1, 1, na, 2)
        k += na
        ty = (np.arange(0., f, dtype=f32).reshape(-1, 1, 1, 1) + f32(0.5))
        tx = (np.arange(0., f, dtype=f32).reshape(1, -1, 1, 1) + f32(0.5))
        ty = np.tile(ty, [1, f, 1, 1]) * f32(input_size) / f32(f)
        tx = np.tile(tx, [f, 1, 1, 1]) * f32(input_size) / f32(f)
        tyx = np.tile(np.concatenate([ty, tx], -1), [1, 1, na, 1])
        y1x1 = (tyx - pr / f32(2.)).reshape(-1, 2)
        y2x2 = (tyx + pr / f32(2.)).reshape(-1, 2)
        for o, v in zip(outs, (y1x1, y2x2, y1x1 / f32(2.) + y2x2 / f32(2.), y2x2 - y1x1)):
            o.append(v.astype(f32))
    y1x1, y2x2, yx, hw = (torch.from_numpy(np.concatenate(o, 0)) for o in outs)
    return y1x1, y2x2, yx, hw, torch.cat([yx - hw / 2., yx + hw / 2.], 1)


def ssd_match(*a):
    pass                                                       # the mocked ssd_loss below matches / mines internally (oracle)


def softmax_ce_const(*a):
    pass


def ssd_loss(pred, Cn, yx, hw, gt, ngt, best, status, rgindex, counts, negloss, sel_idx, sel_cnt, grad_scale, loss_parts, dpred):
    from oracle import ssd300_ref as R
    anchors = R.priors()
    pr = pred.detach().clone().requires_grad_(True)
    tot = 0.
    for i in range(pred.shape[0]):
        li = R.one_image_loss(pr[i, :, Cn:Cn + 2], pr[i, :, Cn + 2:Cn + 4], pr[i, :, :Cn], anchors, gt[i])
        loss_parts[i, 3] = li.detach()
        tot = tot + li
    g, = torch.autograd.grad(tot * grad_scale, pr)
    dpred.copy_(g)


def sgd_momentum(p, m, g, lr, momentum, wd, grad_scale, l2_partial, p_cast):
    l2_partial.zero_()
    l2_partial[0] = (p * p).sum() / 2
    m.mul_(momentum).add_(g * grad_scale + wd * p)
    p.sub_(lr * m)
    if p_cast is not None and p_cast is not p:
        p_cast.copy_(p.to(p_cast.dtype))


def sum_f32(x, out):
    out[0] = x.sum()


def zero(x):
    x.zero_()


def loss_total(a, na, stride_a, b, scale_a, scale_b, sum_a, sum_b, total):
    sa = a.reshape(-1)[:na].sum()                 # the callers hand over the strided VIEW (its data pointer + stride_a is what the library sees)
    sb = b.sum()
    if sum_a is not None:
        sum_a[0] = sa
    if sum_b is not None:
        sum_b[0] = sb
    total[0] = scale_a * sa + scale_b * sb


def cast_from_f32(x, out):
    out.copy_(x.to(out.dtype))


# ---- box side through the oracles (autograd supplies the gradients the kernels return)
def cast_to_f32(x, out):
    out.copy_(x.float())


def yolov3_workspace(preds, N, device):
    return torch.zeros(4, dtype=torch.uint8)


def yolov3_loss(preds, priors_flat, head_stride, gt, scales, grad_scale, loss_parts, d_preds, ws):
    from oracle import yolov3_ref as YR
    N = preds[0].shape[0]
    ps = [p.detach().clone().requires_grad_(True) for p in preds]
    tot = 0.
    for i in range(N):
        d = YR.one_image_loss([p[i] for p in ps], gt[i], num_classes=preds[0].shape[-1] - 5, coord_scale=scales[0], noobj_scale=scales[1],
                              obj_scale=scales[2], class_scale=scales[3], detail=True)
        loss_parts[i, 4] = d['total'].detach()
        tot = tot + d['total']
    grads = torch.autograd.grad(tot * grad_scale, ps)
    for dp, g in zip(d_preds, grads):
        dp.copy_(g)


def retina_anchors(input_dim, shapes, nas, prior_hw_flat, device):
    from oracle import retinanet_ref as RR
    return RR.anchors([0, input_dim, 3], shapes)


def retina_match_workspace(A, N, P, device):
    return torch.zeros(4, dtype=torch.uint8)


def retina_match(*a):
    pass                                                       # the mocked loss below matches internally


def retina_loss(pconf, pbox, yx, hw, gt, ngt, best, status, rgindex, counts, alpha, gamma, grad_scale, loss_parts, dconf, dbox):
    from oracle import retinanet_ref as RR
    anc = retina_loss.anchors
    pc, pb = pconf.detach().clone().requires_grad_(True), pbox.detach().clone().requires_grad_(True)
    tot = 0.
    for i in range(pconf.shape[0]):
        d = RR.one_image_loss(pb[i, :, :2], pb[i, :, 2:], pc[i], anc, gt[i], alpha, gamma, detail=True)
        loss_parts[i, 0], loss_parts[i, 1] = d['conf_loss'].detach(), (d['total'] - d['conf_loss']).detach()
        tot = tot + d['total']
    gc, gb = torch.autograd.grad(tot * grad_scale, [pc, pb])
    dconf.copy_(gc); dbox.copy_(gb)


def scratch_slot(slot):
    pass


# ---- CenterNet: input transform, 2x2 average pooling, Adam, loss / decode
def preprocess_norm(images, div, mean3, std3, ldx, dtype, x):
    x.zero_()
    v = (images.float() / div - torch.tensor(mean3, dtype=torch.float32, device=images.device)) / torch.tensor(std3, dtype=torch.float32, device=images.device)
    x[:, :3] = v.reshape(-1, 3).to(x.dtype)


def add_relu_fwd(a, lda, b, ldb, y, ldy, M, C_):
    y[:M, :C_] = torch.relu(a[:M, :C_].float() + b[:M, :C_].float()).to(y.dtype)


def relu_bwd(y, dy, ldy, dx, lddx, M, C_, accumulate=False):
    g = dy[:M, :C_].float() * (y[:M, :C_].float() > 0)
    dx[:M, :C_] = (g + (dx[:M, :C_].float() if accumulate else 0.)).to(dx.dtype)


def avgpool2x2_fwd(x, y, N, H, W, ld):
    v = x.float().reshape(N, H // 2, 2, W // 2, 2, ld)
    y.copy_((((v[:, :, 0, :, 0] + v[:, :, 0, :, 1]) + v[:, :, 1, :, 0]) + v[:, :, 1, :, 1]).div(4.).reshape(-1, ld).to(y.dtype))


def avgpool2x2_bwd(dy, dx, N, H, W, ld):
    v = (dy.float() / 4.).to(dx.dtype).reshape(N, H // 2, 1, W // 2, 1, ld).expand(N, H // 2, 2, W // 2, 2, ld)
    dx.copy_(v.reshape(-1, ld))


def adam(p, m, v, g, lr_t, beta1, beta2, eps, wd, grad_scale, l2_partial, p_cast):
    if l2_partial is not None:
        l2_partial.zero_()
        l2_partial[0] = 0.5 * (p.double() ** 2).sum().float()
    gg = g * grad_scale + wd * p
    m.add_((gg - m) * (1 - beta1))
    v.add_((gg * gg - v) * (1 - beta2))
    p.sub_((m * lr_t) / (torch.sqrt(v) + eps))
    if p_cast is not None:
        p_cast.copy_(p.to(p_cast.dtype))


def refinedet_loss(arm_loc, arm_conf, odm_loc, odm_conf, yx, hw, gt, ngt, best, status, rgindex, counts, negloss, sel_idx, sel_cnt, grad_scale,
                   loss_parts, d_arm_loc, d_arm_conf, d_odm_loc, d_odm_conf):
    """matching, mining and the two-stage loss by the oracle (the mocked retina_match / softmax_ce_const / nms_batched in front of it do nothing)"""
    from oracle import refinedet_ref as FR
    anc = (yx - hw / 2., yx + hw / 2., yx, hw)
    leaves = [t.detach().clone().requires_grad_(True) for t in (arm_loc, arm_conf, odm_loc, odm_conf)]
    tot = 0.
    for i in range(arm_loc.shape[0]):
        li = FR.one_image_loss(leaves[0][i, :, :2], leaves[0][i, :, 2:], leaves[1][i], leaves[2][i, :, :2], leaves[2][i, :, 2:], leaves[3][i], anc, gt[i],
                               odm_conf.shape[-1])
        loss_parts[i, 6] = li.detach()
        tot = tot + li
    for dst, g in zip((d_arm_loc, d_arm_conf, d_odm_loc, d_odm_conf), torch.autograd.grad(tot * grad_scale, leaves)):
        dst.copy_(g)


def refinedet_decode(arm_loc, arm_conf, odm_loc, odm_conf, yx, hw, thr):
    from oracle import refinedet_ref as FR
    conf, boxes, keep = FR.decode(arm_loc, arm_conf, odm_loc, odm_conf, (None, None, yx, hw), odm_conf.shape[-1])
    return conf.contiguous(), boxes.contiguous(), keep.to(torch.uint8), ((conf >= thr) & keep[:, None]).to(torch.uint8)


def centernet_workspace(N, H, W, Cn, device):
    return torch.zeros(4, dtype=torch.uint8)


def centernet_loss(keypoints, offset, size, gt, stride, grad_scale, loss_parts, d_keypoints, d_offset, d_size, ws):
    from oracle import centernet_ref as CR
    k, o, z = (t.detach().clone().requires_grad_(True) for t in (keypoints, offset, size))
    tot = 0.
    for i in range(keypoints.shape[0]):
        li = CR.one_image_loss(k[i], o[i], z[i], gt[i], stride)
        loss_parts[i, 3] = li.detach()
        tot = tot + li
    grads = torch.autograd.grad(tot * grad_scale, [k, o, z])
    for dst, g in zip((d_keypoints, d_offset, d_size), grads):
        dst.copy_(g)


def centernet_decode(keypoints, offset, size, stride, score_threshold, top_k, ws):
    from oracle import centernet_ref as CR
    return CR.decode(keypoints, offset, size, score_threshold, top_k, stride)


@contextlib.contextmanager
def installed():
    """swap the launching functions of odtk.ops for the ones above (and back)"""
    import odtk  # noqa: F401
    from odtk import ops
    names = [n for n, v in globals().items() if callable(v) and not n.startswith('_') and n not in ('installed', 'contextlib') and hasattr(ops, n)]
    old = {n: getattr(ops, n) for n in names}
    from odtk import _lib

    def recorded(fn):
        # what _lib.call does for the real library: while a launch list is being recorded (SSD300 use_graph='list') the call is kept for replay
        def w(*a, **k):
            rec = _lib.recording()
            if rec is not None:
                def replay():
                    fn(*a, **k)
                rec.append((replay, ()))
            return fn(*a, **k)
        w.__name__ = fn.__name__
        return w
    try:
        for n in names:
            f = globals()[n]
            setattr(ops, n, recorded(f) if inspect.isfunction(f) else f)
        yield
    finally:
        for n, v in old.items():
            setattr(ops, n, v)


def yolov2_loss(pred, priors_flat, stride, gt, scales, grad_scale, loss_parts, d_pred):
    from oracle import yolov2_ref as YR
    N, H, W, P, E = pred.shape
    priors = [[priors_flat[2 * k], priors_flat[2 * k + 1]] for k in range(P)]
    x = pred.detach().clone().requires_grad_(True)
    per = [YR.image_loss(x[i], gt[i], priors, scales, E - 5) for i in range(N)]
    torch.stack(per).sum().backward()
    d_pred.copy_((x.grad * grad_scale).view(d_pred.shape))
    loss_parts.zero_()
    loss_parts[:, 4] = torch.stack(per).detach()


def yolov2_decode_candidates(pred0, priors_flat, stride):
    from oracle import yolov2_ref as YR
    P = pred0.shape[2]
    return YR.decode(pred0, [[priors_flat[2 * k], priors_flat[2 * k + 1]] for k in range(P)])


# ---- Light-Head R-CNN (csrc/lhrcnn.hip): pixel kernels in plain torch, the RPN loss by the oracle (the mocked match / NMS in front of it do nothing)
def depthwise_conv(x, ldx, filt, y, ldy, N, H, W, C_, kh, kw, flip=False, accumulate=False):
    w = filt.float().reshape(kh, kw, C_)
    if flip:
        w = w.flip(0, 1)
    xin = x[:, :C_].float().reshape(N, H, W, C_).permute(0, 3, 1, 2)
    out = F.conv2d(F.pad(xin, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2)), w.permute(2, 0, 1).unsqueeze(1), None, groups=C_)
    out = out.permute(0, 2, 3, 1).reshape(-1, C_)
    y[:, :C_] = ((y[:, :C_].float() + out) if accumulate else out).to(y.dtype)


def depthwise_wgrad(x, ldx, dy, lddy, dfilt, N, H, W, C_, kh, kw):
    w = torch.zeros(C_, 1, kh, kw, requires_grad=True, device=x.device)
    xin = x[:, :C_].float().reshape(N, H, W, C_).permute(0, 3, 1, 2)
    out = F.conv2d(F.pad(xin, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2)), w, None, groups=C_)
    g, = torch.autograd.grad(out, w, dy[:, :C_].float().reshape(N, H, W, C_).permute(0, 3, 1, 2))
    dfilt.view(-1)[: kh * kw * C_] += g.squeeze(1).permute(1, 2, 0).reshape(-1)


def lhrcnn_match(anc, conf, gt, ws):
    return None


def lhrcnn_rpn_loss(anc, conf, bbox, gt, ws, num_classes, grad_scale, img_h, img_w, d_conf, d_bbox):
    """matching, both NMS, the loss, its gradients and the R-CNN slots by the oracle's rpn_one_image"""
    from oracle import lhrcnn_ref as LR
    N = gt.shape[0]
    row = anc['row'].long()
    a = dict(y1x1=anc['y1x1'], y2x2=anc['y2x2'], yx=anc['yx'], hw=anc['hw'])
    cf = conf.detach().clone().requires_grad_(True)
    bb = bbox.detach().clone().requires_grad_(True)
    tot = 0.
    lim = torch.tensor([img_h - 1., img_w - 1., img_h - 1., img_w - 1.])
    for k in ('roi_box', 'roi_prop', 'roi_truth'):
        ws[k].zero_()
    ws['roi_img'].fill_(-1); ws['roi_label'].fill_(-1); ws['roi_kind'].zero_()
    for i in range(N):
        loss, pos_prop, pos_lab, truth, neg_prop = LR.rpn_one_image(bb[i, row, :2], bb[i, row, 2:], cf[i, row], a, gt[i])
        tot = tot + loss
        ws['rpn_parts'][i, 3] = loss.detach()
        kp, kn = pos_prop.shape[0], neg_prop.shape[0]
        s = i * 256
        prop = torch.cat([pos_prop, neg_prop]).detach()
        prop = torch.minimum(torch.clamp(prop, min=0.), lim)
        ws['roi_prop'][s: s + kp + kn] = prop
        ws['roi_box'][s: s + kp + kn] = prop / lim
        ws['roi_truth'][s: s + kp] = truth.detach()
        ws['roi_img'][s: s + kp + kn] = i
        ws['roi_label'][s: s + kp] = pos_lab.to(torch.int32)
        ws['roi_label'][s + kp: s + kp + kn] = num_classes - 1
        ws['roi_kind'][s: s + kp] = 1
        ws['roi_kind'][s + kp: s + kp + kn] = 2
        ws['roi_counts'][i, 0], ws['roi_counts'][i, 1] = kp, kn
    g1, g2 = torch.autograd.grad(tot * grad_scale, [cf, bb])
    d_conf.copy_(g1); d_bbox.copy_(g2)


def crop_and_resize_fwd(feat, ldf, N, H, W, C_, boxes, box_img, crop, out, ldo):
    from oracle import lhrcnn_ref as LR
    f = feat[:, :C_].float().reshape(N, H, W, C_)
    live = box_img >= 0
    v = LR.crop_and_resize(f, boxes, box_img.clamp(min=0), crop).reshape(boxes.shape[0], -1)
    out[:, : crop * crop * C_] = (v * live.view(-1, 1)).to(out.dtype)


def crop_and_resize_bwd(d_out, ldo, N, H, W, C_, boxes, box_img, crop, d_feat, ldf):
    from oracle import lhrcnn_ref as LR
    f = torch.zeros(N, H, W, C_, requires_grad=True, device=d_out.device)
    live = box_img >= 0
    v = LR.crop_and_resize(f, boxes, box_img.clamp(min=0), crop).reshape(boxes.shape[0], -1)
    g, = torch.autograd.grad(v, f, d_out[:, : crop * crop * C_].float() * live.view(-1, 1))
    d_feat.zero_()
    d_feat[:, :C_] = g.reshape(-1, C_)


def lhrcnn_rcnn_loss(logits, ldl, pbbox, ldb, N, Cn, ws, grad_scale, d_logits, d_pbbox):
    kind, label = ws['roi_kind'], ws['roi_label'].long()
    z = logits[:, :Cn].detach().float().clone().requires_grad_(True)
    b = pbbox[:, :4].detach().float().clone().requires_grad_(True)
    live, pos = kind != 0, kind == 1
    lse = torch.logsumexp(z, dim=1)
    ce = (lse - z.gather(1, label.clamp(min=0).view(-1, 1)).squeeze(1)) * live
    d = b - ws['roi_truth']
    sl1 = torch.where(d.abs() < 1., 0.5 * d * d, d.abs() - 0.5).sum(-1) * pos
    rows, npos = int(live.sum()), int(pos.sum())
    per_ce, per_box = ce.view(N, 256).sum(1) / rows, sl1.view(N, 256).sum(1) / npos
    ws['rcnn_parts'][:, 0], ws['rcnn_parts'][:, 1] = per_ce.detach(), per_box.detach()
    g1, g2 = torch.autograd.grad((per_ce.sum() + per_box.sum()) * grad_scale, [z, b])
    d_logits.zero_(); d_pbbox.zero_()
    d_logits[:, :Cn] = g1.to(d_logits.dtype); d_pbbox[:, :4] = g2.to(d_pbbox.dtype)


def lhrcnn_rpn_decode(anc, conf0, bbox0, img_h, img_w, prop, score):
    row = anc['row'].long()
    p = bbox0[row]
    yx = p[:, :2] * anc['hw'] + anc['yx']
    hw = torch.exp(p[:, 2:]) * anc['hw']
    lim = torch.tensor([img_h - 1., img_w - 1., img_h - 1., img_w - 1.])
    prop.copy_(torch.minimum(torch.clamp(torch.cat([yx - hw / 2., yx + hw / 2.], -1), min=0.), lim))
    score.copy_(torch.softmax(conf0[row], -1)[:, 0])


def lhrcnn_gather_rois(prop, sel, cnt, img_h, img_w, roi_box, roi_prop, roi_img):
    k = min(int(cnt[0]), roi_img.shape[0])
    lim = torch.tensor([img_h - 1., img_w - 1., img_h - 1., img_w - 1.])
    roi_prop.zero_(); roi_box.zero_(); roi_img.fill_(-1)
    roi_prop[:k] = prop[sel.view(-1)[:k].long()]
    roi_box[:k] = roi_prop[:k] / lim
    roi_img[:k] = 0


def lhrcnn_rcnn_decode(logits, ldl, pbbox, ldb, roi_prop, roi_img, Cn, thr, conf, boxes, cand):
    live = roi_img >= 0
    cf = torch.softmax(logits[:, :Cn].float(), -1)
    fg = (cf.argmax(-1) < Cn - 1) & live
    conf.copy_(cf[:, : Cn - 1] * live.view(-1, 1))
    cand.copy_(((cf[:, : Cn - 1] >= thr) & fg.view(-1, 1)).to(cand.dtype))
    p_yx, p_hw = roi_prop[:, 0:2] / 2. + roi_prop[:, 2:4] / 2., roi_prop[:, 2:4] - roi_prop[:, 0:2]
    t = pbbox[:, :4].float()
    yx, hw = t[:, :2] * p_hw + p_yx, p_hw * torch.exp(t[:, 2:])
    boxes.copy_(torch.cat([yx - hw / 2., yx + hw / 2.], -1) * live.view(-1, 1))
