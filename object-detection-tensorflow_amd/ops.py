"""Torch-tensor front end of the libodtk C-ABI.

PyTorch is plumbing here: it owns device memory and streams.  Every function takes
contiguous CUDA(=HIP) tensors, passes raw pointers + the current HIP stream through
ctypes, and fails loudly (OdtkError) if the HIP library is absent -- no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from ._lib import BF16, F32, F32X3, ConvDesc, call

_TORCH_DT = {BF16: torch.bfloat16, F32: torch.float32}


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(dt: int):
    return _TORCH_DT[dt]


def chunk(dt: int) -> int:
    """channels per 16-byte chunk"""
    return 8 if dt == BF16 else 4


def pad_to(c: int, m: int) -> int:
    return (c + m - 1) // m * m


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "libodtk takes device pointers"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def same_pad(in_size: int, k: int, stride: int, dil: int = 1):
    """TF SAME arithmetic: (out, pad_before, pad_after)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - in_size, 0)
    return out, total // 2, total - total // 2


def conv_desc(N, H, W, C_, ldx, K, ldy, k, stride=1, dil=1, dtype=BF16, out_dtype=None) -> ConvDesc:
    Ho, pt, _ = same_pad(H, k, stride, dil)
    Wo, pl, _ = same_pad(W, k, stride, dil)
    return ConvDesc(N=N, H=H, W=W, C=C_, ldx=ldx, Ho=Ho, Wo=Wo, K=K, ldy=ldy, R=k, S=k, stride=stride,
                    dil=dil, pad_t=pt, pad_l=pl, dtype=dtype, out_dtype=dtype if out_dtype is None else out_dtype)


def debug_set(key: int, value: int):
    """Test/debug knobs of libodtk (include/odtk.h: odtk_debug_set)."""
    call("odtk_debug_set", int(key), int(value))


def scratch_slot(slot: int):
    """Select the library scratch slot of the following launches (include/odtk.h: odtk_scratch_slot): one per concurrently used stream."""
    call("odtk_scratch_slot", int(slot))


def conv_last_kernel() -> str:
    """Device kernel the last conv2d_* call dispatched to."""
    return _lib.load().odtk_conv_last_kernel().decode()


# ------------------------------------------------------------------ conv family
def conv2d_fwd(d: ConvDesc, x, w, bias, y, relu: bool):
    call("odtk_conv2d_fwd", C.byref(d), _p(x), _p(w), _p(bias), _p(y), int(relu), _stream())


def conv2d_fwd_pool2x2(d: ConvDesc, x, w, bias, y, relu: bool, y_pool, idx):
    """conv (+ bias, ReLU) and the 2x2 / stride-2 SAME max pool behind it in one launch where libodtk can (include/odtk.h); y may be None: the
    un-pooled output is then never stored"""
    call("odtk_conv2d_fwd_pool2x2", C.byref(d), _p(x), _p(w), _p(bias), _p(y), int(relu), _p(y_pool), int(y_pool.shape[-1]), _p(idx), _stream())


def conv2d_fwd_pool2x2_fused(d: ConvDesc) -> bool:
    return bool(_lib.load().odtk_conv2d_fwd_pool2x2_fused(C.byref(d)))


def conv2d_dgrad(d: ConvDesc, dy, lddy: int, w_t, relu_src, dx, accumulate: bool):
    call("odtk_conv2d_dgrad", C.byref(d), _p(dy), int(lddy), _p(w_t), _p(relu_src), _p(dx), int(accumulate),
         _stream())


def conv2d_relu_bits_supported(producer: ConvDesc, consumer: ConvDesc, consumer_lddy: int) -> bool:
    """ReLU mask as sign bits between `producer`'s forward pass and `consumer`'s input-gradient pass (include/odtk.h)"""
    return bool(_lib.load().odtk_conv2d_relu_bits_supported(C.byref(producer), C.byref(consumer), int(consumer_lddy)))


def conv2d_fwd_bits(d: ConvDesc, x, w, bias, y, relu: bool, relu_bits):
    call("odtk_conv2d_fwd_bits", C.byref(d), _p(x), _p(w), _p(bias), _p(y), int(relu), _p(relu_bits), _stream())


def conv2d_dgrad_bits(d: ConvDesc, dy, lddy: int, w_t, relu_bits, dx, accumulate: bool):
    call("odtk_conv2d_dgrad_bits", C.byref(d), _p(dy), int(lddy), _p(w_t), _p(relu_bits), _p(dx), int(accumulate), _stream())


def conv2d_x3_supported(d: ConvDesc) -> int:
    """bit 0: the forward pass of this F32X3 descriptor runs as split bf16 products, bit 1: the filter gradient, bit 2: the input gradient (include/odtk.h)"""
    return int(_lib.load().odtk_conv2d_x3_supported(C.byref(d)))


def conv2d_wgrad(d: ConvDesc, x, dy, lddy: int, dw, dbias=None):
    call("odtk_conv2d_wgrad", C.byref(d), _p(x), _p(dy), int(lddy), _p(dw), _p(dbias), _stream())


def filter_prepare(w, K, R, S, C_, Kp, dtype, w_c, w_t):
    call("odtk_filter_prepare", _p(w), K, R, S, C_, Kp, dtype, _p(w_c), _p(w_t), _stream())


class FilterPrepareBatch:
    """All dgrad-layout filters of a model refreshed by ONE launch (odtk_filter_prepare_batched).
    entries: list of (w f32 view [K*RS*C], w_t tensor, K, R, S, C, Kp)."""

    def __init__(self, entries, dtype, device):
        import struct
        blob = b''
        begin = 0
        for (w, wt, K, R, S, C_, Kp) in entries:
            ct, kt = (C_ + 31) // 32, (Kp + 63) // 64
            blob += struct.pack('<QQiiiiiiii', w.data_ptr(), wt.data_ptr(), K, R * S, C_, Kp, begin, ct, kt, 0)
            begin += kt * R * S * ct
        self.keep = entries                                   # the item table holds raw pointers
        self.n, self.blocks, self.dtype = len(entries), begin, dtype
        self.items = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)

    def run(self):
        call("odtk_filter_prepare_batched", _p(self.items), self.n, self.blocks, self.dtype, _stream())


# ------------------------------------------------------------------ layers
def preprocess(images, mean3, ldx, dtype, x):
    m = (C.c_float * 3)(*mean3)
    call("odtk_preprocess", _p(images), images.numel() // 3, m, ldx, dtype, _p(x), _stream())


def preprocess_norm(images, div, mean3, std3, ldx, dtype, x):
    """x = (images / div - mean) / std per channel (CenterNet.py:63)"""
    m, s = (C.c_float * 3)(*mean3), (C.c_float * 3)(*std3)
    call("odtk_preprocess_norm", _p(images), images.numel() // 3, float(div), m, s, ldx, dtype, _p(x), _stream())


def add_relu_fwd(a, lda, b, ldb, y, ldy, M, C_):
    call("odtk_add_relu_fwd", _p(a), int(lda), _p(b), int(ldb), _p(y), int(ldy), int(M), int(C_), dt_of(y), _stream())


def relu_bwd(y, dy, ldy, dx, lddx, M, C_, accumulate=False):
    call("odtk_relu_bwd", _p(y), _p(dy), int(ldy), _p(dx), int(lddx), int(M), int(C_), dt_of(y), int(accumulate), _stream())


def avgpool2x2_fwd(x, y, N, H, W, ld):
    call("odtk_avgpool2x2_fwd", _p(x), _p(y), N, H, W, ld, dt_of(x), _stream())


def avgpool2x2_bwd(dy, dx, N, H, W, ld):
    call("odtk_avgpool2x2_bwd", _p(dy), _p(dx), N, H, W, ld, dt_of(dy), _stream())


def maxpool_fwd(x, y, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    call("odtk_maxpool_fwd", _p(x), _p(y), N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l, dt_of(x), _stream())


def maxpool_bwd(x, y, dy, dx, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    call("odtk_maxpool_bwd", _p(x), _p(y), _p(dy), _p(dx), N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l,
         dt_of(x), _stream())


def maxpool_fwd_argmax(x, y, arg, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    """max pooling (k <= 3, any stride: pool5's overlapping 3x3 / stride-1 windows) that records the arg-max (arg: int32 per 16-byte output chunk)"""
    call("odtk_maxpool_fwd_argmax", _p(x), _p(y), _p(arg), N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l, dt_of(x), _stream())


def maxpool_bwd_argmax(arg, dy, dx, N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l):
    call("odtk_maxpool_bwd_argmax", _p(arg), _p(dy), _p(dx), N, H, W, C_, ld, Ho, Wo, k, stride, pad_t, pad_l, dt_of(dy), _stream())


def maxpool2x2_fwd_idx(x, y, idx, N, H, W, C_, ld, Ho, Wo):
    """2x2/s2 SAME pooling that also records the arg-max (idx: uint16 per 16-byte output chunk)."""
    call("odtk_maxpool2x2_fwd_idx", _p(x), _p(y), _p(idx), N, H, W, C_, ld, Ho, Wo, dt_of(x), _stream())


def maxpool2x2_bwd_idx(idx, dy, dx, N, H, W, C_, ld, Ho, Wo):
    call("odtk_maxpool2x2_bwd_idx", _p(idx), _p(dy), _p(dx), N, H, W, C_, ld, Ho, Wo, dt_of(dy), _stream())


def bn_workspace_bytes(M, C_):
    return int(_lib.load().odtk_bn_workspace_bytes(M, C_))


def bn_fwd(z, M, C_, ldz, gamma, beta, mmean, mvar, save_mean, save_invstd, training, relu, y, ldy,
           rows_per_img, y_img_stride, ws):
    call("odtk_bn_fwd", _p(z), M, C_, ldz, dt_of(z), _p(gamma), _p(beta), _p(mmean), _p(mvar), _p(save_mean),
         _p(save_invstd), int(training), int(relu), _p(y), dt_of(y), ldy, rows_per_img, y_img_stride, _p(ws),
         _stream())


def bn_bwd(z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, dz,
           dgamma, dbeta, ws):
    call("odtk_bn_bwd", _p(z), _p(y), _p(dy), M, C_, ldz, dt_of(z), dt_of(dy), ldy, rows_per_img, y_img_stride,
         _p(gamma), _p(save_mean), _p(save_invstd), int(relu), _p(dz), _p(dgamma), _p(dbeta), _p(ws), _stream())


class SyncBN:
    """Batch norm over the GLOBAL batch of a data-parallel job (include/odtk.h: odtk_bn_moments ... odtk_bn_bwd_given): the
    replicas exchange [2C] floats per layer and pass, forward (all-gather, done as an all-reduce of a zero-padded buffer so that
    it also runs on the gloo backend) and backward (all-reduce).  Same arguments as bn_fwd / bn_bwd."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.buf = {}

    def _scratch(self, C_, device):
        key = (C_, str(device))
        if key not in self.buf:
            self.buf[key] = (torch.zeros(self.world, 2, C_, device=device), torch.zeros(2 * C_, device=device),
                             torch.zeros(2 * C_, device=device))
        return self.buf[key]

    def fwd(self, z, M, C_, ldz, gamma, beta, mmean, mvar, save_mean, save_invstd, relu, y, ldy, rows_per_img, y_img_stride, ws):
        mom, _, _ = self._scratch(C_, z.device)
        mom.zero_()
        mine = mom[self.rank]
        call("odtk_bn_moments", _p(z), M, C_, ldz, dt_of(z), _p(mine[0]), _p(mine[1]), _p(ws), _stream())
        self.dist.all_reduce(mom, op=self.dist.ReduceOp.SUM, group=self.group)          # rows of the other ranks were zero: a gather
        call("odtk_bn_fwd_given", _p(z), M, C_, ldz, dt_of(z), _p(gamma), _p(beta), _p(mom), self.world, _p(mmean), _p(mvar),
             _p(save_mean), _p(save_invstd), int(relu), _p(y), dt_of(y), ldy, rows_per_img, y_img_stride, _p(ws), _stream())

    def bwd(self, z, y, dy, M, C_, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, dz, dgamma, dbeta, ws):
        _, sums, glob = self._scratch(C_, z.device)
        call("odtk_bn_bwd_sums", _p(z), _p(y), _p(dy), M, C_, ldz, dt_of(z), dt_of(dy), ldy, rows_per_img, y_img_stride, _p(save_mean),
             _p(save_invstd), int(relu), _p(sums), _p(ws), _stream())
        dbeta.copy_(sums[:C_]); dgamma.copy_(sums[C_:])          # the LOCAL sums: the gradient all-reduce adds the replicas' up
        glob.copy_(sums)
        self.dist.all_reduce(glob, op=self.dist.ReduceOp.SUM, group=self.group)
        call("odtk_bn_bwd_given", _p(z), _p(y), _p(dy), M, C_, ldz, dt_of(z), dt_of(dy), ldy, rows_per_img, y_img_stride, _p(gamma),
             _p(save_mean), _p(save_invstd), int(relu), _p(glob), int(M) * self.world, _p(dz), _p(ws), _stream())


def add2d(a, lda, b, ldb, y, ldy, M, C_):
    """y[:, :C] = a[:, :C] (+ b[:, :C]); operands are 2-D views with their own pitch (a channel slice of a concat buffer)"""
    call("odtk_add2d", _p(a), int(lda), _p(b), int(ldb), _p(y), int(ldy), int(M), int(C_), dt_of(a), _stream())


def upsample2x_fwd(x, ldx, y, ldy, N, H, W, C_):
    call("odtk_upsample2x_fwd", _p(x), int(ldx), _p(y), int(ldy), N, H, W, int(C_), dt_of(x), _stream())


def upsample2x_bwd(dy, lddy, dx, lddx, N, H, W, C_, accumulate=False):
    call("odtk_upsample2x_bwd", _p(dy), int(lddy), _p(dx), int(lddx), N, H, W, int(C_), dt_of(dy), int(accumulate), _stream())


def resize_bilinear_fwd(x, ldx, y, ldy, N, H, W, Ho, Wo, C_, accumulate=False):
    """tf.image.resize_bilinear (TF-1.x grid, align_corners=False) on NHWC rows; accumulate: y += resize(x)"""
    call("odtk_resize_bilinear_fwd", _p(x), int(ldx), _p(y), int(ldy), N, H, W, Ho, Wo, int(C_), dt_of(x), int(accumulate), _stream())


def resize_bilinear_bwd(dy, lddy, dx, lddx, N, H, W, Ho, Wo, C_, accumulate=False):
    call("odtk_resize_bilinear_bwd", _p(dy), int(lddy), _p(dx), int(lddx), N, H, W, Ho, Wo, int(C_), dt_of(dy), int(accumulate), _stream())


def resize_bilinear2_fwd(x, ldx, y, ldy, N, H, W, Ho, Wo, C_, align_corners, accumulate=False):
    """tf.image.resize_bilinear with the align_corners switch, any scaling (include/odtk.h: odtk_resize_bilinear2_fwd)"""
    call("odtk_resize_bilinear2_fwd", _p(x), int(ldx), _p(y), int(ldy), N, H, W, Ho, Wo, int(C_), dt_of(x), int(align_corners), int(accumulate), _stream())


def resize_bilinear2_bwd(dy, lddy, dx, lddx, N, H, W, Ho, Wo, C_, align_corners, accumulate=False, relu_src=None):
    call("odtk_resize_bilinear2_bwd", _p(dy), int(lddy), _p(dx), int(lddx), N, H, W, Ho, Wo, int(C_), dt_of(dy), int(align_corners), int(accumulate),
         _p(relu_src), _stream())


def copy_channels(src, lds, src_off, dst, ldd, dst_off, M, C_, accumulate=False, relu_src=None):
    """dst[:, dst_off : dst_off + C] (+)= src[:, src_off : src_off + C], element granular (include/odtk.h: odtk_copy_channels)"""
    call("odtk_copy_channels", _p(src), int(lds), int(src_off), _p(dst), int(ldd), int(dst_off), int(M), int(C_), dt_of(src), int(accumulate),
         _p(relu_src), _stream())


def gn_workspace(N, C_, device):
    return torch.zeros(int(_lib.load().odtk_gn_workspace_bytes(N, C_)), dtype=torch.uint8, device=device)


def gn_fwd(x, ldx, y, ldy, N, HW, C_, groups, gamma, beta, relu, save):
    """tf.contrib.layers.group_norm (+ ReLU) on NHWC rows; save [N, groups, 2] = mean, rstd (None in inference)"""
    call("odtk_gn_fwd", _p(x), int(ldx), _p(y), int(ldy), N, HW, int(C_), int(groups), dt_of(x), _p(gamma), _p(beta), int(relu), _p(save), _stream())


def gn_bwd(x, ldx, y, dy, ldy, dx, lddx, N, HW, C_, groups, gamma, save, relu, accumulate, dgamma, dbeta, ws):
    call("odtk_gn_bwd", _p(x), int(ldx), _p(y), _p(dy), int(ldy), _p(dx), int(lddx), N, HW, int(C_), int(groups), dt_of(x), _p(gamma), _p(save),
         int(relu), int(accumulate), _p(dgamma), _p(dbeta), _p(ws), _stream())


def exp_rows_to_f32(x, ldx, y, M, C_):
    """y f32 [M, C] = exp(x rows): the FCOS distance outputs (FCOS.py:363) as odtk_fcos_loss reads them"""
    call("odtk_exp_rows_to_f32", _p(x), int(ldx), dt_of(x), _p(y), int(M), int(C_), _stream())


def exp_rows_bwd(dy, y, dx, lddx, M, C_):
    call("odtk_exp_rows_bwd", _p(dy), _p(y), _p(dx), int(lddx), dt_of(dx), int(M), int(C_), _stream())


def rows_to_f32(x, ldx, y, ldy, rows_per_img, y_img_stride, M, C_):
    """conv rows [M][ldx] -> f32 prediction tensor (row m of image m // rows_per_img at y + n * y_img_stride + (m % rows_per_img) * ldy)"""
    call("odtk_rows_to_f32", _p(x), int(ldx), dt_of(x), _p(y), int(ldy), int(rows_per_img), int(y_img_stride), int(M), int(C_), _stream())


def rows_from_f32(y, ldy, rows_per_img, y_img_stride, x, ldx, M, C_):
    call("odtk_rows_from_f32", _p(y), int(ldy), int(rows_per_img), int(y_img_stride), _p(x), int(ldx), dt_of(x), int(M), int(C_), _stream())


def l2norm_fwd(x, y, M, C_, ld, gamma):
    call("odtk_l2norm_fwd", _p(x), _p(y), M, C_, ld, dt_of(x), _p(gamma), _stream())


def l2norm_bwd(x, dy, dx, M, C_, ld, gamma, dgamma, accumulate, relu_src):
    call("odtk_l2norm_bwd", _p(x), _p(dy), _p(dx), M, C_, ld, dt_of(x), _p(gamma), _p(dgamma), int(accumulate),
         _p(relu_src), _stream())


def colsum(dy, M, C_, ld, out, accumulate, ws):
    call("odtk_colsum", _p(dy), M, C_, ld, dt_of(dy), _p(out), int(accumulate), _p(ws), _stream())


def sgd_blocks(n):
    return int(_lib.load().odtk_sgd_blocks(n))


def sgd_momentum(p, m, g, lr, momentum, wd, grad_scale, l2_partial, p_cast):
    call("odtk_sgd_momentum", _p(p), _p(m), _p(g), p.numel(), float(lr), float(momentum), float(wd),
         float(grad_scale), _p(l2_partial), _p(p_cast), dt_of(p_cast) if p_cast is not None else BF16, _stream())


def adam(p, m, v, g, lr_t, beta1, beta2, eps, wd, grad_scale, l2_partial, p_cast):
    """fused tf.train.AdamOptimizer step + L2 term (include/odtk.h: odtk_adam); lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)"""
    call("odtk_adam", _p(p), _p(m), _p(v), _p(g), p.numel(), float(lr_t), float(beta1), float(beta2), float(eps), float(wd),
         float(grad_scale), _p(l2_partial), _p(p_cast), dt_of(p_cast) if p_cast is not None else BF16, _stream())


def sum_f32(x, out):
    call("odtk_sum_f32", _p(x), x.numel(), _p(out), _stream())


def zero(x):
    """x <- 0 on the current stream (no torch launch)"""
    call("odtk_zero", _p(x), x.numel() * x.element_size(), _stream())


def loss_total(a, na, stride_a, b, scale_a, scale_b, sum_a, sum_b, total):
    """total[0] = scale_a * sum_i a[i * stride_a] + scale_b * sum(b); the two sums also go to sum_a / sum_b (1-element f32 tensors or None)"""
    call("odtk_loss_total", _p(a), int(na), int(stride_a), _p(b), b.numel(), float(scale_a), float(scale_b), _p(sum_a), _p(sum_b), _p(total), _stream())


def cast_from_f32(x, out):
    call("odtk_cast_from_f32", _p(x), _p(out), x.numel(), dt_of(out), _stream())


def cast_to_f32(x, out):
    call("odtk_cast_to_f32", _p(x), dt_of(x), _p(out), x.numel(), _stream())


# ------------------------------------------------------------------ box side
def ssd_priors(input_size, fsizes, nas, prior_hw_flat, device):
    """Returns (y1x1, y2x2, yx, hw, nmsbox) device tensors."""
    A = sum(f * f * a for f, a in zip(fsizes, nas))
    mk = lambda c: torch.empty(A, c, dtype=torch.float32, device=device)
    y1x1, y2x2, yx, hw, nb = mk(2), mk(2), mk(2), mk(2), mk(4)
    L = len(fsizes)
    call("odtk_ssd_priors", int(input_size), L, (C.c_int * L)(*fsizes), (C.c_int * L)(*nas),
         (C.c_float * len(prior_hw_flat))(*prior_hw_flat), _p(y1x1), _p(y2x2), _p(yx), _p(hw), _p(nb), _stream())
    return y1x1, y2x2, yx, hw, nb


def ssd_match(y1x1, y2x2, hw, gt, ngt, best, status, rgindex, counts):
    N, P, _ = gt.shape
    call("odtk_ssd_match", _p(y1x1), _p(y2x2), _p(hw), y1x1.shape[0], _p(gt), N, P, _p(ngt), _p(best),
         _p(status), _p(rgindex), _p(counts), _stream())


def softmax_ce_const(pred, rows, Cn, ld, label, loss):
    call("odtk_softmax_ce_const", _p(pred), rows, Cn, ld, label, _p(loss), _stream())


def nms_batched(boxes, box_stride, scores, score_bstride, score_estride, valid, valid_bstride, valid_estride,
                valid_value, n, B, max_out_dev, max_out_stride, max_out_const, iou_thr, out_idx, cap, out_cnt):
    call("odtk_nms_batched", _p(boxes), box_stride, _p(scores), score_bstride, score_estride, _p(valid),
         valid_bstride, valid_estride, valid_value, n, B, _p(max_out_dev), max_out_stride, max_out_const,
         float(iou_thr), _p(out_idx), cap, _p(out_cnt), _stream())


def ssd_loss(pred, Cn, yx, hw, gt, ngt, best, status, rgindex, counts, negloss, sel_idx, sel_cnt, grad_scale,
             loss_parts, dpred):
    N, A, ld = pred.shape
    call("odtk_ssd_loss", _p(pred), N, A, Cn, ld, _p(yx), _p(hw), _p(gt), gt.shape[1], _p(ngt), _p(best),
         _p(status), _p(rgindex), _p(counts), _p(negloss), _p(sel_idx), sel_idx.shape[1], _p(sel_cnt),
         float(grad_scale), _p(loss_parts), _p(dpred), _stream())


def ssd_decode(pred0, Cn, yx, hw, thr, conf, boxes, keep, cand):
    A, ld = pred0.shape
    call("odtk_ssd_decode", _p(pred0), A, Cn, ld, _p(yx), _p(hw), float(thr), _p(conf), _p(boxes), _p(keep),
         _p(cand), _stream())


# ------------------------------------------------------------------ RetinaNet box side (K16)
def retina_anchors(input_dim, shapes, nas, prior_hw_flat, device):
    """shapes: [(fh, fw)] per level; returns (y1x1, y2x2, yx, hw) device tensors [A, 2]."""
    A = sum(fh * fw * a for (fh, fw), a in zip(shapes, nas))
    mk = lambda: torch.empty(A, 2, dtype=torch.float32, device=device)
    y1x1, y2x2, yx, hw = mk(), mk(), mk(), mk()
    L = len(shapes)
    call("odtk_retina_anchors", int(input_dim), L, (C.c_int * L)(*[s[0] for s in shapes]),
         (C.c_int * L)(*[s[1] for s in shapes]), (C.c_int * L)(*nas), (C.c_float * len(prior_hw_flat))(*prior_hw_flat),
         _p(y1x1), _p(y2x2), _p(yx), _p(hw), _stream())
    return y1x1, y2x2, yx, hw


def retina_match_workspace(A, N, P, device):
    return torch.empty(int(_lib.load().odtk_retina_match_workspace_bytes(A, N, P)), dtype=torch.uint8, device=device)


def retina_match(y1x1, y2x2, hw, gt, ngt, best, status, rgindex, counts, ws):
    N, P, _ = gt.shape
    call("odtk_retina_match", _p(y1x1), _p(y2x2), _p(hw), y1x1.shape[0], _p(gt), N, P, _p(ngt), _p(best), _p(status),
         _p(rgindex), _p(counts), _p(ws), _stream())


def retina_loss(pconf, pbox, yx, hw, gt, ngt, best, status, rgindex, counts, alpha, gamma, grad_scale, loss_parts,
                dconf, dbox):
    N, A, Cn = pconf.shape
    call("odtk_retina_loss", _p(pconf), _p(pbox), N, A, Cn, _p(yx), _p(hw), _p(gt), gt.shape[1], _p(ngt), _p(best),
         _p(status), _p(rgindex), _p(counts), float(alpha), float(gamma), float(grad_scale), _p(loss_parts), _p(dconf),
         _p(dbox), _stream())


def refinedet_loss(arm_loc, arm_conf, odm_loc, odm_conf, yx, hw, gt, ngt, best, status, rgindex, counts, negloss, sel_idx, sel_cnt, grad_scale,
                   loss_parts, d_arm_loc, d_arm_conf, d_odm_loc, d_odm_conf):
    """RefineDet.py:422-567 for the batch (include/odtk.h: odtk_refinedet_loss); loss_parts [N, 8]"""
    N, A, Cn = odm_conf.shape
    call("odtk_refinedet_loss", _p(arm_loc), _p(arm_conf), _p(odm_loc), _p(odm_conf), N, A, Cn, _p(yx), _p(hw), _p(gt), gt.shape[1], _p(ngt),
         _p(best), _p(status), _p(rgindex), _p(counts), _p(negloss), _p(sel_idx), sel_idx.shape[1], _p(sel_cnt), float(grad_scale), _p(loss_parts),
         _p(d_arm_loc), _p(d_arm_conf), _p(d_odm_loc), _p(d_odm_conf), _stream())


def refinedet_decode(arm_loc, arm_conf, odm_loc, odm_conf, yx, hw, thr):
    """RefineDet.py:189-206 for one image -> conf [A, C-1], boxes [A, 4], keep [A], cand [A, C-1]"""
    A, Cn = odm_conf.shape
    dev = odm_conf.device
    conf = torch.empty(A, Cn - 1, device=dev)
    boxes = torch.empty(A, 4, device=dev)
    keep = torch.empty(A, dtype=torch.uint8, device=dev)
    cand = torch.empty(A, Cn - 1, dtype=torch.uint8, device=dev)
    call("odtk_refinedet_decode", _p(arm_loc), _p(arm_conf), _p(odm_loc), _p(odm_conf), A, Cn, _p(yx), _p(hw), float(thr), _p(conf), _p(boxes),
         _p(keep), _p(cand), _stream())
    return conf, boxes, keep, cand


def retina_decode(pconf, pbox, yx, hw, thr):
    """RetinaNet.py:224-238 for one image: pconf [A,C], pbox [A,4] -> conf [A,C-1], boxes [A,4], keep [A], cand [A,C-1]."""
    A, Cn = pconf.shape
    dev = pconf.device
    conf = torch.empty(A, Cn - 1, device=dev)
    boxes = torch.empty(A, 4, device=dev)
    keep = torch.empty(A, dtype=torch.uint8, device=dev)
    cand = torch.empty(A, Cn - 1, dtype=torch.uint8, device=dev)
    call("odtk_retina_decode", _p(pconf), _p(pbox), A, Cn, _p(yx), _p(hw), float(thr), _p(conf), _p(boxes), _p(keep), _p(cand), _stream())
    return conf, boxes, keep, cand


# ---------------------------------------------------------------------------------------------------------
# CenterNet / FCOS box side (include/odtk.h; csrc/dense_heads.hip)
# ---------------------------------------------------------------------------------------------------------
def centernet_workspace(N, H, W, Cn, device):
    return torch.empty(int(_lib.load().odtk_centernet_workspace_bytes(N, H, W, Cn)), dtype=torch.uint8, device=device)


def centernet_loss(keypoints, offset, size, gt, stride, grad_scale, loss_parts, d_keypoints, d_offset, d_size, ws):
    """CenterNet.py:187-251 for the whole batch; loss_parts [N,4] = keypoint, offset, size, total."""
    N, H, W, Cn = keypoints.shape
    call("odtk_centernet_loss", _p(keypoints), _p(offset), _p(size), _p(gt), N, H, W, Cn, gt.shape[1], float(stride),
         float(grad_scale), _p(loss_parts), _p(d_keypoints), _p(d_offset), _p(d_size), _p(ws), _stream())


def centernet_decode(keypoints, offset, size, stride, score_threshold, top_k, ws):
    """CenterNet.py:159-185 for one image [H,W,C]; returns (scores, bbox, class_id) trimmed to the kept count."""
    H, W, Cn = keypoints.shape
    dev = keypoints.device
    scores = torch.empty(top_k, device=dev)
    bbox = torch.empty(top_k, 4, device=dev)
    cls = torch.empty(top_k, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    call("odtk_centernet_decode", _p(keypoints), _p(offset), _p(size), H, W, Cn, float(stride), float(score_threshold),
         int(top_k), _p(scores), _p(bbox), _p(cls), _p(cnt), _p(ws), _stream())
    k = int(cnt.item())
    return scores[:k], bbox[:k], cls[:k]


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _fcos_shapes(conf):
    flat = []
    for c in conf:
        flat += [c.shape[-3], c.shape[-2]]
    return (C.c_int * len(flat))(*flat)


def fcos_workspace(conf, N, device):
    return torch.empty(int(_lib.load().odtk_fcos_workspace_bytes(_fcos_shapes(conf), N)), dtype=torch.uint8, device=device)


def fcos_loss(conf, reg, center, gt, grad_scale, loss, d_conf, d_reg, d_center, ws):
    """FCOS.py:153-189 + :266-348.  conf / reg / center: lists of the five level tensors [N,H,W,C|4|1]."""
    N, Cn = conf[0].shape[0], conf[0].shape[-1]
    call("odtk_fcos_loss", _ptr_array(conf), _ptr_array(reg), _ptr_array(center), _fcos_shapes(conf), _p(gt), N, Cn,
         gt.shape[1], float(grad_scale), _p(loss), _ptr_array(d_conf), _ptr_array(d_reg), _ptr_array(d_center), _p(ws), _stream())


def fcos_decode_candidates(conf, reg, center):
    """FCOS.py:197-246 for one image (level tensors [H,W,C|4|1]); returns pconf [L,C], pbbox [L,4]."""
    dev = conf[0].device
    L = sum(c.shape[0] * c.shape[1] for c in conf)
    Cn = conf[0].shape[-1]
    pconf = torch.empty(L, Cn, device=dev)
    pbbox = torch.empty(L, 4, device=dev)
    call("odtk_fcos_decode_candidates", _ptr_array(conf), _ptr_array(reg), _ptr_array(center), _fcos_shapes(conf), Cn,
         _p(pconf), _p(pbbox), _stream())
    return pconf, pbbox


# ---------------------------------------------------------------------------------------------------------
# YOLOv3 box side (include/odtk.h; csrc/dense_heads.hip)
# ---------------------------------------------------------------------------------------------------------
def _yolo_shapes(preds):
    flat = []
    for p in preds:
        flat += [p.shape[-4], p.shape[-3]]
    return (C.c_int * len(flat))(*flat)


def _farr(values):
    return (C.c_float * len(values))(*[float(v) for v in values])


def yolov3_workspace(preds, N, device):
    P = preds[0].shape[-2]
    return torch.empty(max(4, int(_lib.load().odtk_yolov3_workspace_bytes(_yolo_shapes(preds), P, N))), dtype=torch.uint8, device=device)


def yolov3_loss(preds, priors_flat, head_stride, gt, scales, grad_scale, loss_parts, d_preds, ws):
    """YOLOv3.py:117-311.  preds: three [N,H,W,P,C+5] tensors (head 1 = coarsest); priors_flat: 3*P*2 floats (h, w) in
    head units; scales = (coord, noobj, obj, class); loss_parts [N,5]."""
    N, P, E = preds[0].shape[0], preds[0].shape[-2], preds[0].shape[-1]
    call("odtk_yolov3_loss", _ptr_array(preds), _yolo_shapes(preds), _farr(priors_flat), _farr(head_stride), _p(gt), N, P, E - 5,
         gt.shape[1], float(scales[0]), float(scales[1]), float(scales[2]), float(scales[3]), float(grad_scale), _p(loss_parts),
         _ptr_array(d_preds), _p(ws), _stream())


def yolov2_loss(pred, priors_flat, stride, gt, scales, grad_scale, loss_parts, d_pred):
    """YOLOv2.py:102-167.  pred [N,H,W,P,C+5]; priors_flat: P*2 floats (h, w) in cell units; scales = (coord, noobj, obj, class); loss_parts [N,5]."""
    N, H, W, P, E = pred.shape
    call("odtk_yolov2_loss", _p(pred), N, H, W, P, E - 5, _farr(priors_flat), float(stride), _p(gt), gt.shape[1], float(scales[0]), float(scales[1]),
         float(scales[2]), float(scales[3]), float(grad_scale), _p(loss_parts), _p(d_pred), _stream())


def yolov2_decode_candidates(pred0, priors_flat, stride):
    """YOLOv2.py:177-186 for one image [H,W,P,C+5]; returns confidence [L,C], bbox [L,4] (pixels)."""
    H, W, P, E = pred0.shape
    conf = torch.empty(H * W * P, E - 5, device=pred0.device)
    bbox = torch.empty(H * W * P, 4, device=pred0.device)
    call("odtk_yolov2_decode_candidates", _p(pred0), H, W, P, E - 5, _farr(priors_flat), float(stride), _p(conf), _p(bbox), _stream())
    return conf, bbox


def yolov3_decode_candidates(preds, priors_flat, decode_scale):
    """YOLOv3.py:320-350 for one image (three [H,W,P,C+5] tensors); returns confidence [L,C], bbox [L,4]."""
    dev = preds[0].device
    P, E = preds[0].shape[-2], preds[0].shape[-1]
    L = sum(p.shape[0] * p.shape[1] * P for p in preds)
    conf = torch.empty(L, E - 5, device=dev)
    bbox = torch.empty(L, 4, device=dev)
    call("odtk_yolov3_decode_candidates", _ptr_array(preds), _yolo_shapes(preds), _farr(priors_flat), _farr(decode_scale), P, E - 5,
         _p(conf), _p(bbox), _stream())
    return conf, bbox


# ------------------------------------------------------------------ Light-Head R-CNN (csrc/lhrcnn.hip)
def depthwise_conv(x, ldx, filt, y, ldy, N, H, W, C_, kh, kw, flip=False, accumulate=False):
    """depthwise half of tf.layers.separable_conv2d (stride 1, SAME); flip: the input gradient; filt: f32 [kh][kw][C]"""
    call("odtk_depthwise_conv", _p(x), ldx, _p(filt), _p(y), ldy, N, H, W, C_, kh, kw, int(flip), int(accumulate), dt_of(x), _stream())


def depthwise_wgrad(x, ldx, dy, lddy, dfilt, N, H, W, C_, kh, kw):
    call("odtk_depthwise_wgrad", _p(x), ldx, _p(dy), lddy, _p(dfilt), N, H, W, C_, kh, kw, dt_of(x), _stream())


def lhrcnn_match(anc, conf, gt, ws):
    """anc: dict(y1x1, y2x2, yx, hw [A,2] f32, row [A] i32, A_full); conf [N, A_full, 2]; gt [N, P, 5]; ws: lhrcnn_workspace(...)"""
    N, P = gt.shape[0], gt.shape[1]
    A = anc['yx'].shape[0]
    call("odtk_lhrcnn_match", _p(anc['y1x1']), _p(anc['y2x2']), _p(anc['yx']), _p(anc['hw']), _p(anc['row']), A, anc['A_full'], _p(conf), _p(gt), N, P,
         ws['cap'], _p(ws['counts']), _p(ws['status']), _p(ws['pos_anchor']), _p(ws['pos_gt']), _p(ws['pos_label']), _p(ws['pos_score']), _p(ws['pos_box']),
         _p(ws['pos_valid']), _p(ws['neg_anchor']), _p(ws['neg_score']), _p(ws['neg_box']), _p(ws['neg_valid']), _stream())


def lhrcnn_rpn_loss(anc, conf, bbox, gt, ws, num_classes, grad_scale, img_h, img_w, d_conf, d_bbox):
    N, P = gt.shape[0], gt.shape[1]
    A = anc['yx'].shape[0]
    call("odtk_lhrcnn_rpn_loss", _p(anc['y1x1']), _p(anc['y2x2']), _p(anc['yx']), _p(anc['hw']), _p(anc['row']), A, anc['A_full'], _p(conf), _p(bbox), _p(gt),
         N, P, ws['cap'], num_classes, _p(ws['pos_anchor']), _p(ws['pos_gt']), _p(ws['pos_label']), _p(ws['neg_anchor']), _p(ws['sel_pos']), _p(ws['cnt_pos']),
         _p(ws['sel_neg']), _p(ws['cnt_neg']), float(grad_scale), int(img_h), int(img_w), _p(ws['rpn_parts']), _p(d_conf), _p(d_bbox), _p(ws['roi_box']),
         _p(ws['roi_prop']), _p(ws['roi_truth']), _p(ws['roi_img']), _p(ws['roi_label']), _p(ws['roi_kind']), _p(ws['roi_counts']), _stream())


def lhrcnn_workspace(N, A, P, device):
    """the buffers of one RPN loss evaluation: candidate lists (row pitch cap = A + P), NMS picks, the 256-row R-CNN slots of every image"""
    cap = A + P
    if cap > 32768:
        raise ValueError(f'LHRCNN: {A} anchors inside the picture + {P} ground-truth slots = {cap} NMS candidates per image; odtk_nms_batched takes at most 32768 '
                         '(smaller pictures, a larger stride or fewer anchor shapes)')
    i32, f32, u8 = torch.int32, torch.float32, torch.uint8
    z = lambda *s, dtype=f32: torch.zeros(*s, dtype=dtype, device=device)      # noqa: E731
    return dict(cap=cap, counts=z(N, 8, dtype=i32), status=z(N, A, dtype=u8),
                pos_anchor=z(N, cap, dtype=i32), pos_gt=z(N, cap, dtype=i32), pos_label=z(N, cap, dtype=i32), pos_score=z(N, cap), pos_box=z(N, cap, 4),
                pos_valid=z(N, cap, dtype=u8), neg_anchor=z(N, cap, dtype=i32), neg_score=z(N, cap), neg_box=z(N, cap, 4), neg_valid=z(N, cap, dtype=u8),
                sel_pos=z(N, 128, dtype=i32), cnt_pos=z(N, dtype=i32), sel_neg=z(N, 256, dtype=i32), cnt_neg=z(N, dtype=i32), rpn_parts=z(N, 4),
                roi_box=z(N * 256, 4), roi_prop=z(N * 256, 4), roi_truth=z(N * 256, 4), roi_img=z(N * 256, dtype=i32), roi_label=z(N * 256, dtype=i32),
                roi_kind=z(N * 256, dtype=i32), roi_counts=z(N, 2, dtype=i32), rcnn_parts=z(N, 2))


def crop_and_resize_fwd(feat, ldf, N, H, W, C_, boxes, box_img, crop, out, ldo):
    call("odtk_crop_and_resize_fwd", _p(feat), ldf, N, H, W, C_, _p(boxes), _p(box_img), boxes.shape[0], crop, _p(out), ldo, dt_of(feat), _stream())


def crop_and_resize_bwd(d_out, ldo, N, H, W, C_, boxes, box_img, crop, d_feat, ldf):
    """d_feat: f32 [N*H*W, ldf], zeroed by the call"""
    call("odtk_crop_and_resize_bwd", _p(d_out), ldo, N, H, W, C_, _p(boxes), _p(box_img), boxes.shape[0], crop, _p(d_feat), ldf, dt_of(d_out), _stream())


def lhrcnn_rcnn_loss(logits, ldl, pbbox, ldb, N, Cn, ws, grad_scale, d_logits, d_pbbox):
    call("odtk_lhrcnn_rcnn_loss", _p(logits), ldl, _p(pbbox), ldb, N, Cn, _p(ws['roi_label']), _p(ws['roi_kind']), _p(ws['roi_truth']), _p(ws['roi_counts']),
         float(grad_scale), _p(ws['rcnn_parts']), _p(d_logits), _p(d_pbbox), _stream())


def lhrcnn_rpn_decode(anc, conf0, bbox0, img_h, img_w, prop, score):
    call("odtk_lhrcnn_rpn_decode", _p(anc['yx']), _p(anc['hw']), _p(anc['row']), anc['yx'].shape[0], anc['A_full'], _p(conf0), _p(bbox0), int(img_h), int(img_w),
         _p(prop), _p(score), _stream())


def lhrcnn_gather_rois(prop, sel, cnt, img_h, img_w, roi_box, roi_prop, roi_img):
    call("odtk_lhrcnn_gather_rois", _p(prop), _p(sel), _p(cnt), roi_img.shape[0], int(img_h), int(img_w), _p(roi_box), _p(roi_prop), _p(roi_img), _stream())


def lhrcnn_rcnn_decode(logits, ldl, pbbox, ldb, roi_prop, roi_img, Cn, thr, conf, boxes, cand):
    call("odtk_lhrcnn_rcnn_decode", _p(logits), ldl, _p(pbbox), ldb, _p(roi_prop), _p(roi_img), roi_img.shape[0], Cn, float(thr), _p(conf), _p(boxes), _p(cand),
         _stream())
