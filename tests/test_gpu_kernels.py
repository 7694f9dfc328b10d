"""GPU parity tests, kernel level: every call goes through the C-ABI (libodtk.so) and is
checked against the CPU oracle / plain torch fp32 math on the same seeded inputs.

Tolerances: f32 path 2e-4 relative to the output scale (exact-f32 MFMA, different summation
order); bf16 path 2e-2 (operands rounded to bf16, f32 accumulate) against a reference fed the
same bf16-rounded operands."""
import math

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ssd300_ref as R  # noqa: E402


def _ops():
    import odtk  # noqa: F401
    from odtk import ops
    return ops


def to_rows(x_nhwc, ld, dtype, dev):
    """[N,H,W,C] cpu f32 -> device [N*H*W, ld] in dtype (zero padded)."""
    N, H, W, C = x_nhwc.shape
    t = torch.zeros(N * H * W, ld, dtype=torch.float32)
    t[:, :C] = x_nhwc.reshape(-1, C)
    return t.to(dtype).to(dev).contiguous()


def from_rows(t, N, H, W, C):
    return t[:, :C].float().cpu().reshape(N, H, W, C)


CONV_CASES = [
    # N, H, W, C, K, k, stride, dil
    (2, 19, 19, 32, 48, 3, 1, 1),
    (1, 10, 10, 16, 24, 3, 2, 1),     # stride 2, asymmetric SAME pad (pad_before 0)
    (2, 19, 19, 64, 128, 3, 1, 2),    # dilation 2 (conv6)
    (2, 9, 9, 256, 100, 3, 1, 1),     # head: Cout not a multiple of 8/32
    (2, 5, 5, 256, 150, 3, 1, 1),     # head: Cout = 150
    (2, 20, 20, 8, 64, 3, 1, 1),      # C=8 -> K tail inside a 16B-chunk slab (conv1_1 shape class)
    (2, 19, 19, 128, 64, 1, 1, 1),    # 1x1
    (3, 38, 38, 64, 64, 3, 1, 1),     # several pixel tiles, PT=64 path
    (1, 5, 5, 128, 256, 3, 2, 1),     # 5 -> 3
    (2, 21, 17, 40, 72, 3, 1, 1),     # non-square, C not a multiple of the k-slab
]


# geometries of the models that are not built yet: the reference's RetinaNet has 7 / 14 / 28 / 56 (x4) channels
# (filters_list from the kernel size, RetinaNet.py:27), a 7x7 / stride-2 stem and stride-2 3x3 shortcuts; CenterNet's 4x4 / stride-2
# transposed convolution is the dgrad of a 4x4 / stride-2 convolution
BACKBONE_CASES = [
    (2, 64, 64, 3, 16, 7, 2, 1),      # stem: 3 real channels in one chunk, 7x7, stride 2 (asymmetric SAME pad 2 / 3)
    # round 6: more 49-tap geometries (the stems of the ResNet / DLA classes; they run on the register-staged kernels: the LDS-DMA gathers' tap masks are 32 bits --
    # widened to 64 and measured, the stems were SLOWER there: CenterNet 535 -> 590 us, FCOS 141 -> 224, RetinaNet 760 -> 721)
    (2, 40, 52, 3, 16, 7, 1, 1),      # ... 7x7 / stride 1 (CenterNet's DLA stem), ragged tiles
    (2, 64, 64, 3, 64, 7, 2, 1),      # ... 64 output channels (the ResNet stem of FCOS)
    (1, 20, 22, 16, 24, 7, 1, 1),     # ... 16 input channels: the input gradient walks 49 taps x 24 channels too
    (2, 32, 32, 16, 7, 1, 1, 1),      # bottleneck 1x1 to 7 channels (pitch 8)
    (2, 32, 32, 7, 7, 3, 1, 1),       # 7 -> 7, 3x3
    (2, 32, 32, 7, 28, 1, 1, 1),      # 7 -> 28 (pitch 32)
    (2, 32, 32, 16, 28, 3, 1, 1),     # shortcut 16 -> 28, 3x3
    (2, 32, 32, 28, 14, 1, 1, 1),
    (2, 32, 32, 14, 14, 3, 2, 1),     # stride 2 inside a bottleneck
    (2, 32, 32, 28, 56, 3, 2, 1),     # stride-2 shortcut
    (1, 16, 16, 224, 256, 1, 1, 1),   # pyramid lateral on the last stage (56 * 4 channels)
    (2, 16, 16, 256, 180, 3, 1, 1),   # class subnet output, 9 anchors x 20 classes
    (2, 8, 8, 256, 36, 3, 1, 1),      # box subnet output, 9 x 4
    (2, 16, 16, 256, 256, 3, 2, 1),   # p6 / p7
    (2, 16, 16, 64, 64, 4, 2, 1),     # 4x4 / stride 2: its dgrad is tf.layers.conv2d_transpose(4, 2, 'same') (CenterNet.py:349-361)
    (2, 2, 2, 256, 36, 3, 1, 1),      # the coarsest pyramid levels at small inputs: 8 and 18 rows in all
    (2, 2, 2, 256, 189, 3, 1, 1),
    (2, 3, 3, 256, 256, 3, 2, 1),     # p7 from a 3 x 3 p6
    (2, 3, 3, 256, 256, 3, 1, 1),
    (1, 1, 1, 256, 256, 3, 1, 1),     # one pixel
    # round 4: maps of >= 4 096 pixels -- the narrow f32 filter-gradient kernel (pixel-major LDS-DMA tiles of 8 x 32 | 4 x 32 pixels, persistent workgroups) and
    # the 32-row filter tile of the f32 LDS-DMA gather; ragged tiles on both edges, every (KI, CI) instantiation
    (2, 64, 64, 28, 28, 3, 1, 1),     # 3x3, 28 -> 28 (one 32 x 32 tile per tap)
    (2, 50, 70, 7, 7, 3, 1, 1),       # 3x3, 7 -> 7: ragged tiles on both edges (50 = 6 x 8 + 2, 70 = 2 x 32 + 6)
    (1, 90, 47, 16, 28, 3, 1, 1),     # 3x3, 16 -> 28
    (2, 64, 64, 16, 7, 1, 1, 1),      # 1x1 (KI = CI = 1)
    (2, 50, 70, 7, 28, 1, 1, 1),      # 1x1, ragged
    (3, 40, 40, 28, 56, 1, 1, 1),     # 1x1, 56 output channels (KI = 2)
    (3, 40, 40, 56, 14, 1, 1, 1),     # 1x1, 56 input channels (CI = 2)
    (2, 48, 48, 56, 56, 1, 1, 1),     # 1x1, both (KI = CI = 2)
    (8, 128, 128, 3, 32, 3, 1, 1),    # several tiles per persistent workgroup (512 tiles), 3 input channels in one 16-byte chunk (a first layer)
    (8, 128, 136, 28, 28, 3, 1, 1),   # ... 28 -> 28, ragged right edge
    (12, 128, 128, 16, 8, 1, 1, 1),   # ... 1x1 (1 536 tiles of 4 x 32 pixels)
]


def _ref_conv(x_nhwc, w_krsc, b, stride, dil):
    y = R.conv2d_same(x_nhwc.permute(0, 3, 1, 2), w_krsc, b, stride, dil)
    return y.permute(0, 2, 3, 1).contiguous()


V3_EXTRA_CASES = [
    (2, 38, 38, 64, 128, 3, 1, 1),    # 12 pixel tiles of 256, 9 k-slabs
    (1, 19, 19, 1024, 256, 1, 1, 1),  # 1x1, 16 k-slabs, 2 channel tiles
    (1, 40, 40, 128, 64, 3, 1, 1),    # PT=64 tile, 18 k-slabs
    (2, 19, 19, 512, 100, 3, 1, 1),   # deep k (72 slabs), head-like Cout with a channel tail
    (8, 75, 75, 64, 256, 3, 1, 1),    # 176 x 2 tiles: persistent blocks own > 1 tile on a 256-CU chip
    (5, 150, 150, 16, 64, 3, 1, 1),   # 440 PT=64 tiles, 3 k-slabs
    (2, 13, 70, 64, 64, 3, 1, 1),     # resident-filter 64->64 kernel: ragged 8x32 tiles on both edges
    (40, 33, 65, 64, 64, 3, 1, 1),    # ... 600 tiles: persistent blocks walk several tiles (patch double buffering)
    (3, 38, 38, 128, 256, 3, 1, 1),   # raster-run halo kernel: tiles straddle rows AND images (4332 px = 16.9 tiles), 2 chunks
    (2, 75, 75, 64, 128, 3, 1, 1),    # ... widest supported map, one chunk
    (2, 19, 19, 128, 192, 3, 1, 2),   # ... dilation 2 (conv6 geometry), channel-tile tail
    (2, 19, 19, 64, 300, 3, 1, 1),    # 256-row wgrad tile with a channel tail (300 = 256 + 44)
    (1, 8, 8, 64, 96, 3, 1, 1),       # one partial pixel tile: the halo patch is mostly outside the tensor
    (30, 45, 70, 8, 64, 3, 1, 1),     # first-layer kernel (3 real channels in one 16-B chunk): 540 ragged tiles, several per block
    (1, 60, 150, 192, 160, 3, 1, 1),  # wide raster-run halo kernel (single patch buffer, W = 150): 3 chunks, channel-tile tail
    (1, 20, 159, 64, 128, 3, 1, 1),   # ... widest supported map: the patch needs all 576 rows
    (2, 80, 72, 128, 128, 3, 1, 2),   # ... dilation 2 (dil * W = 144: the early groups end exactly at the first live row)
    (1, 40, 128, 128, 128, 3, 1, 1),  # ... rows of 128 pixels (G1 = 4, G2 = 8), two chunks
    (1, 30, 104, 192, 64, 3, 1, 1),   # ... 104 (G1 = 3, G2 = 6), three chunks, 64-row filter tile falls back to the 8-wave kernel in forward
    (2, 24, 120, 128, 160, 3, 1, 1),  # ... 120 (G1 = 3, G2 = 7)
    (1, 21, 160, 128, 128, 3, 1, 1),  # ... 160 and 175: the 608-row patch (G1 = 5, G2 = 10)
    (1, 12, 175, 64, 130, 3, 1, 1),
    (2, 50, 80, 128, 128, 3, 1, 1),   # ... rows of 80-95 pixels (round 4: conv3_x of the 320-pixel models): 448-row patch (G1 = 2, G2 = 5), two chunks
    (2, 40, 40, 128, 160, 3, 1, 2),   # ... dilation 2 (dil * W = 80, 164 halo rows)
    (24, 75, 75, 64, 200, 3, 1, 1),   # 128 x 512 tiles of four 128 x 128 wave tiles (>= 2 rounds of 256 workgroups): ragged last tile, channel tail
    (24, 19, 19, 64, 512, 3, 1, 1),   # 128 x 192 tiles (136 workgroups of 256 pixels would leave CUs idle, 184 of 192 pixels fit one round)
    (6, 20, 17, 256, 512, 3, 1, 1),   # four-wave filter gradient (256 x 256 tiles, fixture wgrad-v8): two k tiles, nine column tiles, ragged last 32-pixel slab
    (5, 19, 23, 256, 256, 3, 2, 1),   # ... stride 2 with the asymmetric SAME pad (the pixel walk steps input rows / columns by 2)
    (7, 5, 3, 256, 256, 3, 1, 1),     # ... 15-pixel images: one 32-pixel slab spans three images (walk: dn = 2 images + 2 pixels)
    (2, 40, 64, 128, 128, 3, 1, 1),   # block-pair halo filter gradient (fixture wgrad-c64-pairs): 2 x 2 pairs, whole 8 x 32 tiles
    (3, 33, 95, 64, 192, 3, 1, 1),    # ... 1 x 3 pairs, ragged right / bottom tiles, fewer tiles than workgroups per pair
    (1, 17, 150, 192, 64, 3, 1, 1),   # ... 3 x 1 pairs at the conv2_x width
]


@pytest.fixture(params=[(2, 0), (2, 256 + 65536), (2, 16384 + 65536), (2, 8192 + 65536), (2, 8192), (2, 65536), (2, 1 << 30), (2, 1 << 15)], ids=["v3-8wave", "v3-globaldma", "v3-interleaved", "v3-nosplitk", "v6-halo-nosplitk", "v3-nohalo", "wgrad-v8", "wgrad-c64-pairs"])
def v3_engine(request):
    """Force the 8-wave conv kernels wherever they are supported (odtk_debug_set key 1 = 2); key 2 bit 8 selects 64-bit global addressing for the LDS-DMA,
    bit 14 the interleaved slab body, bit 13 turns split-K off, bit 30 lets the four-wave filter-gradient kernel (256 x 256 tiles) take short pixel ranges,
    bit 15 lets the block-pair halo filter gradient (wgrad3x3_c64k64_kernel on C, K multiples of 64) take small and ragged problems."""
    ops = _ops()
    ops.debug_set(1, request.param[0])
    ops.debug_set(2, request.param[1])
    yield
    ops.debug_set(1, 0)
    ops.debug_set(2, 0)


@pytest.mark.parametrize("case", CONV_CASES + V3_EXTRA_CASES)
def test_conv_v3_engine(case, dev, v3_engine):
    _conv_case(case, "bf16", dev)


@pytest.mark.parametrize("case", V3_EXTRA_CASES)
def test_conv_legacy_engine_extra(case, dev):
    ops = _ops()
    ops.debug_set(1, 1)
    try:
        _conv_case(case, "bf16", dev)
    finally:
        ops.debug_set(1, 0)


# Round 5: the small-map gather kernel (csrc/conv_v9.hip: 64 x 64 tiles, whole reduction per workgroup, parity phases for stride-2 input gradients) and the
# chunk-range split-K of the raster-run halo kernel.  odtk_debug_set key 6: bit 7 = the small-map kernel wherever it is supported (every case below, whatever
# its size), bit 6 = never (the same cases on the kernels it replaced: the 3 x 3 / stride 1 ones with few tiles then take the halo kernel's chunk split).
V9_EXTRA_CASES = [
    (1, 8, 8, 64, 64, 3, 2, 1),       # stride 2 on an even map (asymmetric SAME pad: the tap parities of the four phases swap)
    (3, 13, 17, 128, 64, 3, 2, 1),    # ... odd x odd, ragged phase grids (7 x 9, 7 x 8, 6 x 9, 6 x 8), tiles straddle images
    (2, 12, 12, 64, 128, 1, 2, 1),    # ... 1 x 1 / stride 2: three of the four phases have NO tap (zeros, still masked / accumulated)
    (2, 26, 26, 64, 128, 3, 2, 1),    # ... DarkNet-53 down-sampling geometry
    (1, 7, 9, 192, 96, 5, 2, 1),      # ... 5 x 5 taps: 9, 6, 6, 4 taps per phase
    (1, 1, 1, 64, 64, 3, 2, 1),       # ... a single pixel: three empty phases
    (16, 52, 52, 128, 256, 3, 2, 1),  # ... >= 160 tiles of 128 x 256: the phases on the 8-WAVE kernel, DarkNet-53 geometry
    (12, 104, 104, 64, 128, 3, 2, 1), # ... its 64-channel filter tile (dx has 64 channels)
    (50, 51, 37, 192, 192, 3, 2, 1),  # ... odd x odd map, ragged phase grids and tiles, two channel tiles with a tail (dx 192 channels, dy 192)
    (24, 52, 52, 256, 512, 1, 2, 1),  # ... 1 x 1 / stride 2 on the 8-wave kernel: three phases without a tap (their first DMA pieces still land before the image)
    (9, 45, 70, 8, 32, 3, 1, 1),      # first-layer kernel with 32 output channels (DarkNet-53's conv1: one MFMA row tile, 64-byte output rows), ragged tiles
    (4, 5, 5, 256, 128, 1, 1, 1),     # 1 x 1 on a 5 x 5 map: two tiles straddling four images
    (2, 3, 3, 256, 100, 3, 1, 1),     # 18 pixels: one ragged tile, channel tail 100 = 64 + 36
    (2, 10, 10, 72, 150, 3, 1, 1),    # C % 64 != 0: the per-lane tap walk (a slab straddles taps), channel tail
    (1, 9, 11, 24, 40, 3, 1, 1),      # ... 24 channels: 2.67 taps per slab, 216-element reduction = 3.4 slabs (zero-filled k tail)
    (2, 19, 19, 512, 512, 3, 1, 1),   # long reduction (72 slabs) through the four-stage ring
    (2, 19, 19, 1024, 150, 3, 1, 1),  # 144 slabs
    (3, 13, 13, 192, 256, 3, 1, 1),   # halo kernel's chunk split when the small-map kernel is off: three chunks over <= 3 parts
    (2, 19, 19, 104, 256, 3, 1, 1),   # ... with a PARTIAL last chunk (104 = 64 + 40 channels: the columns past C are zero-filled by the range check)
    (24, 19, 19, 104, 512, 3, 1, 1),  # ... the unsplit halo kernel on 104 channels (136 tiles), and its input gradient on 512
    (6, 38, 38, 152, 128, 3, 1, 1),   # ... 152 = 2 x 64 + 24
]


@pytest.fixture(params=[128, 64], ids=["v9-forced", "v9-off"])
def v9_engine(request):
    ops = _ops()
    ops.debug_set(6, request.param)
    yield
    ops.debug_set(6, 0)


@pytest.mark.parametrize("case", V9_EXTRA_CASES + V3_EXTRA_CASES[:12])
def test_conv_v9_engine(case, dev, v9_engine):
    _conv_case(case, "bf16", dev)
    assert _ops().conv_last_kernel() != ""


# Round 5 (late): 3 x 3 / stride 1 layers that would be split over channel chunks run on 128 x 128 tiles of the halo kernel (two workgroups per CU, whole reduction
# per workgroup) when those tiles give >= 2/3 of a workgroup per CU and the row is <= 47 pixels; odtk_debug_set(6, 16384) = off
HALO128_CASES = [
    (8, 26, 26, 256, 512, 3, 1, 1),     # DarkNet-53 at 8 images: 172 tiles, rows of 26 (the no-early-refill instantiation)
    (32, 19, 19, 1024, 150, 3, 1, 1),   # SSD300 pred2: 182 tiles, channel tail 150 = 128 + 22, 144 slabs
    (4, 40, 40, 128, 512, 3, 1, 1),     # rows of 40: the instantiation that refills patch groups 0 / 1 early
    (5, 33, 47, 192, 320, 3, 1, 1),     # rows of 47 = the largest patch (224 rows), ragged pixel and channel tiles, three chunks
    (8, 26, 26, 512, 256, 3, 1, 1),     # 64 x 128 tiles: the forward has 256 output channels (2 x 43 tiles of 128 x 128, 4 x 43 of 64 x 128); its input gradient 128 x 128
    (8, 13, 13, 512, 1024, 3, 1, 1),    # ... DarkNet-53's 13 x 13 forward (16 x 11 tiles of 64 x 128); the input gradient (8 x 11) stays on the chunk split
    (6, 40, 33, 128, 200, 3, 1, 1),     # ... rows of 33 (early patch refill), channel tail 200 = 3 x 64 + 8
    (8, 19, 19, 512, 512, 3, 1, 1),     # ... SSD300's conv5_x at batch 8 (what smoke() runs): 8 x 23 tiles of 64 x 128, 72 slabs
]


@pytest.mark.parametrize("off", [0, 16384], ids=["tiles128", "tiles128-off"])
@pytest.mark.parametrize("case", HALO128_CASES)
def test_conv_halo_kernel_on_128x128_tiles(case, off, dev):
    ops = _ops()
    ops.debug_set(6, off)
    try:
        _conv_case(case, "bf16", dev)
    finally:
        ops.debug_set(6, 0)


def test_conv_v9_is_taken_where_expected(dev):
    """the dispatch itself: SSD300's small layers at batch 32 and every stride-2 input gradient on whole 64-channel chunks run on the small-map kernel,
    the 13 x 13 / 3 x 3 layer of DarkNet-53 at 8 images on the halo kernel's chunk split, the trunk where it was"""
    ops = _ops()

    def kernels(N, H, C, K, k, s, d=1):
        Kp = ops.pad_to(K, 8)
        desc = ops.conv_desc(N, H, H, C, C, K, Kp, k, s, d, ops.BF16, ops.BF16)
        M = N * desc.Ho * desc.Wo
        x = torch.zeros(N * H * H, C, dtype=torch.bfloat16, device=dev)
        w = torch.zeros(K * k * k * C, dtype=torch.bfloat16, device=dev)
        wt = torch.zeros(C * k * k * Kp, dtype=torch.bfloat16, device=dev)
        y = torch.zeros(M, Kp, dtype=torch.bfloat16, device=dev)
        ops.conv2d_fwd(desc, x, w, torch.zeros(K, device=dev), y, True)
        f = ops.conv_last_kernel()
        ops.conv2d_dgrad(desc, y, Kp, wt, None, x, False)
        return f, ops.conv_last_kernel()
    assert kernels(32, 5, 128, 256, 3, 1) == ("conv_gather_v9_kernel", "conv_gather_v9_kernel")
    assert kernels(32, 10, 512, 128, 1, 1) == ("conv_gather_v9_kernel", "conv_gather_v9_kernel")
    assert kernels(32, 19, 256, 512, 3, 2)[1] == "conv_gather_v9_kernel"          # parity phases: the small-map kernel while the 8-wave kernel's tiles would cover less than ~60 % of the CUs (96 here)
    assert kernels(8, 208, 64, 128, 3, 2)[1] == "conv_gather_v3_kernel<64>"       # ... the 8-wave kernel beyond (1 352 tiles of 64 x 256: dx has 64 channels)
    assert kernels(8, 13, 512, 1024, 3, 1) == ("conv_gather_v6_kernel", "conv_gather_v6_kernel+splitk")     # forward: 176 tiles of 64 x 128, no split; dx has 512 channels: 88 -> chunk split
    assert kernels(8, 26, 256, 512, 3, 1) == ("conv_gather_v6_kernel", "conv_gather_v6_kernel")             # 172 tiles of 128 x 128 forward, of 64 x 128 for dx (256 channels)
    assert kernels(32, 38, 512, 512, 3, 1) == ("conv_gather_v6_kernel", "conv_gather_v6_kernel")
    torch.cuda.synchronize()


# every distinct convolution geometry of SSD300 (SSD300.py:192-314, heads :85-90) at batch 2, through the AUTO dispatch:
# first-layer kernel, 64->64 halo kernels, 8-wave gather, raster-run halo gather (also dilated), split-K, strided dgrad
SSD300_LAYER_CASES = [
    (2, 300, 300, 8, 64, 3, 1, 1), (2, 300, 300, 64, 64, 3, 1, 1), (2, 150, 150, 64, 128, 3, 1, 1),
    (2, 150, 150, 128, 128, 3, 1, 1), (2, 75, 75, 128, 256, 3, 1, 1), (2, 75, 75, 256, 256, 3, 1, 1),
    (2, 38, 38, 256, 512, 3, 1, 1), (2, 38, 38, 512, 512, 3, 1, 1), (2, 19, 19, 512, 512, 3, 1, 1),
    (2, 19, 19, 512, 1024, 3, 1, 2), (2, 19, 19, 1024, 1024, 1, 1, 1), (2, 19, 19, 1024, 256, 1, 1, 1),
    (2, 19, 19, 256, 512, 3, 2, 1), (2, 10, 10, 512, 128, 1, 1, 1), (2, 10, 10, 128, 256, 3, 2, 1),
    (2, 5, 5, 256, 128, 1, 1, 1), (2, 5, 5, 128, 256, 3, 1, 1), (2, 5, 5, 128, 256, 3, 2, 1),
    (2, 38, 38, 512, 100, 3, 1, 1), (2, 19, 19, 1024, 150, 3, 1, 1), (2, 10, 10, 512, 150, 3, 1, 1),
    (2, 5, 5, 256, 150, 3, 1, 1), (2, 5, 5, 256, 100, 3, 1, 1), (2, 3, 3, 256, 100, 3, 1, 1),
]


@pytest.mark.parametrize("case", SSD300_LAYER_CASES)
def test_conv_ssd300_layer_geometries_bf16(case, dev):
    _conv_case(case, "bf16", dev)


# ... and at the BENCHMARKED batch (BASELINE configs[1]: 32 images / GPU): the v6 grid of 1 408 tiles, persistent blocks that walk
# several tiles, the split-K decisions and the 32-bit byte offsets of the LDS-DMA all depend on N.  Reference: torch-CPU f32
# convolution fed the same bf16-rounded operands (6 TFLOP in all, ~20 s of host time).
@pytest.mark.parametrize("case", [(32,) + c[1:] for c in SSD300_LAYER_CASES])
def test_conv_ssd300_layer_geometries_bf16_batch32(case, dev):
    torch.set_num_threads(16)
    _conv_case(case, "bf16", dev)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_fwd_dgrad_wgrad(case, dt, dev):
    _conv_case(case, dt, dev)


@pytest.mark.parametrize("case", BACKBONE_CASES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_backbone_geometries(case, dt, dev):
    _conv_case(case, dt, dev)


def _conv_case(case, dt, dev):
    ops = _ops()
    N, H, W, C, K, k, stride, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    DT = ops.F32 if dt == "f32" else ops.BF16
    ch = ops.chunk(DT)
    tol = 2e-4 if dt == "f32" else 2e-2
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, k, k, C, generator=g) / math.sqrt(k * k * C)
    b = torch.randn(K, generator=g)
    if dt == "bf16":
        x = x.to(dtype).float()
    ldx = ops.pad_to(C, ch)
    Kp = ops.pad_to(K, 8)
    d = ops.conv_desc(N, H, W, ldx, ldx, K, Kp, k, stride, dil, DT, DT)
    Ho, Wo = d.Ho, d.Wo
    xd = to_rows(x, ldx, dtype, dev)
    wpad = torch.zeros(K, k, k, ldx)
    wpad[..., :C] = w
    wd_master = wpad.to(dev)
    w_c = torch.empty(K * k * k * ldx, dtype=dtype, device=dev)
    w_t = torch.empty(ldx * k * k * Kp, dtype=dtype, device=dev)
    ops.filter_prepare(wd_master, K, k, k, ldx, Kp, DT, w_c, w_t)
    wq = w_c.float().cpu().reshape(K, k, k, ldx)[..., :C]            # operand as the kernel sees it
    if dt == "f32":
        assert torch.equal(wq, w)
    # ---- forward (+bias+relu)
    yd = torch.zeros(N * Ho * Wo, Kp, dtype=dtype, device=dev)
    ops.conv2d_fwd(d, xd, w_c, b.to(dev), yd, True)
    torch.cuda.synchronize()
    y = from_rows(yd, N, Ho, Wo, K)
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    z_ref = _ref_conv(xr, wr, b, stride, dil)
    y_ref = F.relu(z_ref)
    scale = float(y_ref.detach().abs().max()) + 1e-6
    assert float((y - y_ref.detach()).abs().max()) <= tol * scale, "conv fwd mismatch"
    assert float(yd[:, K:].float().abs().max()) == 0.0 if Kp > K else True
    # ---- backward: random dy, no relu (pure linear ops)
    dy = torch.randn(N, Ho, Wo, K, generator=g)
    if dt == "bf16":
        dy = dy.to(dtype).float()
    z_ref.backward(dy)
    dyd = to_rows(dy, Kp, dtype, dev)
    dxd = torch.full((N * H * W, ldx), 7.0, dtype=dtype, device=dev)
    ops.conv2d_dgrad(d, dyd, Kp, w_t, None, dxd, False)
    dwd = torch.zeros(K, k, k, ldx, dtype=torch.float32, device=dev)
    dbd = torch.zeros(K, device=dev)
    ops.conv2d_wgrad(d, xd, dyd, Kp, dwd, dbd)
    torch.cuda.synchronize()
    db_ref = dy.reshape(-1, K).sum(0)
    assert float((dbd.cpu() - db_ref).abs().max()) <= (1e-4 if dt == "f32" else 1e-3) * (float(db_ref.abs().max()) + 1.0), "dbias mismatch"
    dx = from_rows(dxd, N, H, W, C)
    sx = float(xr.grad.abs().max()) + 1e-6
    assert float((dx - xr.grad).abs().max()) <= tol * sx, "dgrad mismatch"
    dw = dwd.cpu()[..., :C]
    sw = float(wr.grad.abs().max()) + 1e-6
    assert float((dw - wr.grad).abs().max()) <= tol * sw, "wgrad mismatch"
    # ---- dgrad with fused relu mask + accumulate
    src = torch.randn(N, H, W, C, generator=g)
    srcd = to_rows(src, ldx, dtype, dev)
    prev = torch.randn(N, H, W, C, generator=g)
    if dt == "bf16":
        prev = prev.to(dtype).float()
    dxd2 = to_rows(prev, ldx, dtype, dev)
    ops.conv2d_dgrad(d, dyd, Kp, w_t, srcd, dxd2, True)
    torch.cuda.synchronize()
    exp = (xr.grad + prev) * (srcd[:, :C].float().cpu().reshape(N, H, W, C) > 0)
    got = from_rows(dxd2, N, H, W, C)
    assert float((got - exp).abs().max()) <= (tol if dt == "f32" else 3e-2) * (float(exp.abs().max()) + 1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_filter_prepare_batched_equals_per_layer(dt, dev):
    """One launch for every layer (odtk_filter_prepare_batched) == the per-layer transform, bit for bit."""
    ops = _ops()
    DT = ops.F32 if dt == "f32" else ops.BF16
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 3, 8, 64), (100, 3, 256, 104), (150, 3, 40, 152), (256, 1, 1024, 256), (24, 3, 16, 24)]   # K, k, C, Kp
    entries, refs = [], []
    for (K, k, C, Kp) in shapes:
        w = torch.randn(K, k, k, C, generator=g).to(dev)
        wt = torch.full((C * k * k * Kp,), 7.0, dtype=dtype, device=dev)
        ref = torch.full((C * k * k * Kp,), 9.0, dtype=dtype, device=dev)
        ops.filter_prepare(w, K, k, k, C, Kp, DT, None, ref)
        entries.append((w.view(-1), wt, K, k, k, C, Kp)); refs.append(ref)
    ops.FilterPrepareBatch(entries, DT, dev).run()
    torch.cuda.synchronize()
    for e, r in zip(entries, refs):
        assert torch.equal(e[1], r)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 75, 75, 16, 2, 2), (2, 19, 19, 32, 3, 1), (1, 38, 38, 8, 2, 2)])
def test_maxpool(geom, dt, dev):
    ops = _ops()
    N, H, W, C, k, s = geom
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, C, generator=g).to(dtype).float()
    Ho, pt, _ = ops.same_pad(H, k, s)
    Wo, pl, _ = ops.same_pad(W, k, s)
    xd = to_rows(x, C, dtype, dev)
    yd = torch.empty(N * Ho * Wo, C, dtype=dtype, device=dev)
    ops.maxpool_fwd(xd, yd, N, H, W, C, C, Ho, Wo, k, s, pt, pl)
    xr = x.clone().requires_grad_(True)
    y_ref = R.maxpool_same(xr.permute(0, 3, 1, 2), k, s).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert torch.equal(from_rows(yd, N, Ho, Wo, C), y_ref.detach())
    dy = torch.randn(N, Ho, Wo, C, generator=g).to(dtype).float()
    y_ref.backward(dy)
    dxd = torch.empty_like(xd)
    ops.maxpool_bwd(xd, yd, to_rows(dy, C, dtype, dev), dxd, N, H, W, C, C, Ho, Wo, k, s, pt, pl)
    torch.cuda.synchronize()
    got = from_rows(dxd, N, H, W, C)
    # random data: no ties (bf16: ties possible -> compare only total mass and tie-free positions)
    if dt == "f32":
        assert float((got - xr.grad).abs().max()) < 1e-5
    else:
        assert abs(float(got.sum() - xr.grad.to(dtype).float().sum())) < 0.05 * float(xr.grad.abs().sum()) + 1.0


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 12, 12, 16), (2, 75, 75, 64), (3, 9, 13, 40), (4, 150, 150, 128)])
def test_maxpool2x2_recorded_argmax_equals_gather_path(geom, dt, dev):
    """The arg-max-index pooling pair must reproduce odtk_maxpool_fwd / _bwd bit for bit (incl. bf16 ties: both route
    to the FIRST maximum in window scan order) on even, odd (SAME pad-after) and non-square maps."""
    ops = _ops()
    N, H, W, C = geom
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, C, generator=g)
    if dt == "bf16":
        x = (x * 4).round() / 4                    # plenty of exact ties
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    xd = to_rows(x, C, dtype, dev)
    y0 = torch.empty(N * Ho * Wo, C, dtype=dtype, device=dev)
    y1 = torch.empty_like(y0)
    idx = torch.zeros(N * Ho * Wo * (C // ops.chunk(ops.F32 if dt == "f32" else ops.BF16)), dtype=torch.int16, device=dev)
    ops.maxpool_fwd(xd, y0, N, H, W, C, C, Ho, Wo, 2, 2, 0, 0)
    ops.maxpool2x2_fwd_idx(xd, y1, idx, N, H, W, C, C, Ho, Wo)
    dy = to_rows(torch.randn(N, Ho, Wo, C, generator=g), C, dtype, dev)
    dx0 = torch.full_like(xd, 3.0)
    dx1 = torch.full_like(xd, 5.0)
    ops.maxpool_bwd(xd, y0, dy, dx0, N, H, W, C, C, Ho, Wo, 2, 2, 0, 0)
    ops.maxpool2x2_bwd_idx(idx, dy, dx1, N, H, W, C, C, Ho, Wo)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(dx0, dx1)


@pytest.mark.parametrize("dt,ydt", [("f32", "f32"), ("bf16", "bf16"), ("bf16", "f32")])
@pytest.mark.parametrize("shape", [(2 * 19 * 19, 1024, True), (2 * 38 * 38, 100, False), (3 * 5 * 5, 150, False),
                                   (2 * 3 * 3, 256, True),
                                   (6 * 38 * 38, 100, False), (14 * 19 * 19, 256, True),    # M > 4096: split-row path
                                   # round 5: narrow bf16 maps (8 / 16 / 32 channels) pack 256 / 128 / 64 row lanes per block; many row splits of one column group
                                   (3 * 61 * 47, 16, True), (2 * 40 * 52, 32, True), (5000, 8, False), (70001, 64, True)])
@pytest.mark.parametrize("launches", [1, 10, 2, 3, 0], ids=["one-launch", "one-launch-64ch", "two-launches", "three-launches", "auto"])
def test_batchnorm(shape, dt, ydt, launches, dev):
    """Maps of <= 1024 rows take the single-launch kernels (statistics + finalize + apply: one workgroup per 16-byte channel chunk with 512 row lanes, or
    -- 'one-launch-64ch', the round-2 shape -- per 64 channels with 64 row lanes; here the limit is raised to 4096 rows to cover more shapes); larger maps
    three launches (statistics, finalize, apply) or two (apply with the finalize folded into its prologue over <= 32 row splits), picked by shape ('auto':
    two where 32 splits still give the statistics pass >= 128 workgroups); odtk_debug_set(4, ...) forces each."""
    ops = _ops()
    try:
        if launches in (1, 10):
            ops.debug_set(4, 4096)
            ops.debug_set(4, -3 if launches == 10 else -4)
        elif launches in (2, 3):
            ops.debug_set(4, 0)
            ops.debug_set(4, -5)
            ops.debug_set(4, -1 if launches == 3 else -2)
        _batchnorm_case(ops, shape, dt, ydt, dev)
    finally:
        ops.debug_set(4, 1024)
        ops.debug_set(4, -1)
        ops.debug_set(4, -4)
        ops.debug_set(4, -6)


def _batchnorm_case(ops, shape, dt, ydt, dev):
    M, C, relu = shape
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    ydtype = torch.float32 if ydt == "f32" else torch.bfloat16
    ch = 4 if dt == "f32" else 8
    ldz = ops.pad_to(C, ch)
    g = torch.Generator().manual_seed(2)
    z = (torch.randn(M, C, generator=g) * 3 + torch.randn(C, generator=g)).to(dtype).float()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    zd = torch.zeros(M, ldz, dtype=dtype, device=dev); zd[:, :C] = z.to(dtype).to(dev)
    mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
    sm = torch.empty(C, device=dev); si = torch.empty(C, device=dev)
    ws = torch.zeros(ops.bn_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    # head-style dense output: image-major with pitch C (rows_per_img = M/ nimg)
    nimg = 2 if M % 2 == 0 else 3
    rpi = M // nimg
    yd = torch.zeros(M, C, dtype=ydtype, device=dev)
    ops.bn_fwd(zd, M, C, ldz, gamma.to(dev), beta.to(dev), mm, mv, sm, si, True, relu, yd, C, rpi, rpi * C, ws)
    zr = z.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    mean = zr.mean(0); var = ((zr - mean) ** 2).mean(0)
    yr = (zr - mean) * torch.rsqrt(var + 1e-3) * gr + br
    if relu:
        yr = F.relu(yr)
    torch.cuda.synchronize()
    tol = 1e-4 if ydt == "f32" else 2e-2
    assert float((yd.float().cpu() - yr.detach()).abs().max()) <= tol * (float(yr.abs().max()) + 1e-6)
    assert float((sm.cpu() - mean.detach()).abs().max()) < 1e-4
    unb = var.detach() * M / (M - 1)
    assert float((mv.cpu() - (0.99 + 0.01 * unb)).abs().max()) < 1e-4
    assert float((mm.cpu() - 0.01 * mean.detach()).abs().max()) < 1e-4
    dy = torch.randn(M, C, generator=g).to(ydtype).float()
    # reference backward uses the mask from the GPU output so relu boundaries agree
    yr.backward(dy)
    dzd = torch.full((M, ldz), 5.0, dtype=dtype, device=dev)
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ops.bn_bwd(zd, yd, dy.to(ydtype).to(dev), M, C, ldz, C, rpi, rpi * C, gamma.to(dev), sm, si, relu, dzd, dg, db, ws)
    torch.cuda.synchronize()
    tolb = 5e-4 if (dt == "f32") else 3e-2
    sc = float(zr.grad.abs().max()) + 1e-6
    assert float((dzd[:, :C].float().cpu() - zr.grad).abs().max()) <= tolb * sc
    if ldz > C:
        assert float(dzd[:, C:].float().abs().max()) == 0.0
    # gamma / beta gradients (the reference masks with ITS ReLU boundary: rows within rounding of 0 may differ in bf16)
    tolg = 2e-3 if ydt == "f32" else 3e-2
    assert float((db.cpu() - br.grad).abs().max()) <= tolg * (float(br.grad.abs().max()) + 1e-6)
    assert float((dg.cpu() - gr.grad).abs().max()) <= tolg * (float(gr.grad.abs().max()) + 1e-6)
    # inference mode
    yd2 = torch.zeros(M, C, dtype=ydtype, device=dev)
    ops.bn_fwd(zd, M, C, ldz, gamma.to(dev), beta.to(dev), mm, mv, None, None, False, relu, yd2, C, rpi, rpi * C, ws)
    yi = (z - mm.cpu()) * torch.rsqrt(mv.cpu() + 1e-3) * gamma + beta
    if relu:
        yi = F.relu(yi)
    torch.cuda.synchronize()
    assert float((yd2.float().cpu() - yi).abs().max()) <= tol * (float(yi.abs().max()) + 1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_l2norm_colsum_sgd(dt, dev):
    ops = _ops()
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    M, C = 2 * 38 * 38, 512
    g = torch.Generator().manual_seed(3)
    x = F.relu(torch.randn(M, C, generator=g)).to(dtype).float()
    gamma = torch.tensor([20.0])
    xd = x.to(dtype).to(dev)
    yd = torch.empty_like(xd)
    ops.l2norm_fwd(xd, yd, M, C, C, gamma.to(dev))
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True)
    yr = xr * torch.rsqrt(torch.clamp((xr * xr).sum(1, keepdim=True), min=1e-12)) * gr
    torch.cuda.synchronize()
    tol = 1e-5 if dt == "f32" else 1e-2
    assert float((yd.float().cpu() - yr.detach()).abs().max()) <= tol * float(yr.abs().max())
    dy = torch.randn(M, C, generator=g).to(dtype).float()
    yr.backward(dy)
    prev = torch.randn(M, C, generator=g).to(dtype)
    dxd = prev.clone().to(dev)
    dgd = torch.zeros(1, device=dev)
    ops.l2norm_bwd(xd, dy.to(dtype).to(dev), dxd, M, C, C, gamma.to(dev), dgd, True, xd)
    torch.cuda.synchronize()
    exp = (xr.grad + prev.float()) * (x > 0)
    tolb = 1e-4 if dt == "f32" else 3e-2
    assert float((dxd.float().cpu() - exp).abs().max()) <= tolb * float(exp.abs().max())
    assert abs(float(dgd.cpu()) - float(gr.grad)) <= 2e-3 * abs(float(gr.grad)) + 1e-3
    # deterministic mode (odtk_debug_set key 5): the blocks' gamma sums in block order instead of float atomics -- twice the same bits, accumulated into dgamma
    ops.debug_set(5, 1)
    try:
        outs = []
        for _ in range(2):
            dg2 = torch.full((1,), 0.5, device=dev)
            dx2 = prev.clone().to(dev)
            ops.l2norm_bwd(xd, dy.to(dtype).to(dev), dx2, M, C, C, gamma.to(dev), dg2, True, xd)
            torch.cuda.synchronize()
            outs.append(float(dg2.cpu()))
            assert torch.equal(dx2, dxd)
    finally:
        ops.debug_set(5, 1)                                # (the library's default since round 6)
    assert outs[0] == outs[1] and abs(outs[0] - 0.5 - float(gr.grad)) <= 2e-3 * abs(float(gr.grad)) + 1e-3, outs
    # colsum
    ws = torch.zeros(ops.bn_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    out = torch.ones(C, device=dev)
    ops.colsum(dy.to(dtype).to(dev), M, C, C, out, True, ws)
    torch.cuda.synchronize()
    ref = dy.sum(0) + 1
    assert float((out.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    # sgd momentum + l2 partial + cast copy
    n = 100003
    p = torch.randn(n, generator=g); m = torch.randn(n, generator=g); gg = torch.randn(n, generator=g)
    pd, md, gd = p.to(dev), m.to(dev), gg.to(dev)
    part = torch.zeros(ops.sgd_blocks(n), device=dev)
    pc = torch.empty(n, dtype=dtype, device=dev)
    ops.sgd_momentum(pd, md, gd, 0.01, 0.9, 1e-4, 1.0, part, pc)
    tot = torch.zeros(1, device=dev)
    ops.sum_f32(part, tot)
    torch.cuda.synchronize()
    m_ref = 0.9 * m + (gg + 1e-4 * p)
    p_ref = p - 0.01 * m_ref
    assert float((md.cpu() - m_ref).abs().max()) < 1e-6
    assert float((pd.cpu() - p_ref).abs().max()) < 1e-6
    assert abs(float(tot.cpu()) - float((p * p).sum() / 2)) < 1e-3 * float((p * p).sum() / 2)
    assert float((pc.float().cpu() - p_ref.to(dtype).float()).abs().max()) < 1e-6


def test_preprocess(dev):
    ops = _ops()
    img = torch.rand(2, 30, 30, 3) * 255
    for dt, ld in ((torch.bfloat16, 8), (torch.float32, 4)):
        x = torch.full((2 * 30 * 30, ld), 9.0, dtype=dt, device=dev)
        ops.preprocess(img.to(dev), R.MEAN_RGB, ld, ops.dt_of(x), x)
        torch.cuda.synchronize()
        ref = (img - torch.tensor(R.MEAN_RGB)).reshape(-1, 3)
        assert torch.equal(x[:, :3].float().cpu(), ref.to(dt).float())
        assert float(x[:, 3:].float().abs().max()) == 0.0


# ---------------------------------------------------------------------------- box side
def _gpu_priors(ops, dev):
    from odtk.ssd300 import prior_spec
    fs, nas, hw = prior_spec()
    return ops.ssd_priors(300, fs, nas, hw, dev)


def test_priors_bit_exact(dev):
    ops = _ops()
    y1x1, y2x2, yx, hw, nb = _gpu_priors(ops, dev)
    ref = R.priors()
    torch.cuda.synchronize()
    assert y1x1.shape[0] == 8828
    for got, exp in zip((y1x1, y2x2, yx, hw), ref):
        assert torch.equal(got.cpu(), exp)
    exp_nb = torch.cat([ref[2] - ref[3] / 2., ref[2] + ref[3] / 2.], -1)
    assert torch.equal(nb.cpu(), exp_nb)


def _match_gpu(ops, dev, pri, gt):
    N, P, _ = gt.shape
    A = pri[0].shape[0]
    ngt = torch.empty(N, dtype=torch.int32, device=dev)
    best = torch.empty(N, P, dtype=torch.int32, device=dev)
    status = torch.empty(N, A, dtype=torch.uint8, device=dev)
    rg = torch.empty(N, A, dtype=torch.int32, device=dev)
    counts = torch.empty(N, 4, dtype=torch.int32, device=dev)
    ops.ssd_match(pri[0], pri[1], pri[3], gt.to(dev), ngt, best, status, rg, counts)
    return ngt, best, status, rg, counts


def test_match_bit_exact(dev):
    ops = _ops()
    pri = _gpu_priors(ops, dev)
    anchors = R.priors()
    _, gt = R.synthetic_batch(8, seed=5)
    # an image with two identical GT boxes (duplicate best anchors) and one with 59 objects
    gt[1, 1] = gt[1, 0]
    gt[1, 2:] = -1
    g = torch.Generator().manual_seed(9)
    n = 59
    h = torch.rand(n, generator=g) * 100 + 10; w = torch.rand(n, generator=g) * 100 + 10
    gt[2, :n] = torch.stack([h / 2 + torch.rand(n, generator=g) * (300 - h), w / 2 + torch.rand(n, generator=g) * (300 - w),
                             h, w, torch.randint(0, 20, (n,), generator=g).float()], 1)
    gt[2, n:] = -1
    ngt, best, status, rg, counts = _match_gpu(ops, dev, pri, gt)
    torch.cuda.synchronize()
    for i in range(gt.shape[0]):
        mt = R.match(anchors, gt[i])
        G = mt["G"]
        assert int(ngt[i]) == G
        assert torch.equal(best[i, :G].cpu().long(), mt["best"])
        st = torch.full((8828,), 0, dtype=torch.uint8)
        other = torch.nonzero(mt["othermask"]).squeeze(1)
        st[other] = torch.where(mt["pos"], torch.tensor(1, dtype=torch.uint8), torch.tensor(2, dtype=torch.uint8))
        assert torch.equal(status[i].cpu(), st)
        assert torch.equal(rg[i].cpu().long()[other], mt["rgindex"])
        num_pos = G + int(mt["pos"].sum()); num_neg = int((~mt["pos"]).sum())
        assert counts[i].cpu().tolist()[:3] == [num_pos, num_neg, min(3 * num_pos, num_neg)]


@pytest.mark.parametrize("n,thr,max_out", [(50, 0.5, 20), (700, 0.5, 20), (8828, 0.7, 513), (8828, 0.7, 8828),
                                           (3000, 0.3, 100), (1, 0.5, 5), (65, 0.0, 65)])
@pytest.mark.parametrize("engine", ["split", "single"])
def test_nms_bit_exact_vs_reference_kernel(n, thr, max_out, engine, dev):
    """split = sort -> suppression bit matrix -> scan (+ single-kernel fallback when the candidate
    margin is exhausted, exercised by max_out = 8828); single = one workgroup per problem."""
    ops = _ops()
    ops.debug_set(3, 1 if engine == "single" else 0)
    try:
        _nms_case(ops, n, thr, max_out, dev)
    finally:
        ops.debug_set(3, 0)


def _nms_case(ops, n, thr, max_out, dev):
    g = torch.Generator().manual_seed(n + max_out)
    B = 3
    idx_all, cnt_all = [], []
    yx = torch.rand(B, n, 2, generator=g) * 300
    hw = torch.rand(B, n, 2, generator=g) * 80 + 5
    boxes = torch.cat([yx - hw / 2, yx + hw / 2], -1).contiguous()
    boxes[0, : n // 3] = boxes[0, : n // 3][:, [2, 3, 0, 1]]          # flipped corners (coordinate-order agnostic)
    # distinct scores: equal scores are a documented tie-policy difference (heap order vs index order)
    scores = torch.stack([(torch.randperm(n, generator=g).float() + 1) / n for _ in range(B)])
    valid = (torch.rand(B, n, generator=g) > 0.2).to(torch.uint8) * 2
    out_idx = torch.full((B, max_out), -1, dtype=torch.int32, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    mo = torch.tensor([max_out, max(max_out // 2, 1), 0], dtype=torch.int32, device=dev)
    ops.nms_batched(boxes.to(dev), n * 4, scores.to(dev), n, 1, valid.to(dev), n, 1, 2, n, B, mo, 1, 0, thr,
                    out_idx, max_out, out_cnt)
    torch.cuda.synchronize()
    for b in range(B):
        m = valid[b] == 2
        ids = torch.nonzero(m).squeeze(1)
        ref = R.nms(boxes[b][m].numpy(), scores[b][m].numpy(), int(mo[b]), thr)
        ref = ids[torch.from_numpy(ref.astype(np.int64))].tolist()
        c = int(out_cnt[b])
        assert out_idx[b, :c].cpu().tolist() == ref


@pytest.mark.parametrize("mode", ["all_equal", "zeros_and_a_few", "two_values"])
def test_nms_split_path_with_thousands_of_tied_scores(mode, dev):
    """Hard-negative mining with a saturated background softmax: thousands of candidates share ONE score (exactly 0).  The top-k stage of the split path
    refines inside the tied histogram bin (radix select over the score AND index bits: the keys are unique), so the candidates it hands to the bit matrix
    are exactly the first ones in (score desc, index asc) order -- the keep list equals the single-workgroup kernel's and a plain greedy loop in that
    order, and (round 3) the problem no longer FALLS BACK to that kernel (1.2 ms per SSD300 step on the bench's fixed batch from the tenth step on)."""
    ops = _ops()
    n, B, thr = 8828, 4, 0.7
    g = torch.Generator().manual_seed(123)
    yx = torch.rand(n, 2, generator=g) * 300
    hw = torch.rand(n, 2, generator=g) * 80 + 5
    boxes = torch.cat([yx - hw / 2, yx + hw / 2], -1).contiguous()
    if mode == "all_equal":
        scores = torch.zeros(B, n)
    elif mode == "zeros_and_a_few":
        scores = torch.zeros(B, n)
        idx = torch.randint(0, n, (B, 300), generator=g)
        scores.scatter_(1, idx, torch.rand(B, 300, generator=g))
    else:
        scores = (torch.rand(B, n, generator=g) > 0.5).float() * 0.25
    max_out = torch.tensor([300, 900, 1500, 2500], dtype=torch.int32)
    cap = 4096

    def run(engine):
        ops.debug_set(3, 1 if engine == "single" else 0)
        try:
            out = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            cnt = torch.zeros(B, dtype=torch.int32, device=dev)
            ops.nms_batched(boxes.to(dev), 0, scores.to(dev), n, 1, None, 0, 0, 0, n, B, max_out.to(dev), 1, 0, thr, out, cap, cnt)
            torch.cuda.synchronize()
            return out.cpu(), cnt.cpu()
        finally:
            ops.debug_set(3, 0)
    o_split, c_split = run("split")
    o_single, c_single = run("single")
    assert torch.equal(c_split, c_single)
    for b in range(B):
        k = int(c_split[b])
        assert k > 0 and torch.equal(o_split[b, :k], o_single[b, :k]), (mode, b)
    # image 0 against a plain greedy loop in (score desc, index asc) order with the kernel's IoU
    order = sorted(range(n), key=lambda i: (-float(scores[0, i]), i))
    bx = boxes.double()
    area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
    keep = []
    for i in order:
        if len(keep) >= int(max_out[0]):
            break
        if keep:
            kk = torch.tensor(keep)
            ih = (torch.minimum(bx[i, 2], bx[kk, 2]) - torch.maximum(bx[i, 0], bx[kk, 0])).clamp(min=0)
            iw = (torch.minimum(bx[i, 3], bx[kk, 3]) - torch.maximum(bx[i, 1], bx[kk, 1])).clamp(min=0)
            inter = ih * iw
            if bool((inter / (area[i] + area[kk] - inter) > thr).any()):
                continue
        keep.append(i)
    assert o_split[0, :int(c_split[0])].tolist() == keep


def _loss_gpu(ops, dev, pri, pred, gt):
    N, A, ld = pred.shape
    predd = pred.to(dev).contiguous()
    gtd = gt.to(dev)
    ngt, best, status, rg, counts = _match_gpu(ops, dev, pri, gt)
    negloss = torch.empty(N, A, device=dev)
    ops.softmax_ce_const(predd, N * A, 21, ld, 20, negloss)
    sel = torch.full((N, A), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(N, dtype=torch.int32, device=dev)
    ops.nms_batched(pri[4], 0, negloss, A, 1, status, A, 1, 2, A, N, counts[:, 2:], 4, 0, 0.7, sel, A, cnt)
    parts = torch.empty(N, 4, device=dev)
    dpred = torch.empty_like(predd)
    ops.ssd_loss(predd, 21, pri[2], pri[3], gtd, ngt, best, status, rg, counts, negloss, sel, cnt, 1.0 / N, parts, dpred)
    torch.cuda.synchronize()
    return parts.cpu(), dpred.cpu(), sel.cpu(), cnt.cpu(), negloss.cpu()


def test_ssd_loss_and_grad_vs_oracle(dev):
    ops = _ops()
    pri = _gpu_priors(ops, dev)
    anchors = R.priors()
    N = 4
    _, gt = R.synthetic_batch(N, seed=11)
    g = torch.Generator().manual_seed(12)
    pred = torch.randn(N, 8828, 25, generator=g)
    parts, dpred, sel, cnt, negloss = _loss_gpu(ops, dev, pri, pred, gt)
    pr = pred.clone().requires_grad_(True)
    tot = 0
    for i in range(N):
        d = R.one_image_loss(pr[i, :, 21:23], pr[i, :, 23:], pr[i, :, :21], anchors, gt[i], detail=True)
        tot = tot + d["total"]
        for j, key in enumerate(["neg_loss", "pos_conf_loss", "coord", "total"]):
            assert abs(float(parts[i, j]) - float(d[key])) <= 2e-5 * max(1.0, abs(float(d[key]))), (i, key)
        # mined negatives: identical set unless two CE scores differ by < 1 ulp-ish between expf impls
        got = set(sel[i, : int(cnt[i])].tolist()); exp = set(d["sel_anchor"].tolist())
        assert len(got ^ exp) <= max(2, len(exp) // 100)
        # scores feeding the NMS agree to float rounding
        assert float((negloss[i][d["neg_anchor"]] - d["total_neg_loss"].detach()).abs().max()) < 5e-6
    (tot / N).backward()
    assert float((dpred - pr.grad).abs().max()) <= 1e-5 + 1e-3 * float(pr.grad.abs().max())


def test_mining_nms_bit_exact_given_same_scores(dev):
    """NMS keep-set bit-exact when both sides consume the SAME float scores."""
    ops = _ops()
    pri = _gpu_priors(ops, dev)
    anchors = R.priors()
    N = 3
    _, gt = R.synthetic_batch(N, seed=21)
    g = torch.Generator().manual_seed(22)
    pred = torch.randn(N, 8828, 25, generator=g)
    parts, dpred, sel, cnt, negloss = _loss_gpu(ops, dev, pri, pred, gt)
    nb = pri[4].cpu()
    for i in range(N):
        mt = R.match(anchors, gt[i])
        other = torch.nonzero(mt["othermask"]).squeeze(1)
        neg_idx = other[~mt["pos"]]
        num_pos = mt["G"] + int(mt["pos"].sum())
        k = min(3 * num_pos, len(neg_idx))
        ref = R.nms(nb[neg_idx].numpy(), negloss[i][neg_idx].numpy(), k, 0.7)
        assert sel[i, : int(cnt[i])].tolist() == neg_idx[torch.from_numpy(ref.astype(np.int64))].tolist()


def test_decode_and_detect_vs_oracle(dev):
    ops = _ops()
    pri = _gpu_priors(ops, dev)
    anchors = R.priors()
    g = torch.Generator().manual_seed(31)
    pred0 = torch.randn(8828, 25, generator=g)
    pred0[:, :21] *= 3
    pred0[:, 21:] *= 0.3
    A = 8828
    conf = torch.empty(A, 20, device=dev); boxes = torch.empty(A, 4, device=dev)
    keep = torch.empty(A, dtype=torch.uint8, device=dev); cand = torch.empty(A, 20, dtype=torch.uint8, device=dev)
    thr = 0.5
    ops.ssd_decode(pred0.to(dev), 21, pri[2], pri[3], thr, conf, boxes, keep, cand)
    out_idx = torch.full((20, 20), -1, dtype=torch.int32, device=dev)
    out_cnt = torch.zeros(20, dtype=torch.int32, device=dev)
    ops.nms_batched(boxes, 0, conf, 1, 20, cand, 1, 20, 1, A, 20, None, 0, 20, 0.5, out_idx, 20, out_cnt)
    torch.cuda.synchronize()
    rconf, rboxes, rkeep = R.decode(pred0, anchors)
    assert torch.equal(torch.nonzero(keep.cpu()).squeeze(1), rkeep)
    assert float((conf.cpu()[rkeep] - rconf).abs().max()) < 1e-5
    assert float((boxes.cpu()[rkeep] - rboxes).abs().max()) < 1e-3
    s_ref, b_ref, c_ref = R.detect(pred0, anchors, thr, 20, 0.5)
    sc, bb, cc = [], [], []
    for c in range(20):
        ids = out_idx[c, : int(out_cnt[c])].long()
        sc.append(conf[ids, c].cpu()); bb.append(boxes[ids].cpu()); cc += [c] * len(ids)
    sc = torch.cat(sc); bb = torch.cat(bb)
    assert cc == c_ref.tolist()
    assert float((sc - s_ref).abs().max()) < 1e-5
    assert float((bb - b_ref).abs().max()) < 1e-3


def test_wgrad_split_reduce_is_deterministic_and_matches_the_atomics(dev):
    """Filter gradients with > 1 pixel split under odtk_debug_set(5, 1): the blocks store partial tiles and wgrad_reduce_kernel
    adds them in split order (the default path uses float atomics: run-to-run differences in the last bits).  Two runs are
    bit-identical; the atomic path agrees to f32 round-off; dw is ACCUMULATED into (FCOS shares its head filters over 5 levels)."""
    ops = _ops()
    for (N, H, W, C, K, k, s, dil), v8 in [((6, 38, 38, 128, 256, 3, 1, 1), False), ((6, 20, 17, 256, 512, 3, 1, 1), True),
                                             ((8, 19, 19, 64, 100, 3, 1, 1), False),
                                             # round 5: the 64 -> 64 halo kernel (one partial per workgroup) and the first-layer kernel (one per wave)
                                             ((3, 41, 50, 64, 64, 3, 1, 1), False), ((3, 41, 50, 8, 64, 3, 1, 1), False),
                                             # round 6: the block-pair halo kernel (conv2_x; dbg bit 15 takes it on small problems): one partial per workgroup in its
                                             # block of a dW-shaped slot buffer
                                             ((3, 41, 50, 128, 128, 3, 1, 1), 1 << 15), ((2, 33, 64, 64, 128, 3, 1, 1), 1 << 15)]:
        Kp = ops.pad_to(K, 8)
        desc = ops.conv_desc(N, H, W, C, C, K, Kp, k, s, dil, ops.BF16, ops.BF16)
        M = N * desc.Ho * desc.Wo
        g = torch.Generator().manual_seed(11)
        x = torch.randn(N * H * W, C, generator=g).to(torch.bfloat16).to(dev)
        dy = torch.zeros(M, Kp, dtype=torch.bfloat16, device=dev)
        dy[:, :K] = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        outs = []
        for mode in (1, 1, 0):
            ops.debug_set(5, mode)
            ops.debug_set(2, (1 << 30) if v8 is True else int(v8))
            try:
                dw = torch.full((K, k, k, C), 0.5, device=dev)
                db = torch.full((K,), -2.0, device=dev)
                ops.conv2d_wgrad(desc, x, dy, Kp, dw, db)
                torch.cuda.synchronize()
                if v8 == 1 << 15:
                    assert ops.conv_last_kernel() == 'wgrad3x3_c64k64_kernel', ops.conv_last_kernel()
            finally:
                ops.debug_set(5, 1)                        # (the library's default since round 6)
                ops.debug_set(2, 0)
            outs.append((dw.clone(), db.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        sc = float((outs[2][0] - 0.5).abs().max())
        assert float((outs[0][0] - outs[2][0]).abs().max()) <= 2e-5 * sc + 1e-6
        assert float((outs[0][1] - outs[2][1]).abs().max()) <= 2e-5 * float((outs[2][1] + 2.0).abs().max()) + 1e-6
        # accumulate semantics: the 0.5 / -2 that were in the buffers are still there
        xr = x.float().view(N, H, W, C).permute(0, 3, 1, 2)
        wr = torch.zeros(K, C, k, k, device=dev, requires_grad=True)
        pt, pl = desc.pad_t, desc.pad_l
        Ho, Wo = desc.Ho, desc.Wo
        pb = max((Ho - 1) * s + (k - 1) * dil + 1 - H - pt, 0)
        pr = max((Wo - 1) * s + (k - 1) * dil + 1 - W - pl, 0)
        yr = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, stride=s, dilation=dil)
        yr.backward(dy[:, :K].float().view(N, Ho, Wo, K).permute(0, 3, 1, 2))
        ref = wr.grad.permute(0, 2, 3, 1) + 0.5
        assert float((outs[0][0] - ref).abs().max()) <= 2e-3 * float(wr.grad.abs().max())
        refb = dy[:, :K].float().sum(0) - 2.0
        assert float((outs[0][1] - refb).abs().max()) <= 2e-3 * float(refb.abs().max())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 40, 40, 20, 20, 64, True), (2, 40, 40, 5, 5, 512, True), (1, 10, 12, 3, 5, 16, True), (2, 8, 8, 16, 16, 32, True),
                                  (2, 9, 7, 4, 3, 24, False), (1, 6, 6, 1, 1, 8, True)])
def test_resize_bilinear_align_corners_any_scale(geom, dt, dev):
    """odtk_resize_bilinear2_* (PFPNetR.py:320-322: conv4_3 to 1/2, 1/4, 1/8 with align_corners=True): forward against the formula of
    tensorflow/core/kernels/resize_bilinear_op.cc (src = dst * (in - 1) / (out - 1), taps floor and min(floor + 1, in - 1)), backward
    against autograd of the same formula; accumulate and the ReLU mask of a bias + ReLU input."""
    from tests import mock_ops as MO
    ops = _ops()
    N, H, W, Ho, Wo, C, ac = geom
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N * H * W, C, generator=g).to(dtype)
    xd = x.to(dev)
    y = torch.full((N * Ho * Wo, C), 7.0, dtype=dtype, device=dev)
    ops.resize_bilinear2_fwd(xd, C, y, C, N, H, W, Ho, Wo, C, ac)
    ref = MO._bilinear2(x.float().reshape(N, H, W, C), Ho, Wo, ac).reshape(-1, C)
    tol = 1e-5 if dt == "f32" else 1e-2
    assert float((y.float().cpu() - ref).abs().max()) <= tol * (float(ref.abs().max()) + 1)
    dy = torch.randn(N * Ho * Wo, C, generator=g).to(dtype)
    xr = x.float().reshape(N, H, W, C).clone().requires_grad_(True)
    MO._bilinear2(xr, Ho, Wo, ac).backward(dy.float().reshape(N, Ho, Wo, C))
    gref = xr.grad.reshape(-1, C)
    dx = torch.full((N * H * W, C), 3.0, dtype=dtype, device=dev)
    ops.resize_bilinear2_bwd(dy.to(dev), C, dx, C, N, H, W, Ho, Wo, C, ac)
    assert float((dx.float().cpu() - gref).abs().max()) <= tol * (float(gref.abs().max()) + 1)
    # accumulate + ReLU mask (the mask applies to the NEW gradient only)
    base = torch.randn(N * H * W, C, generator=g).to(dtype)
    dx2 = base.clone().to(dev)
    ops.resize_bilinear2_bwd(dy.to(dev), C, dx2, C, N, H, W, Ho, Wo, C, ac, True, xd)
    want = base.float() + gref * (x.float() > 0)
    assert float((dx2.float().cpu() - want).abs().max()) <= 2 * tol * (float(want.abs().max()) + 1)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_copy_channels_unaligned_concat(dt, dev):
    """odtk_copy_channels: tf.concat of 512 + 85 + 85 + 85 channels (PFPNetR.py:366-396) and its gradient: element-granular offsets"""
    ops = _ops()
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(6)
    M = 333
    parts = [(512, 512), (85, 88), (85, 88), (85, 88)]
    srcs = [torch.randn(M, ld, generator=g).to(dtype).to(dev) for _, ld in parts]
    dst = torch.zeros(M, 768, dtype=dtype, device=dev)
    off = 0
    for (c, ld), s in zip(parts, srcs):
        ops.copy_channels(s, ld, 0, dst, 768, off, M, c)
        off += c
    want = torch.cat([s[:, :c] for (c, _), s in zip(parts, srcs)], 1)
    assert torch.equal(dst[:, :767], want) and float(dst[:, 767].float().abs().max()) == 0.0
    # gradient: slices back out, accumulate, ReLU mask of the destination's forward tensor
    dcat = torch.randn(M, 768, generator=g).to(dtype).to(dev)
    out = torch.full((M, 88), 2.0, dtype=dtype, device=dev)
    ops.copy_channels(dcat, 768, 597, out, 88, 0, M, 85)
    assert torch.equal(out[:, :85], dcat[:, 597:682]) and float((out[:, 85:].float() - 2.0).abs().max()) == 0.0
    fwd = torch.randn(M, 512, generator=g).to(dtype).to(dev)
    acc = torch.ones(M, 512, dtype=dtype, device=dev)
    ops.copy_channels(dcat, 768, 0, acc, 512, 0, M, 512, True, fwd)
    want = (1.0 + dcat[:, :512].float() * (fwd.float() > 0)).to(dtype)
    assert torch.equal(acc, want)


@pytest.mark.parametrize("case", [(12, 150, 150, 128, 64, 3, 1, 1), (48, 75, 75, 64, 64, 3, 1, 1), (13, 150, 148, 64, 40, 3, 1, 1)])
def test_conv_halo_kernel_64_channel_tiles(case, dev):
    """Cout <= 64 on the raster-run halo kernel: 64 x 512 tiles (conv2_1's input gradient, 128 -> 64 channels at W = 150, batch 32); needs
    >= 512 pixel tiles, so the shapes are large; the third has a channel tail (40 of 64 filter rows) and a ragged last tile"""
    _conv_case(case, "bf16", dev)


@pytest.mark.parametrize("geom", [(2, 16, 64, True), (3, 13, 70, True), (32, 300, 300, False), (5, 37, 45, True), (1, 8, 32, True)])
def test_conv_relu_pool2x2_fused_equals_conv_then_pool(geom, dev):
    """odtk_conv2d_fwd_pool2x2 (conv + bias + ReLU + tf.layers.max_pooling2d(2, 2, 'same'), SSD300.py:201-209, in the epilogue of the 64 -> 64 halo
    kernel) against the two launches it replaces: the pooled map and the recorded arg-max are BIT-identical (the same accumulators, the same bf16
    stores, first maximum in scan order), with and without the un-pooled output; odd sizes have windows that hang over the bottom / right edge.
    (32, 300, 300) is conv1_2 + pool1 of SSD300 at batch 32: 45 000 tiles, persistent multi-tile walks.)"""
    ops = _ops()
    N, H, W, check_cpu = geom
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = to_rows(torch.randn(N, H, W, 64, generator=g), 64, torch.bfloat16, dev)
    w = (torch.randn(64, 3, 3, 64, generator=g) * 0.06).to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    b = (torch.randn(64, generator=g) * 0.1).to(dev)
    d = ops.conv_desc(N, H, W, 64, 64, 64, 64, 3, 1, 1)
    assert ops.conv2d_fwd_pool2x2_fused(d)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    nchunk = N * Hp * Wp * 8
    y_ref = torch.zeros(N * H * W, 64, dtype=torch.bfloat16, device=dev)
    p_ref = torch.zeros(N * Hp * Wp, 64, dtype=torch.bfloat16, device=dev)
    i_ref = torch.zeros(nchunk, dtype=torch.int16, device=dev)
    ops.conv2d_fwd(d, x, w, b, y_ref, True)
    assert ops.conv_last_kernel() == 'conv3x3_c64k64_kernel'
    ops.maxpool2x2_fwd_idx(y_ref, p_ref, i_ref, N, H, W, 64, 64, Hp, Wp)
    for keep in (True, False):
        y = torch.full((N * H * W, 64), 7.0, dtype=torch.bfloat16, device=dev) if keep else None
        p = torch.full((N * Hp * Wp, 64), -3.0, dtype=torch.bfloat16, device=dev)
        i = torch.full((nchunk,), -1, dtype=torch.int16, device=dev)
        ops.conv2d_fwd_pool2x2(d, x, w, b, y, True, p, i)
        torch.cuda.synchronize()
        assert torch.equal(p, p_ref), f'pooled map differs (keep={keep})'
        assert torch.equal(i, i_ref), f'arg-max codes differ (keep={keep})'
        if keep:
            assert torch.equal(y, y_ref)
    if check_cpu:                                             # ... and the pair itself against plain torch
        xr = x.float().cpu().reshape(N, H, W, 64)
        wr = w.float().cpu().reshape(64, 3, 3, 64)
        conv = torch.relu(_ref_conv(xr, wr, b.cpu(), 1, 1))
        ref = F.max_pool2d(F.pad(conv.permute(0, 3, 1, 2), (0, 2 * Wp - W, 0, 2 * Hp - H), value=float('-inf')), 2, 2).permute(0, 2, 3, 1)
        got = from_rows(p_ref, N, Hp, Wp, 64)
        assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("geom", [(2, 150, 150, 64, 128), (2, 150, 150, 128, 128), (3, 75, 75, 128, 256), (1, 64, 64, 64, 200), (2, 13, 79, 64, 136),
                                  (1, 3, 158, 64, 128), (2, 6, 144, 192, 72), (32, 150, 150, 128, 128), (32, 75, 75, 256, 256)])
def test_conv_relu_pool2x2_fused_in_the_raster_run_halo_kernel(geom, dev):
    """odtk_conv2d_fwd_pool2x2 on the layers the round-4 variant of the raster-run halo kernel covers (C % 64 == 0, rows of 64..79 or 144..159 pixels: conv2_2 +
    pool2 and conv3_3 + pool3 of SSD300.py:219-252): tiles of whole row pairs, pooled in the epilogue.  Pooled map, recorded arg-max and (when kept) the
    un-pooled map are BIT-identical to conv + pool as two launches -- same slab order per output, same rounding, first maximum in scan order -- with windows
    that hang over the right / bottom edge (75 x 75 -> 38 x 38), a last tile of fewer rows (H = 75: 18 x 4 + 3; H = 3: 2 + 1), a ragged channel tile and two
    channel tiles; the last two are the SSD300 layers at batch 32."""
    ops = _ops()
    N, H, W, C_, K = geom
    g = torch.Generator().manual_seed(H * 1000 + W + K)
    Kp = ops.pad_to(K, 8)
    x = to_rows(torch.randn(N, H, W, C_, generator=g), C_, torch.bfloat16, dev)
    w = (torch.randn(K, 3, 3, C_, generator=g) * (0.6 / math.sqrt(9 * C_))).to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    b = (torch.randn(K, generator=g) * 0.1).to(dev)
    d = ops.conv_desc(N, H, W, C_, C_, K, Kp, 3, 1, 1)
    assert ops.conv2d_fwd_pool2x2_fused(d)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    nchunk = N * Hp * Wp * (K // 8)
    y_ref = torch.zeros(N * H * W, Kp, dtype=torch.bfloat16, device=dev)
    p_ref = torch.zeros(N * Hp * Wp, Kp, dtype=torch.bfloat16, device=dev)
    i_ref = torch.zeros(nchunk, dtype=torch.int16, device=dev)
    ops.conv2d_fwd(d, x, w, b, y_ref, True)
    # small problems: the stand-alone convolution sums split-K partials or runs on the small-map kernel (round 5) -- another order, last-bit differences
    split_k = 'splitk' in ops.conv_last_kernel() or 'v9' in ops.conv_last_kernel()
    res = {}
    for keep in (True, False):
        y = torch.full((N * H * W, Kp), 7.0, dtype=torch.bfloat16, device=dev) if keep else None
        p = torch.zeros(N * Hp * Wp, Kp, dtype=torch.bfloat16, device=dev)
        i = torch.full((nchunk,), -1, dtype=torch.int16, device=dev)
        ops.conv2d_fwd_pool2x2(d, x, w, b, y, True, p, i)
        assert ops.conv_last_kernel() == 'conv_gather_v6_kernel'
        torch.cuda.synchronize()
        res[keep] = (y, p, i)
    y, p, i = res[True]
    if split_k:
        assert float((y.float() - y_ref.float()).abs().max()) <= 2 ** -7 * float(y_ref.float().abs().max())
    else:
        assert torch.equal(y, y_ref)                              # same slab order per output element, same rounding: bit-identical
    ops.maxpool2x2_fwd_idx(y, p_ref, i_ref, N, H, W, K, Kp, Hp, Wp)          # the pool kernel on the fused launch's own un-pooled map
    torch.cuda.synchronize()
    if not torch.equal(p[:, :K], p_ref[:, :K]):
        bad = (p[:, :K] != p_ref[:, :K]).nonzero()
        raise AssertionError(f'pooled map differs: {bad.shape[0]} elements, first {bad[:6].tolist()}')
    assert torch.equal(i, i_ref), 'arg-max codes differ'
    assert torch.equal(res[False][1], p) and torch.equal(res[False][2], i), 'without the un-pooled output: other results'
    if N <= 3:                                                # ... and the pair itself against plain torch
        xr = x.float().cpu().reshape(N, H, W, C_)
        wr = w.float().cpu().reshape(K, 3, 3, C_)
        conv = torch.relu(_ref_conv(xr, wr, b.cpu(), 1, 1))
        ref = F.max_pool2d(F.pad(conv.permute(0, 3, 1, 2), (0, 2 * Wp - W, 0, 2 * Hp - H), value=float('-inf')), 2, 2).permute(0, 2, 3, 1)
        got = from_rows(p_ref, N, Hp, Wp, K)
        assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("geom", [(2, 40, 64), (3, 37, 45), (32, 300, 300)])
def test_relu_mask_as_sign_bits_first_layer_pair(geom, dev):
    """odtk_conv2d_fwd_bits / odtk_conv2d_dgrad_bits (round 4): conv1_1's forward kernel writes one byte of signs per 16-byte chunk beside its activation,
    conv1_2's input-gradient kernel masks with those bytes -- the activation and the masked gradient are BIT-identical to odtk_conv2d_fwd /
    odtk_conv2d_dgrad(relu_src = the activation), the bytes equal (y > 0) packed; ragged tiles, and the SSD300 layers at batch 32"""
    ops = _ops()
    N, H, W = geom
    g = torch.Generator().manual_seed(H * 100 + W)
    d1 = ops.conv_desc(N, H, W, 8, 8, 64, 64, 3, 1, 1)
    d2 = ops.conv_desc(N, H, W, 64, 64, 64, 64, 3, 1, 1)
    assert ops.conv2d_relu_bits_supported(d1, d2, 64)
    x = torch.zeros(N * H * W, 8, dtype=torch.bfloat16, device=dev)
    x[:, :3] = torch.randn(N * H * W, 3, generator=g).to(torch.bfloat16).to(dev)
    w1 = (torch.randn(64, 3, 3, 8, generator=g) * 0.2).to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    b1 = (torch.randn(64, generator=g) * 0.1).to(dev)
    y_ref = torch.zeros(N * H * W, 64, dtype=torch.bfloat16, device=dev)
    y = torch.full_like(y_ref, 3.0)
    bits = torch.full((N * H * W * 8,), 255, dtype=torch.uint8, device=dev)
    ops.conv2d_fwd(d1, x, w1, b1, y_ref, True)
    ops.conv2d_fwd_bits(d1, x, w1, b1, y, True, bits)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    want = ((y_ref.float() > 0).reshape(-1, 8).to(torch.int32) * (2 ** torch.arange(8, dtype=torch.int32, device=dev))).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert 0.2 < float((y_ref.float() > 0).float().mean()) < 0.8
    w2 = (torch.randn(64, 3, 3, 64, generator=g) * 0.05)
    wt = torch.empty(64 * 9 * 64, dtype=torch.bfloat16, device=dev)
    ops.filter_prepare(w2.to(dev), 64, 3, 3, 64, 64, ops.BF16, None, wt)
    dy = torch.randn(N * H * W, 64, generator=g).to(torch.bfloat16).to(dev)
    dx_ref = torch.zeros(N * H * W, 64, dtype=torch.bfloat16, device=dev)
    dx = torch.full_like(dx_ref, 5.0)
    ops.conv2d_dgrad(d2, dy, 64, wt, y_ref, dx_ref, False)
    assert ops.conv_last_kernel() == 'conv3x3_c64k64_kernel'
    ops.conv2d_dgrad_bits(d2, dy, 64, wt, bits, dx, False)
    assert ops.conv_last_kernel() == 'conv3x3_c64k64_kernel'
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)


def test_conv_pool2x2_unfused_shapes_run_as_two_launches(dev):
    """a shape the fused kernel does not cover (128 channels) goes through conv + pool inside the same entry point; without the un-pooled buffer it is
    refused loudly"""
    ops = _ops()
    from odtk._lib import OdtkError
    N, H, W, C_, K = 2, 20, 22, 128, 128
    g = torch.Generator().manual_seed(5)
    x = to_rows(torch.randn(N, H, W, C_, generator=g), C_, torch.bfloat16, dev)
    w = (torch.randn(K, 3, 3, C_, generator=g) * 0.05).to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    d = ops.conv_desc(N, H, W, C_, C_, K, K, 3, 1, 1)
    assert not ops.conv2d_fwd_pool2x2_fused(d)
    y = torch.zeros(N * H * W, K, dtype=torch.bfloat16, device=dev)
    y2 = torch.zeros_like(y)
    p = torch.zeros(N * (H // 2) * (W // 2), K, dtype=torch.bfloat16, device=dev)
    p2 = torch.zeros_like(p)
    i = torch.zeros(p.shape[0] * K // 8, dtype=torch.int16, device=dev)
    i2 = torch.zeros_like(i)
    ops.conv2d_fwd_pool2x2(d, x, w, None, y, True, p, i)
    ops.conv2d_fwd(d, x, w, None, y2, True)
    ops.maxpool2x2_fwd_idx(y2, p2, i2, N, H, W, K, K, H // 2, W // 2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(p, p2) and torch.equal(i, i2)
    with pytest.raises(OdtkError):
        ops.conv2d_fwd_pool2x2(d, x, w, None, None, True, p, i)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 19, 19, 64, 3, 1), (32, 19, 19, 512, 3, 1), (3, 9, 13, 40, 3, 2), (2, 12, 12, 16, 2, 2), (2, 7, 7, 8, 2, 1)])
def test_maxpool_recorded_argmax_overlapping_windows(geom, dt, dev):
    """odtk_maxpool_fwd_argmax / _bwd_argmax (pool5: 3x3 / stride 1 / SAME) against the gather path it replaces: identical outputs and identical routed
    gradients (first maximum in scan order), ties included -- the inputs are quantised so that many windows hold repeated maxima"""
    ops = _ops()
    N, H, W, C_, k, s = geom
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(H + 7 * C_)
    x = to_rows((torch.randn(N, H, W, C_, generator=g) * 2).round() / 2, C_, tdt, dev)          # half-integer values: ties
    Ho, pt, _ = ops.same_pad(H, k, s)
    Wo, pl, _ = ops.same_pad(W, k, s)
    dy = to_rows(torch.randn(N, Ho, Wo, C_, generator=g), C_, tdt, dev)
    kc = 4 if dt == "f32" else 8
    y0, y1 = (torch.zeros(N * Ho * Wo, C_, dtype=tdt, device=dev) for _ in range(2))
    dx0, dx1 = (torch.full((N * H * W, C_), 3.0, dtype=tdt, device=dev) for _ in range(2))
    arg = torch.zeros(N * Ho * Wo * (C_ // kc), dtype=torch.int32, device=dev)
    ops.maxpool_fwd(x, y0, N, H, W, C_, C_, Ho, Wo, k, s, pt, pl)
    ops.maxpool_bwd(x, y0, dy, dx0, N, H, W, C_, C_, Ho, Wo, k, s, pt, pl)
    ops.maxpool_fwd_argmax(x, y1, arg, N, H, W, C_, C_, Ho, Wo, k, s, pt, pl)
    ops.maxpool_bwd_argmax(arg, dy, dx1, N, H, W, C_, C_, Ho, Wo, k, s, pt, pl)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    if dt == "f32":
        assert torch.equal(dx0, dx1)
    else:                                   # both sum <= 9 bf16 values in f32 and round once; the order of the windows is the same
        assert float((dx0.float() - dx1.float()).abs().max()) <= 2 ** -7 * float(dx0.float().abs().max())
    # every output's gradient lands exactly once
    assert abs(float(dx1.float().sum()) - float(dy.float().sum())) <= 1e-2 * float(dy.float().abs().sum()) ** 0.5 + 1e-3 * float(dy.float().abs().sum())


@pytest.mark.parametrize("geom", [(2, 20, 20), (3, 37, 45), (30, 45, 70), (32, 300, 300), (1, 2, 33), (5, 7, 300)])
def test_first_layer_filter_gradient_kernel(geom, dev):
    """wgrad3x3_c8k64_kernel (conv1_1's Conv2DBackpropFilter, SSD300.py:193-200: 8 = 3 + 5 zero input channels -> 64; every wave walks its own strips of
    2 rows x 32 columns through a private three-stage LDS ring) against plain torch: dW [64][3][3][8] and the fused bias gradient, ragged right / bottom
    edges, odd heights (a strip with one row), fewer strips than waves, and conv1_1 itself at batch 32 (48 000 strips, 47 per wave)."""
    ops = _ops()
    N, H, W = geom
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(N, H, W, 8, generator=g).to(torch.bfloat16).float()
    x[..., 3:] = 0                                              # the layout of conv1_1's input: three real channels in one chunk
    dy = torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16).float()
    d = ops.conv_desc(N, H, W, 8, 8, 64, 64, 3, 1, 1)
    xd, dyd = to_rows(x, 8, torch.bfloat16, dev), to_rows(dy, 64, torch.bfloat16, dev)
    dw = torch.zeros(64, 3, 3, 8, device=dev)
    db = torch.zeros(64, device=dev)
    dw[0, 0, 0, 0] = 5.0                                        # the kernel ACCUMULATES into dw / dbias
    db[3] = -2.0
    ops.conv2d_wgrad(d, xd, dyd, 64, dw, db)
    assert ops.conv_last_kernel() == 'wgrad3x3_c8k64_kernel'
    torch.cuda.synchronize()
    wr = torch.zeros(64, 3, 3, 8, requires_grad=True)
    _ref_conv(x, wr, None, 1, 1).backward(dy)
    ref = wr.grad.clone()
    ref[0, 0, 0, 0] += 5.0
    scale = float(ref.abs().max()) + 1e-6
    assert float((dw.cpu() - ref).abs().max()) <= 2e-3 * scale, (float((dw.cpu() - ref).abs().max()), scale)
    assert float(dw.cpu()[..., 3:].abs().max()) == 0.0          # zero input channels: exactly zero gradient
    db_ref = dy.reshape(-1, 64).sum(0)
    db_ref[3] -= 2.0
    assert float((db.cpu() - db_ref).abs().max()) <= 2e-3 * (float(db_ref.abs().max()) + 1.0)


@pytest.mark.parametrize("n", [1, 32, 1000, 70001])
def test_loss_total_and_zero(n, dev):
    """odtk_loss_total: the scalar a step reports (strided per-image column + the optimizer's partials) in one launch, against float64; odtk_zero clears
    any byte range (unaligned head / tail included)"""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    parts = torch.randn(n, 4, generator=g).to(dev)
    l2 = torch.rand(3 * n + 5, generator=g).to(dev)
    sa, sb, tot = (torch.full((1,), 9.0, device=dev) for _ in range(3))
    ops.loss_total(parts[:, 3], n, 4, l2, 1.0 / 32, 1e-4, sa, sb, tot)
    ra, rb = parts[:, 3].double().sum().item(), l2.double().sum().item()
    assert abs(sa.item() - ra) <= 1e-5 * max(1.0, parts[:, 3].abs().double().sum().item())
    assert abs(sb.item() - rb) <= 1e-5 * rb
    assert abs(tot.item() - (ra / 32 + 1e-4 * rb)) <= 1e-5 * (abs(ra) / 32 + 1e-4 * rb + 1e-6)
    tot2 = torch.zeros(1, device=dev)
    ops.loss_total(parts[:, 3], n, 4, l2, 1.0 / 32, 1e-4, None, None, tot2)
    assert tot2.item() == tot.item()
    buf = torch.full((n + 7,), 3.0, device=dev)
    ops.zero(buf[3:3 + n])
    torch.cuda.synchronize()
    assert float(buf[3:3 + n].abs().sum()) == 0.0 and bool((buf[:3] == 3.0).all()) and bool((buf[3 + n:] == 3.0).all())


# ---- "x3": f32 convolutions on the bf16 MFMA kernels by operand splitting (include/odtk.h).  Geometries of RetinaNet.py:594-643's units (1x1 / 3x3, stride 1 | 2,
# channel counts that are multiples of 8), the heads' 189 / 36 channels, split-K and plain tiles.  Reference: the convolution in f64.  Bound: 3e-5 of the
# output scale (measured ~4e-6; one bf16 product would be 4e-3) -- the f32 kernels' own bound is 2e-4.
X3_CASES = [
    (2, 25, 25, 256, 64, 1, 1, 1),     # bottleneck 1x1
    (2, 25, 25, 64, 64, 3, 1, 1),      # 3x3
    (2, 26, 26, 128, 256, 3, 2, 1),    # stride-2 3x3 (asymmetric SAME pad)
    (2, 26, 26, 256, 512, 1, 2, 1),    # stride-2 1x1 shortcut
    (2, 13, 13, 256, 189, 3, 1, 1),    # class head: 189 = 9 x 21 channels (pitch 192)
    (1, 7, 7, 256, 36, 3, 1, 1),       # box head on a small level: split-K
    (3, 50, 50, 64, 256, 1, 1, 1),     # many pixel tiles
    (1, 4, 4, 2048, 256, 1, 1, 1),     # deep reduction, 16 pixels
    (2, 50, 50, 28, 56, 3, 1, 1),      # 28 channels (RetinaNet.py:27's widths): not a multiple of 8 -- forward / input gradient split, filter gradient exact
    (2, 30, 30, 100, 64, 3, 2, 1),     # 100 channels, stride 2
    # the raster-run halo kernel with f32 output (3x3 / stride 1 over whole 64-channel chunks, enough tiles for one part per tile), one case per variant
    (8, 64, 64, 64, 256, 3, 1, 1),     # 256-pixel tiles, double-buffered patch
    (8, 50, 50, 128, 256, 3, 1, 1),    # 192-pixel tiles (P4 of the 800-pixel pyramid); the input gradient (128 channels out of 768 split ones) too
    (2, 100, 100, 128, 256, 3, 1, 1),  # rows of 96-111 pixels (P3): single patch buffer with early refill; input gradient too
    (1, 130, 130, 64, 189, 3, 1, 1),   # rows of 128-143 pixels, a channel tail
    (32, 64, 64, 64, 160, 3, 1, 1),    # 512-pixel tiles
    (1, 150, 150, 64, 256, 3, 1, 1),   # rows of 144-159 pixels
    (1, 160, 160, 64, 128, 3, 1, 1),   # rows of 160-175 pixels (conv2_x of the 320-pixel models)
    (2, 120, 120, 64, 130, 3, 1, 1),   # rows of 112-127 pixels
    (4, 80, 80, 64, 256, 3, 1, 1),     # rows of 80-95 pixels (conv3_x of the 320-pixel models): 448-row single patch buffer
    (3, 95, 95, 64, 200, 3, 1, 1),     # ... its longest row: the patch is exactly full
    # round 6: few-tile 3x3 layers on the 128 x 128 / 64 x 128 halo tiles with f32 output instead of split-K + finish (DarkNet-53 at 8 images: YOLOv3's default engine)
    (8, 26, 26, 256, 512, 3, 1, 1),    # 26 x 26: 43 x 4 tiles of 128 x 128 (W < 32: no early patch refill); its input gradient: 256 channels out -> 64 x 128 tiles
    (8, 13, 13, 512, 1024, 3, 1, 1),   # 13 x 13: 11 x 8 tiles of 128 x 128 < 2/3 of the CUs -> 64 x 128 tiles (11 x 16)
    (8, 40, 40, 128, 256, 3, 1, 1),    # rows of 32-47 pixels: the early-refill instantiation, 100 x 2 tiles
    (5, 20, 37, 64, 300, 3, 1, 1),     # ragged: channel tail 300 = 2 x 128 + 44, tiles straddling images, one 64-channel chunk (x 3 parts)
    # round 6: the small-map kernel (64 x 64 tiles, whole reduction per workgroup) with f32 output on the split operands: DarkNet-53's 1 x 1 layers at 8 images
    (8, 13, 13, 1024, 512, 1, 1, 1),   # 22 x 8 tiles, 48 slabs (3 x 1 024 channels); input gradient 16 x 22 tiles over 3 x 512
    (8, 26, 26, 512, 256, 1, 1, 1),    # 85 x 4 tiles: the four-stage ring (more tiles than CUs)
    (8, 52, 52, 256, 128, 1, 1, 1),    # 338 x 2 tiles
    # round 6: stride-2 input gradients as four parity phases on the 8-wave kernel with f32 output (>= 128 tiles; DarkNet-53's down-sampling layers)
    (8, 104, 104, 64, 128, 3, 2, 1),   # dx has 64 channels: the 64-channel filter tile, 4 x 85 tiles
    (16, 52, 52, 128, 256, 3, 2, 1),   # 128-channel tile
    (6, 51, 37, 192, 192, 3, 2, 1),    # odd x odd map: ragged phase grids and tiles, two channel tiles with a tail
    (24, 52, 52, 256, 512, 1, 2, 1),   # 1 x 1 / stride 2: three of the four phases have no tap (zeros, still masked / accumulated)
    (3, 9, 11, 64, 100, 3, 1, 1),      # 3 x 3 on a map too small for the halo tiles: nine taps x 3 chunks through the tap walk (27 slabs), channel tail 100, ragged last tile
]


@pytest.mark.parametrize("case", X3_CASES)
def test_conv_x3_operand_splitting_against_f64(case, dev):
    ops = _ops()
    N, H, W, C, K, k, stride, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, H, W, C, generator=g) * torch.exp(torch.randn(N, H, W, 1, generator=g))          # a spread of magnitudes: the low halves matter
    w = torch.randn(K, k, k, C, generator=g) / math.sqrt(k * k * C)
    b = torch.randn(K, generator=g)
    ldy = ops.pad_to(K, 4)
    d = ops.conv_desc(N, H, W, C, C, K, ldy, k, stride, dil, ops.F32X3, ops.F32X3)
    df = ops.conv_desc(N, H, W, C, C, K, ldy, k, stride, dil, ops.F32, ops.F32)
    ops.debug_set(6, 8)                                   # every covered geometry through the split path, also below the policy's size threshold
    try:
        assert ops.conv2d_x3_supported(d) == (7 if C % 8 == 0 else 5) and ops.conv2d_x3_supported(df) == 0
        Ho, Wo = d.Ho, d.Wo
        wd = w.to(dev).contiguous()
        w_c = torch.empty(K * k * k * C, device=dev)
        w_t = torch.empty(C * k * k * ldy, device=dev)
        ops.filter_prepare(wd, K, k, k, C, ldy, ops.F32, w_c, w_t)
        xd = to_rows(x, C, torch.float32, dev)
        yd = torch.full((N * Ho * Wo, ldy), 7.0, device=dev)
        ops.conv2d_fwd(d, xd, w_c, b.to(dev), yd, True)
        assert 'x3' in ops.conv_last_kernel()
        torch.cuda.synchronize()
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        z_ref = _ref_conv(xr, wr, b.double(), stride, dil)
        y_ref = F.relu(z_ref).detach()
        y = from_rows(yd, N, Ho, Wo, K).double()
        tol = 3e-5
        assert float((y - y_ref).abs().max()) <= tol * (float(y_ref.abs().max()) + 1e-6), "x3 forward"
        if ldy > K:
            assert float(yd[:, K:].abs().max()) == 0.0
        # the exact f32 kernel on the same operands: the two engines agree to the same bound
        yf = torch.zeros_like(yd)
        ops.conv2d_fwd(df, xd, w_c, b.to(dev), yf, True)
        assert 'x3' not in ops.conv_last_kernel()
        assert float((yf - yd).abs().max()) <= tol * (float(y_ref.abs().max()) + 1e-6)
        # backward
        dy = torch.randn(N, Ho, Wo, K, generator=g) * torch.exp(torch.randn(N, Ho, Wo, 1, generator=g))
        z_ref.backward(dy.double())
        dyd = to_rows(dy, ldy, torch.float32, dev)
        dxd = torch.full((N * H * W, C), 7.0, device=dev)
        ops.conv2d_dgrad(d, dyd, ldy, w_t, None, dxd, False)
        assert 'x3' in ops.conv_last_kernel()
        dwd = torch.zeros(K, k, k, C, device=dev)
        dbd = torch.zeros(K, device=dev)
        ops.conv2d_wgrad(d, xd, dyd, ldy, dwd, dbd)
        torch.cuda.synchronize()
        dx = from_rows(dxd, N, H, W, C).double()
        sx = float(xr.grad.abs().max()) + 1e-6
        assert float((dx - xr.grad).abs().max()) <= tol * sx, "x3 input gradient"
        sw = float(wr.grad.abs().max()) + 1e-6
        assert float((dwd.cpu().double() - wr.grad).abs().max()) <= tol * sw, "x3 filter gradient"
        db_ref = dy.double().reshape(-1, K).sum(0)
        assert float((dbd.cpu().double() - db_ref).abs().max()) <= 1e-5 * float(dy.double().reshape(-1, K).abs().sum(0).max()), "bias gradient"
        # accumulation into dw / dbias (the caller's zeroed gradient buffers): a second call doubles them
        ops.conv2d_wgrad(d, xd, dyd, ldy, dwd, dbd)
        torch.cuda.synchronize()
        assert float((dwd.cpu().double() - 2 * wr.grad).abs().max()) <= 2 * tol * sw
        assert float((dbd.cpu().double() - 2 * db_ref).abs().max()) <= 2e-5 * float(dy.double().reshape(-1, K).abs().sum(0).max())
        # the input gradient through a producer's ReLU mask, accumulated into what dx holds ((prev + new) where src > 0, else 0)
        src = torch.randn(N, H, W, C, generator=g)
        prev = torch.randn(N, H, W, C, generator=g)
        srcd = to_rows(src, C, torch.float32, dev)
        dx2 = to_rows(prev, C, torch.float32, dev)
        ops.conv2d_dgrad(d, dyd, ldy, w_t, srcd, dx2, True)
        torch.cuda.synchronize()
        exp = (xr.grad + prev.double()) * (src > 0)
        assert float((from_rows(dx2, N, H, W, C).double() - exp).abs().max()) <= tol * (float(exp.abs().max()) + 1e-6), "masked, accumulated input gradient"
    finally:
        ops.debug_set(6, 0)
