"""Known answers of TensorFlow 1.x's OWN kernels and documentation, quoted from memory of TensorFlow's source tree (no
TensorFlow binary exists in this environment; every vector names the file / test it comes from so that it can be checked
against a TensorFlow checkout).  They pin the layer BELOW the reference's Python: the semantics oracle/tf_shim, the oracles
and the HIP kernels all have to share.  Used by tests/test_tf_known_answers_cpu.py and tests/test_gpu_tf_known_answers.py.
"""
import numpy as np

# ---- tensorflow/core/kernels/non_max_suppression_op_test.cc (NonMaxSuppressionOpTest / V2 / V3: same vectors) -------------
# boxes are (y1, x1, y2, x2); three clusters of overlapping boxes
NMS_BOXES = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], np.float32)
NMS_BOXES_FLIPPED = np.array([[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, .9, 1, -0.1], [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]], np.float32)
NMS_SCORES = np.array([.9, .75, .6, .95, .5, .3], np.float32)
NMS_CASES = [
    # name of the TEST_F,                                   boxes,             scores,            max, iou, score_thr, expected
    ('TestSelectFromThreeClusters',                          NMS_BOXES,         NMS_SCORES,        3,   .5,  None,     [3, 0, 5]),
    ('TestSelectFromThreeClustersFlippedCoordinates',        NMS_BOXES_FLIPPED, NMS_SCORES,        3,   .5,  None,     [3, 0, 5]),
    ('TestSelectAtMostTwoBoxesFromThreeClusters',            NMS_BOXES,         NMS_SCORES,        2,   .5,  None,     [3, 0]),
    ('TestSelectWithNegativeScores',                         NMS_BOXES,         NMS_SCORES - 10.,  6,   .5,  None,     [3, 0, 5]),
    ('TestSelectAtMostThirtyBoxesFromThreeClusters',         NMS_BOXES,         NMS_SCORES,        30,  .5,  None,     [3, 0, 5]),
    ('TestSelectSingleBox',                                  NMS_BOXES[:1],     NMS_SCORES[:1],    3,   .5,  None,     [0]),
    ('TestSelectFromTenIdenticalBoxes',                      np.tile(NMS_BOXES[:1], (10, 1)), np.full(10, .9, np.float32), 3, .5, None, [0]),
    ('V3 TestSelectFromThreeClustersWithScoreThreshold',     NMS_BOXES,         NMS_SCORES,        3,   .5,  .4,       [3, 0]),
    ('TestEmptyInput',                                       np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 30, .5, None, []),
]

# ---- tensorflow/python/ops/image_ops_test.py, ResizeImagesTest -------------------------------------------------------------
# testResizeUp (align_corners=False, the TF-1.x grid: src = dst * in / out): 3 x 2 -> 6 x 4
RESIZE_LEGACY_IN = np.array([64, 32, 32, 64, 50, 100], np.float32).reshape(3, 2, 1)
RESIZE_LEGACY_OUT = np.array([64.0, 48.0, 32.0, 32.0, 48.0, 48.0, 48.0, 48.0, 32.0, 48.0, 64.0, 64.0,
                              41.0, 61.5, 82.0, 82.0, 50.0, 75.0, 100.0, 100.0, 50.0, 75.0, 100.0, 100.0], np.float32).reshape(6, 4, 1)
# testResizeUpAlignCornersTrue: 3 x 2 -> 5 x 4, scale = (in - 1) / (out - 1)
RESIZE_ALIGN_IN = np.array([6, 3, 3, 6, 6, 9], np.float32).reshape(3, 2, 1)
RESIZE_ALIGN_OUT = np.array([6.0, 5.0, 4.0, 3.0, 4.5, 4.5, 4.5, 4.5, 3.0, 4.0, 5.0, 6.0, 4.5, 5.5, 6.5, 7.5, 6.0, 7.0, 8.0, 9.0], np.float32).reshape(5, 4, 1)

# testResizeUpAlignCornersTrue, expected_data[ResizeMethod.NEAREST_NEIGHBOR] on the same 3 x 2 input: source = round(dst * scale)
RESIZE_ALIGN_NEAREST_OUT = np.array([6.0, 6.0, 3.0, 3.0, 3.0, 3.0, 6.0, 6.0, 3.0, 3.0, 6.0, 6.0, 6.0, 6.0, 9.0, 9.0, 6.0, 6.0, 9.0, 9.0],
                                    np.float32).reshape(5, 4, 1)
# testResizeUpBicubic: 6 x 6 u8 -> 8 x 8, align_corners=False, compared with atol=1 there (the table holds rounded integers)
RESIZE_BICUBIC_IN = np.array([128, 128, 64, 64, 128, 128, 64, 64, 64, 64, 128, 128, 64, 64, 128, 128,
                              50, 50, 100, 100, 50, 50, 100, 100, 50, 50, 100, 100, 50, 50, 100, 100, 50, 50, 100, 100], np.float32).reshape(6, 6, 1)
RESIZE_BICUBIC_OUT = np.array([128, 135, 96, 55, 64, 114, 134, 128, 78, 81, 68, 52, 57, 118, 144, 136, 55, 49, 79, 109, 103, 89, 83, 84,
                               74, 70, 95, 122, 115, 69, 49, 55, 100, 105, 75, 43, 50, 89, 105, 100, 57, 54, 74, 96, 91, 65, 55, 58,
                               70, 69, 75, 81, 80, 72, 69, 70, 105, 112, 75, 36, 45, 92, 111, 105], np.float32).reshape(8, 8, 1)

# ---- tensorflow/python/ops/image_ops_test.py, AdjustHueTest / AdjustContrastTest -------------------------------------------------
# testAdjustNegativeHue / testAdjustPositiveHue: 2 x 2 RGB u8 picture, delta -0.25 / +0.25.  TensorFlow converts u8 -> float (x / 255),
# shifts the hue, and converts back with convert_image_dtype(saturate=True): trunc(y * 255.5) -- the tables hold those integers.
HUE_IN = np.array([0, 5, 13, 54, 135, 226, 37, 8, 234, 90, 255, 1], np.float32).reshape(2, 2, 3)
HUE_CASES = [(-0.25, np.array([0, 13, 1, 54, 226, 59, 8, 234, 150, 255, 39, 1], np.float32).reshape(2, 2, 3)),
             (0.25, np.array([13, 0, 11, 226, 54, 221, 234, 8, 92, 1, 217, 255], np.float32).reshape(2, 2, 3))]
# testDoubleContrastFloat: the same picture / 255, contrast_factor 2 about the PER-CHANNEL mean (45.25, 100.75, 118.5)
CONTRAST_IN = HUE_IN
CONTRAST_FACTOR = 2.0
CONTRAST_OUT = np.array([-45.25, -90.75, -92.5, 62.75, 169.25, 333.5, 28.75, -84.75, 349.5, 134.75, 409.25, -116.5], np.float32).reshape(2, 2, 3)
# testDoubleContrastUint8: the u8 form of the same case (saturating conversion of the table above)
CONTRAST_OUT_U8 = np.array([0, 0, 0, 62, 169, 255, 28, 0, 255, 135, 255, 0], np.float32).reshape(2, 2, 3)

# ---- tensorflow/contrib/image/python/kernel_tests/image_ops_test.py, test_rotate_even / test_rotate_odd ---------------------------
# range(36) as 6 x 6 (range(25) as 5 x 5) turned by pi / 2: counter-clockwise about ((w - 1) / 2, (h - 1) / 2); at a quarter turn every
# source coordinate is an integer, so the BILINEAR mode the augmentor uses returns the same table as the test's NEAREST
ROTATE_EVEN_OUT = np.array([[5, 11, 17, 23, 29, 35], [4, 10, 16, 22, 28, 34], [3, 9, 15, 21, 27, 33], [2, 8, 14, 20, 26, 32],
                            [1, 7, 13, 19, 25, 31], [0, 6, 12, 18, 24, 30]], np.float32)
ROTATE_ODD_OUT = np.array([[4, 9, 14, 19, 24], [3, 8, 13, 18, 23], [2, 7, 12, 17, 22], [1, 6, 11, 16, 21], [0, 5, 10, 15, 20]], np.float32)

# ---- tensorflow/core/kernels/crop_and_resize_op_test.cc -----------------------------------------------------------------------------
# the 2 x 2 picture [[1, 2], [3, 4]]: TestCropAndResize2x2To3x3 (whole picture), 2x2To1x1 (one sample at the box centre), 2x2To3x3Flipped
# (box [1, 1, 0, 0] mirrors both axes), 2x2To3x3Extrapolated (box [-1, -1, 1, 1]: samples outside the picture take the extrapolation value, 0 here)
CROP_IN = np.array([1., 2., 3., 4.], np.float32).reshape(1, 2, 2, 1)
CROP_CASES = [([0., 0., 1., 1.], 3, [1, 1.5, 2, 2, 2.5, 3, 3, 3.5, 4]),
              ([0., 0., 1., 1.], 1, [2.5]),
              ([1., 1., 0., 0.], 3, [4, 3.5, 3, 3, 2.5, 2, 2, 1.5, 1]),
              ([-1., -1., 1., 1.], 3, [0, 0, 0, 0, 1, 2, 0, 3, 4])]

# ---- tensorflow/python/training/momentum_test.py, testBasic (doBasic) ------------------------------------------------------------------
# learning_rate 2.0, momentum 0.9, var0 = [1, 2] with gradient [0.1, 0.1] every step: accum = 0.9 * accum + g; var -= lr * accum
#   step 1: accum 0.1, var0 = [1 - 0.1 * 2, 2 - 0.1 * 2];  step 2: accum 0.9 * 0.1 + 0.1, var0 = [1 - 0.1 * 2 - (0.9 * 0.1 + 0.1) * 2, ...]
MOMENTUM_LR, MOMENTUM_M = 2.0, 0.9
MOMENTUM_VAR0, MOMENTUM_GRAD = np.array([1.0, 2.0], np.float32), np.array([0.1, 0.1], np.float32)
MOMENTUM_AFTER = [np.array([1.0 - 0.1 * 2.0, 2.0 - 0.1 * 2.0], np.float32),
                  np.array([1.0 - 0.1 * 2.0 - (0.9 * 0.1 + 0.1) * 2.0, 2.0 - 0.1 * 2.0 - (0.9 * 0.1 + 0.1) * 2.0], np.float32)]
MOMENTUM_ACCUM = [np.array([0.1, 0.1], np.float32), np.array([0.9 * 0.1 + 0.1, 0.9 * 0.1 + 0.1], np.float32)]

# ---- tensorflow/python/kernel_tests/conv_ops_test.py, testConv2D2x2Filter --------------------------------------------------------------
# input 1 .. 18 as [1, 2, 3, 3], filter 1 .. 36 as [2, 2, 3, 3] (HWIO), stride 1, VALID: cross-correlation (no tap flip), 6 outputs.
# (With SAME padding a 2 x 2 filter pads bottom / right only: the VALID outputs are the [0:1, 0:2] corner of the SAME result.)
CONV_IN = np.arange(1, 19, dtype=np.float32).reshape(1, 2, 3, 3)
CONV_FILTER_HWIO = np.arange(1, 37, dtype=np.float32).reshape(2, 2, 3, 3)
CONV_VALID_OUT = np.array([2271., 2367., 2463., 2901., 3033., 3165.], np.float32).reshape(1, 1, 2, 3)

# ---- tensorflow/python/kernel_tests/pooling_ops_test.py, _testMaxPoolSamePadding / _testAvgPoolSamePadding -----------------------------
# input 1 .. 18 as [1, 2, 3, 3], 2 x 2 window, stride 2, SAME: the second window only covers the last column (the padding is on the right), -> 13 .. 18
MAXPOOL_SAME_IN = np.arange(1, 19, dtype=np.float32).reshape(1, 2, 3, 3)
MAXPOOL_SAME_OUT = np.array([13., 14., 15., 16., 17., 18.], np.float32).reshape(1, 1, 2, 3)
# input 1 .. 24 as [1, 2, 4, 3], 2 x 2 average, stride 2, SAME -> 8.5, 9.5, 10.5, 14.5, 15.5, 16.5
AVGPOOL_SAME_IN = np.arange(1, 25, dtype=np.float32).reshape(1, 2, 4, 3)
AVGPOOL_SAME_OUT = np.array([8.5, 9.5, 10.5, 14.5, 15.5, 16.5], np.float32).reshape(1, 1, 2, 3)

# ---- tensorflow/docs_src/api_guides/python/nn.md ("Convolution": the SAME / VALID diagram) ---------------------------------
# input width 13, filter width 6, stride 5:  VALID keeps 2 windows and drops 12, 13;  SAME pads 1 left and 2 right -> 3 windows:
#     pad| 0 |1 2 3 4 5 6 7 8 9 10 11 12 13| 0 0 |pad       out = ceil(13 / 5) = 3, total = (3 - 1) * 5 + 6 - 13 = 3, before = 3 // 2
SAME_PAD_CASES = [  # (in, k, stride, dil) -> (out, pad_before, pad_after)
    ((13, 6, 5, 1), (3, 1, 2)),
    ((300, 3, 1, 1), (300, 1, 1)),        # SSD300 3x3 / s1
    ((19, 3, 2, 1), (10, 1, 1)),          # conv8_2: 19 -> 10 (odd input, total 2)
    ((10, 3, 2, 1), (5, 0, 1)),           # conv9_2: 10 -> 5  (even input: the extra cell goes bottom / right only)
    ((75, 2, 2, 1), (38, 0, 1)),          # pool3: 75 -> 38
    ((19, 3, 1, 2), (19, 2, 2)),          # conv6, dilation 2 (effective kernel 5)
]

# ---- tensorflow/python/ops/nn_fused_batchnorm_test.py, _training_ref ---------------------------------------------------------
# y is normalised with the BIASED batch variance; the variance handed to the moving average is the UNBIASED one
# (var * n / max(n - 1, 1)); tf.layers.batch_normalization: moving = moving * momentum + batch * (1 - momentum), momentum 0.99,
# epsilon 1e-3.  Smallest case: one channel, values 1 and 3 -> mean 2, biased variance 1, unbiased 2.
BN_X = np.array([1., 3.], np.float32).reshape(2, 1, 1, 1)
BN_EXPECT = dict(mean=2.0, var_biased=1.0, var_unbiased=2.0, y=[-1.0 / np.sqrt(1.0 + 1e-3), 1.0 / np.sqrt(1.0 + 1e-3)],
                 moving_mean=0.0 * 0.99 + 2.0 * 0.01, moving_var=1.0 * 0.99 + 2.0 * 0.01)
