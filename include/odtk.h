/* libodtk -- C-ABI of the MI355X-native (gfx950) detector hot path.
 *
 * This is the drop-in boundary beneath the reference's per-model Python class
 * surface (SSD300(config, data_provider).train_one_epoch / .test_one_image,
 * /root/reference/SSD300.py:12-50,473-488).  The reference has no FFI of its own:
 * every entry point below replaces the TensorFlow-1.13 op(s) that the cited
 * reference line invokes through sess.run.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: pointers, ints, floats; no C++/torch types cross this boundary;
 *  - every pointer is a DEVICE pointer owned by the caller unless stated;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *    is enqueued asynchronously, nothing synchronises;
 *  - return 0 on success, non-zero ODTK_ERR_* otherwise; odtk_last_error() gives
 *    the message (thread-local); no exception crosses the ABI;
 *  - activations are NHWC, "rows x channel-pitch": row m = (n*H + h)*W + w,
 *    element (m, c) at base[m*ld + c]; conv weights are [Cout][R][S][Cin] (KRSC);
 *  - boxes are (y, x) ordered in input-pixel units exactly as the reference.
 */
#ifndef ODTK_H_
#define ODTK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODTK_OK 0
#define ODTK_ERR_ARG 1
#define ODTK_ERR_HIP 2

#define ODTK_BF16 0 /* bfloat16 storage, MFMA 32x32x16 bf16, f32 accumulate */
#define ODTK_F32 1  /* float32 storage, MFMA 32x32x2 f32 (exact f32 FMA chain)  */
/* Convolution descriptors only (round 4): float32 STORAGE on every side (x, w, y, dy, dx, dw: exactly the ODTK_F32 layouts and pitches), arithmetic by OPERAND
 * SPLITTING on the bf16 MFMA where that is faster -- a = a_hi + a_lo (two bf16), a b = a_hi b_hi + a_hi b_lo + a_lo b_hi, accumulated in f32: one implicit GEMM with
 * a three times longer reduction at 16x the f32 MFMA rate; products carry 2^-17 instead of 2^-24 (bf16 operands: 2^-9).  The split copies of both operands are
 * made per call in a library-owned arena (per device and odtk_scratch_slot).  Layers where the split passes cost more than the exact kernel takes (C K R S < 20 000,
 * the 7..112-channel backbone layers of RetinaNet.py:594-643) and geometries the bf16 kernels do not cover run on the ODTK_F32 kernels, bit for bit.  dtype and
 * out_dtype must both be ODTK_F32X3; every other entry point takes ODTK_F32 for these tensors. */
#define ODTK_F32X3 2

const char* odtk_last_error(void);
int odtk_version(void);
/* host-side CRC32C (Castagnoli) of n bytes continuing from `crc` (0 to start): the checksum inside TensorFlow checkpoint
 * files, for the reader / writer that stands in for tf.train.Saver and NewCheckpointReader (SSD300.py:31, :464-504) */
unsigned int odtk_crc32c(const void* data, long long n, unsigned int crc);
/* number of compute units / name of the current device (host out pointers) */
int odtk_device_info(int* num_cu, char* name_buf, int name_buf_len);
/* test/debug knobs: key 0 = force the register-staged conv gather kernel (value != 0);
 * key 1 = conv engine: 0 auto, 1 legacy 4-wave kernels, 2 8-wave v3 wherever supported;
 * key 2 = A/B bits of the conv kernels: bits 0-3 ablations (skip DMA / one slab; RESULTS WRONG), 5 no fragment double
 *         buffering, 6 no early/late DMA stagger, 7 "landed early" protocol, 8 64-bit global addressing for the LDS-DMA,
 *         9 no XCD remap (wgrad), 10 s_setprio, 11 no 64->64 / first-layer halo kernels, 12 per-lane tap walk,
 *         13 no split-K, 14 interleaved slab body, 15 block-pair halo filter gradient on any size (tests), 16 no raster-run halo kernel (v6),
 *         17 no block-pair halo filter gradient (the 64x64 halo wgrad kernel only where C = K = 64),
 *         18-25 = n: halo kernel instead of split-K on layers with >= n tiles (0 = split-K policy as is), 26 no wide (W <= 159) halo
 *         variant, 27 no 128x512 halo tiles, 28 no 128x192 halo tiles, 29 no four-wave filter-gradient kernel (v8), 30 v8 also on
 *         short pixel ranges;
 * key 3 = single-kernel NMS (value != 0);
 * key 4 = batch norm: maps of up to `value` rows run statistics + finalize + apply in ONE launch (default 1024, 0 = never); larger maps take three
 *         (statistics, finalize, apply); value -2 = two (apply with the finalize folded in: measured slower, A/B only), -1 = back to three;
 *         -3 / -4 the one-launch kernels in their 64-channel shape only / back; -5 / -6 never pick the two-launch path by shape / back;
 *         -20 / -21 / -22 / -23 rows per workgroup of the apply passes 128 / 512 / 1 024 / by shape (the default: 256, 1 024 for 8- / 16-channel bf16 maps);
 * key 5 = filter gradient flush: 1 (THE DEFAULT since round 6) = deterministic -- every filter-gradient kernel stores partial tiles (one per pixel split /
 *         workgroup / wave) and a reduction launch adds them in a fixed order; the scalar gamma gradient of odtk_l2norm_bwd and RetinaNet's loss sums follow:
 *         whole training steps of every model class are reproducible bit for bit (tools/step_determinism.py); 0 = float atomics into dw (the default of
 *         rounds 1-5).  Same step time to 0.1 % on SSD300 batch 32 (gpurun r6e: the partial stores are cheaper than the atomics, the reduction is ~18 us);
 * key 6 = dispatch A/B switches of the convolution kernels that leave results intact (up to the engines' stated tolerances): bit 2 (4) = ODTK_F32X3 descriptors run
 *         on the exact f32 kernels, bit 3 (8) = ... on the split path wherever it is supported, also below the size policy (tests), bit 4 (16) = no 32-row filter
 *         tile in the f32 LDS-DMA gather, bit 5 (32) = narrow f32 filter gradients on the legacy kernel; round 5: bit 6 (64) = no small-map gather kernel
 *         (conv_v9.hip), bit 7 (128) = the small-map kernel wherever it is supported (tests), bit 8 (256) = no chunk-range split-K on the raster-run halo kernel,
 *         bit 9 (512) = the small-map kernel always on its four-stage ring, bit 10 (1024) = no halo kernel on C % 64 != 0, bit 12 (4096) = float atomics also for
 *         filter gradients with ONE pixel split, bit 13 (8192) = stride-2 input gradients always on the small-map kernel (never the 8-wave kernel's phase launch),
 *         bit 14 (16384) = no 128 x 128- / 64 x 128-tile launch of the raster-run halo kernel (the chunk-range split-K instead), bit 15 (32768) = no 64 x 128 tiles;
 *         bit 17 (131072) = EXPERIMENT, changes results: the ODTK_F32X3 splits write zeros for the low halves -- f32 tensors with ONE bf16 product per f32 product,
 *         the numerics of a cheaper mixed engine at the x3 engine's cost (tools/gate_table.py ... f32x1sim; profiles/r06l_f32x1_numerics_experiment.md);
 * key 7 = group norm: maps of up to `value` pixels per sample run statistics + apply in ONE launch (default 1024, 0 = never) */
int odtk_debug_set(int key, int value);
/* Library-owned scratch (the split-K partial tiles of the small-map convolutions) is one buffer per (device, slot), handed to
 * every later call of this thread until the slot changes.  Calls on ONE stream are ordered and share slot 0; a caller that
 * launches convolutions on several streams CONCURRENTLY (SSD300: the heads beside the extra-layer chain) selects a different
 * slot (0..3) per stream before launching on it.  A buffer that a stream capture has been handed is never freed or moved (the captured graph
 * points into it); one that no capture has seen is freed when a later call outgrows it (after a device synchronize: first steps only). */
int odtk_scratch_slot(int slot);
/* name of the device kernel the last odtk_conv2d_* call of this thread dispatched to (bench.py attributes
 * its HIP-event timings to kernels with it, so the roofline line and the rocprofv3 trace name the same kernel) */
const char* odtk_conv_last_kernel(void);

/* ------------------------------------------------------------------------- *
 * Convolution family: replaces tf.nn.conv2d (SSD300.py:519) and
 * tf.layers.conv2d (SSD300.py:524) forward, and the Conv2DBackpropInput /
 * Conv2DBackpropFilter ops that tf.gradients adds (SSD300.py:154).
 * Implicit GEMM on MFMA, TF "SAME" padding expressed as explicit pad_t/pad_l.
 * ------------------------------------------------------------------------- */
typedef struct odtk_conv_desc {
    int N, H, W, C;      /* input  dims; C = GEMM-K channels, multiple of 8 (bf16) / 4 (f32) */
    int ldx;             /* input row pitch in elements (>= C)                                */
    int Ho, Wo, K;       /* output dims; K = Cout                                             */
    int ldy;             /* output row pitch in elements (>= K)                               */
    int R, S;            /* filter taps                                                       */
    int stride, dil;     /* forward stride / dilation                                         */
    int pad_t, pad_l;    /* TF SAME: pad_before (bottom/right take the extra cell)            */
    int dtype;           /* ODTK_BF16 / ODTK_F32: storage of x, w (and dy)                    */
    int out_dtype;       /* storage of the forward output y / dgrad output dx                 */
} odtk_conv_desc;

/* y[m, k] = act( sum_{r,s,c} x[n, ho*stride-pad_t+r*dil, wo*stride-pad_l+s*dil, c] * w[k,r,s,c] + bias[k] )
 * bias may be NULL; relu != 0 applies max(.,0).  Columns >= K of y are not written. */
int odtk_conv2d_fwd(const odtk_conv_desc* d, const void* x, const void* w, const float* bias,
                    void* y, int relu, void* stream);

/* odtk_conv2d_fwd followed by tf.layers.max_pooling2d(2, 2, 'same') in ONE launch where the kernel can (conv + bias + ReLU + pool1 of the VGG trunk,
 * SSD300.py:201-209: the 369 MB bf16 map of conv1_2 at batch 32 is then never written nor read back).  y_pool [N][ceil(Ho/2)][ceil(Wo/2)][ld_pool]
 * receives the pooled map, idx the recorded first arg-max in the format of odtk_maxpool2x2_fwd_idx (may be NULL outside training).  y may be NULL:
 * the un-pooled output is then not stored at all -- only valid when nothing else reads it (training: the pool routes the gradient by idx, and
 * the ReLU mask of the routed positions is the sign of the POOLED value).  Shapes the fused kernel does not cover run as the two launches
 * (y must then be given; ODTK_ERR_ARG otherwise) -- odtk_conv2d_fwd_pool2x2_fused(d) tells which. */
int odtk_conv2d_fwd_pool2x2(const odtk_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int relu,
                            void* y_pool, int ld_pool, void* idx, void* stream);
int odtk_conv2d_fwd_pool2x2_fused(const odtk_conv_desc* d);

/* ReLU mask as SIGN BITS (round 4).  tf.nn.relu's gradient (SSD300.py:199-200 behind conv1_1) needs one bit per activation; conv1_2's input-gradient
 * pass read the whole 369-MB bf16 activation of batch 32 for it.  odtk_conv2d_fwd_bits = odtk_conv2d_fwd that also writes relu_bits [M][ldy / 8] bytes
 * (bit e of byte (m, j) = y[m][8 j + e] > 0); odtk_conv2d_dgrad_bits = odtk_conv2d_dgrad masked by such bits instead of relu_src.  Only the kernel pair
 * first layer (3(8) -> 64) -> 64 -> 64 halo kernel implements them: odtk_conv2d_relu_bits_supported(producer, consumer, consumer's lddy) != 0;
 * ODTK_ERR_ARG otherwise. */
int odtk_conv2d_relu_bits_supported(const odtk_conv_desc* producer, const odtk_conv_desc* consumer, int consumer_lddy);
int odtk_conv2d_fwd_bits(const odtk_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int relu, void* relu_bits, void* stream);
int odtk_conv2d_dgrad_bits(const odtk_conv_desc* d, const void* dy, int lddy, const void* w_t, const void* relu_bits, void* dx, int accumulate, void* stream);

/* bit 0: the forward pass of this ODTK_F32X3 descriptor runs as split bf16 products; bit 1: the filter gradient does (C % 8 == 0 on top); bit 2: the input
 * gradient does (as the forward pass, plus every 3x3 stride-2 layer: its exact kernel is the slow one); 0 for every other descriptor.  Introspection only
 * (benchmarks, tests): the conv entry points decide by themselves. */
int odtk_conv2d_x3_supported(const odtk_conv_desc* d);

/* dx[n,h,w,c] (+)= sum_{r,s,k} dy[n,ho,wo,k] * w[k,r,s,c]   (transposed conv of the fwd op)
 * w_t is the dgrad-layout filter produced by odtk_filter_to_dgrad: [C][R][S][Kp]
 * with taps flipped and Kp = dy pitch channels.  If relu_src != NULL the result is
 * masked by (relu_src[m, c] > 0) -- the fused ReLU backward of the producer layer
 * (relu_src has the geometry/pitch of dx and dtype d->dtype).  accumulate != 0
 * adds into dx (sum over several consumers).  dx dtype = d->out_dtype. */
int odtk_conv2d_dgrad(const odtk_conv_desc* d, const void* dy, int lddy, const void* w_t,
                      const void* relu_src, void* dx, int accumulate, void* stream);

/* dw[k,r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,k] * x[n,hi,wi,c]   (float32, KRSC, pitch R*S*C)
 * dw must be zeroed (or hold the value to accumulate onto) by the caller.  dbias (optional,
 * f32 [K]) += column sums of dy -- the bias gradient, fused so dy is not read a second time.
 * How the sum reaches dw depends on the launch: several pixel splits go through partial tiles in
 * the library's scratch and a fixed-order reduction launch (the default, deterministic; with
 * odtk_debug_set key 5 = 0: float atomics); ONE pixel split adds its tile with plain 16-byte
 * read-add-store rows.
 * Hence: (1) launches that accumulate into the SAME dw (a head whose filter is shared by
 * several pyramid levels) must be ordered on one stream or by events -- two of them in flight
 * at once race; (2) dw should be 16-byte aligned: an unaligned dw is served by the atomic
 * flush only, and deterministic mode returns an error for it. */
int odtk_conv2d_wgrad(const odtk_conv_desc* d, const void* x, const void* dy, int lddy,
                      float* dw, float* dbias, void* stream);

/* w [K][R][S][C] (f32 master) -> w_t [C][R][S][Kp] in `dtype`, taps flipped, zero padded
 * to Kp; and optionally a straight cast copy w_c [K][R][S][C] in `dtype` (may be NULL). */
int odtk_filter_prepare(const float* w, int K, int R, int S, int C, int Kp, int dtype,
                        void* w_c, void* w_t, void* stream);

/* The same w -> w_t transform for MANY layers in one launch (the optimizer refreshes every dgrad
 * filter each step).  `items_dev` is a DEVICE array of n_items odtk_fp_item; item i owns the blocks
 * [block_begin, block_begin + ktiles*R*S*ctiles) with ctiles = ceil(C/32), ktiles = ceil(Kp/64);
 * total_blocks is the sum. */
typedef struct odtk_fp_item {
    const float* w;      /* f32 master [K][R*S][C]              */
    void* w_t;           /* out [C][R*S flipped][Kp] in `dtype` */
    int K, RS, C, Kp;
    int block_begin, ctiles, ktiles, pad_;
} odtk_fp_item;
int odtk_filter_prepare_batched(const void* items_dev, int n_items, int total_blocks, int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * Elementwise / reduction layers (all NHWC rows x pitch)
 * ------------------------------------------------------------------------- */
/* images f32 [N,H,W,3] (RGB 0..255) -> x[N*H*W][ldx] = img - mean (SSD300.py:52-63),
 * channels 3..ldx-1 zero.  dtype selects bf16/f32 output. */
int odtk_preprocess(const float* images, long long pixels, const float* mean3, int ldx, int dtype,
                    void* x, void* stream);

/* CenterNet's input transform (CenterNet.py:51-65): x = (images / div - mean) / std per RGB channel, in that float32 order;
 * same layout contract as odtk_preprocess (pixels x 3 floats in, pixels x ldx elements out, pad channels zero). */
int odtk_preprocess_norm(const float* images, long long pixels, float div, const float* mean3, const float* std3, int ldx, int dtype,
                         void* x, void* stream);
/* tf.layers.average_pooling2d(2, 2, 'same') on even maps (CenterNet.py:423-431): y [N,H/2,W/2,ld] = mean of the 2x2 window;
 * backward: every input of a window receives dy / 4 (dx fully written).  All ld columns are processed. */
int odtk_avgpool2x2_fwd(const void* x, void* y, int N, int H, int W, int ld, int dtype, void* stream);
int odtk_avgpool2x2_bwd(const void* dy, void* dx, int N, int H, int W, int ld, int dtype, void* stream);

/* y = relu(a + b) on rows with their own pitches (RefineDet's transfer-connection block, RefineDet.py:371), and the gradient of a ReLU taken
 * from its OUTPUT: dx[m][c] (+= when accumulate) = dy[m][c] where y[m][c] > 0.  dy shares y's pitch. */
int odtk_add_relu_fwd(const void* a, int lda, const void* b, int ldb, void* y, int ldy, long long M, int C, int dtype, void* stream);
int odtk_relu_bwd(const void* y, const void* dy, int ldy, void* dx, int lddx, long long M, int C, int dtype, int accumulate, void* stream);

/* tf.layers.max_pooling2d SAME (SSD300.py:539-547): kxk window, stride, pad_before. */
int odtk_maxpool_fwd(const void* x, void* y, int N, int H, int W, int C, int ld, int Ho, int Wo,
                     int k, int stride, int pad_t, int pad_l, int dtype, void* stream);
/* gradient routed to the FIRST maximum of each window (row-major scan), summed over
 * overlapping windows; dx fully written. */
int odtk_maxpool_bwd(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W,
                     int C, int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                     int dtype, void* stream);
/* Max pooling with OVERLAPPING windows (pool5 = tf.layers.max_pooling2d(3, 1, 'same'), SSD300.py:303) and a recorded arg-max: arg holds one uint32 per
 * 16-byte output chunk, 4 bits per channel = the position r * k + s (k <= 3) of the first maximum of the window in scan order -- TF's MaxPoolGrad
 * routing; odtk_maxpool_bwd_argmax then gathers dy over the windows that contain a pixel from arg + dy alone.  Results are identical to
 * odtk_maxpool_fwd / odtk_maxpool_bwd. */
int odtk_maxpool_fwd_argmax(const void* x, void* y, void* arg, int N, int H, int W, int C, int ld, int Ho, int Wo, int k, int stride,
                            int pad_t, int pad_l, int dtype, void* stream);
int odtk_maxpool_bwd_argmax(const void* arg, const void* dy, void* dx, int N, int H, int W, int C, int ld, int Ho, int Wo, int k, int stride,
                            int pad_t, int pad_l, int dtype, void* stream);
/* 2x2 / stride 2 / SAME (pad_before 0) pooling with a recorded arg-max (pool1..pool4): idx holds one uint16 per 16-byte
 * output chunk (2 bits per channel = the first window position holding the maximum, TF's gradient routing); the backward
 * pass then reads dy + idx only.  Results are identical to odtk_maxpool_fwd / _bwd. */
int odtk_maxpool2x2_fwd_idx(const void* x, void* y, void* idx, int N, int H, int W, int C, int ld, int Ho, int Wo,
                            int dtype, void* stream);
int odtk_maxpool2x2_bwd_idx(const void* idx, const void* dy, void* dx, int N, int H, int W, int C, int ld, int Ho,
                            int Wo, int dtype, void* stream);

/* tf.layers.batch_normalization, fused semantics (SSD300.py:506-512), momentum .99 eps 1e-3.
 * z [M][ldz] (dtype) -> y.  Output row m is written at y + (m / rows_per_img)*y_img_stride
 * + (m % rows_per_img)*ldy (lets the head write straight into pred [N,8828,25]).
 * relu: 0 none, 1 tf.nn.relu, 2 tf.nn.leaky_relu(., 0.1) (YOLOv3.py:505); the backward masks / scales dy by the sign of y.
 * training: batch statistics (biased var), saves mean / inv-std (f32 [C]) for backward and
 * updates moving stats with the unbiased variance.  Inference: uses moving stats.
 * workspace: >= odtk_bn_workspace_bytes(M, C) bytes (always required). */
long long odtk_bn_workspace_bytes(int M, int C);
int odtk_bn_fwd(const void* z, int M, int C, int ldz, int dtype, const float* gamma,
                const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                float* save_invstd, int training, int relu, void* y, int y_dtype, int ldy,
                int rows_per_img, long long y_img_stride, void* workspace, void* stream);
/* dz from dy (same addressing rule as y); if relu, dy is first masked by (y > 0).
 * dgamma/dbeta (f32 [C]) are overwritten. dz pad columns (C..ldz) are zeroed. */
int odtk_bn_bwd(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype,
                int y_dtype, int ldy, int rows_per_img, long long y_img_stride,
                const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                void* dz, float* dgamma, float* dbeta, void* workspace, void* stream);

/* Batch norm over the GLOBAL batch of a data-parallel job (SURVEY.md 8e option B): odtk_bn_fwd / odtk_bn_bwd split where the
 * replicas exchange per-channel numbers, so that W ranks with B images each compute exactly what one device computes on W*B.
 *   odtk_bn_moments   : mean[C], var[C] (biased) of the local rows.
 *   (host) all-gather -> moments [W][2][C] = every replica's (mean, var); equal row counts per replica.
 *   odtk_bn_fwd_given : combines them (parallel-variance formula), stores save_mean / save_invstd, updates the moving statistics
 *                       with the unbiased variance over W*M rows, applies scale / offset / activation like odtk_bn_fwd.
 *   odtk_bn_bwd_sums  : sums[0:C] = sum dy' (this replica's dbeta), sums[C:2C] = sum dy' * xhat (its dgamma).
 *   (host) all-reduce(sum) of a COPY of sums -> sums_global; count = W*M.
 *   odtk_bn_bwd_given : dz with the global means sums_global / count.
 * Arguments shared with odtk_bn_fwd / odtk_bn_bwd have the same meaning; workspace as odtk_bn_workspace_bytes. */
int odtk_bn_moments(const void* z, int M, int C, int ldz, int dtype, float* mean, float* var, void* workspace, void* stream);
int odtk_bn_fwd_given(const void* z, int M, int C, int ldz, int dtype, const float* gamma, const float* beta,
                      const float* moments, int replicas, float* moving_mean, float* moving_var, float* save_mean,
                      float* save_invstd, int relu, void* y, int y_dtype, int ldy, int rows_per_img, long long y_img_stride,
                      void* workspace, void* stream);
int odtk_bn_bwd_sums(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype, int y_dtype, int ldy,
                     int rows_per_img, long long y_img_stride, const float* save_mean, const float* save_invstd, int relu,
                     float* sums, void* workspace, void* stream);
int odtk_bn_bwd_given(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype, int y_dtype, int ldy,
                      int rows_per_img, long long y_img_stride, const float* gamma, const float* save_mean,
                      const float* save_invstd, int relu, const float* sums_global, long long count, void* dz, void* workspace,
                      void* stream);

/* Glue of the residual / pyramid detectors (SURVEY.md 8f.1 backbones).  Rows are [M][ld] with their own pitch, so a
 * channel slice of a concat buffer is an ordinary operand; C and every pitch multiples of 16 bytes, pointers 16-byte aligned.
 * odtk_add2d: y = a + b (b NULL: pitched copy; y may alias a: accumulate) -- `conv = conv + conv2` (YOLOv3.py:489-491),
 * tf.concat halves and their gradients (:412).  odtk_upsample2x_*: tf.image.resize_nearest_neighbor to twice the size
 * (:411) and its gradient (sum of the four copies; accumulate != 0 adds to dx). */
int odtk_add2d(const void* a, int lda, const void* b, int ldb, void* y, int ldy, long long M, int C, int dtype, void* stream);
int odtk_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream);
int odtk_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int dtype, int accumulate,
                        void* stream);

/* tf.image.resize_bilinear(x, [Ho, Wo]) on NHWC rows, TF-1.x grid (align_corners=False: src = dst * in / out): the top-down path of
 * the RetinaNet / FCOS pyramids (RetinaNet.py:309, FCOS.py:373), and its gradient (a gather, no atomics; up-scaling only).
 * accumulate != 0 adds to y / dx -- `feat + resize(top_feat)` (RetinaNet.py:310) in one pass.  Pitch rules as odtk_add2d. */
int odtk_resize_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                             int accumulate, void* stream);
int odtk_resize_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                             int accumulate, void* stream);
/* The same with TensorFlow's `align_corners` switch (src = dst * (in - 1) / (out - 1)) and any scaling, down as well as up: PFPNetR resizes
 * conv4_3 to 1/2, 1/4, 1/8 with align_corners=True (PFPNetR.py:320-322).  relu_src (NULL or the forward input, rows as dx): the input is a
 * bias + ReLU activation whose gradient buffer holds d(pre-activation) -- the gradient is zeroed where the activation is <= 0. */
int odtk_resize_bilinear2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                              int align_corners, int accumulate, void* stream);
int odtk_resize_bilinear2_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                              int align_corners, int accumulate, const void* relu_src, void* stream);
/* tf.concat over channels (and its gradient) for pieces that do not start on 16-byte channel boundaries -- PFPNetR's 512 + 85 + 85 + 85
 * features (PFPNetR.py:366-396): dst[m][dst_off + c] (+)= src[m][src_off + c] for c < C; pitches and offsets in ELEMENTS, any alignment;
 * relu_src (NULL or a tensor with the rows of dst) zeroes the copy where relu_src[m][dst_off + c] <= 0. */
int odtk_copy_channels(const void* src, int lds, int src_off, void* dst, int ldd, int dst_off, long long M, int C, int dtype,
                       int accumulate, const void* relu_src, void* stream);

/* tf.contrib.layers.group_norm(groups, epsilon 1e-6) (+ ReLU when relu != 0) on NHWC rows [N*HW][ld]: the normalisation of the
 * reference's FCOS (FCOS.py:438-446).  Statistics per sample and group over HW x (C / groups) elements; save_mean_rstd [N][groups][2]
 * (may be NULL in inference).  Backward: `accumulate` bit 0 adds to dx instead of overwriting it, bit 1 adds to dgamma / dbeta [C] (a norm
 * whose parameters serve several activations: FCOS shares its heads over the pyramid levels, FCOS.py:351,358); the ReLU mask comes from the
 * sign of y.  workspace: odtk_gn_workspace_bytes(N, C). */
long long odtk_gn_workspace_bytes(int N, int C);
int odtk_gn_fwd(const void* x, int ldx, void* y, int ldy, int N, int HW, int C, int groups, int dtype, const float* gamma,
                const float* beta, int relu, float* save_mean_rstd, void* stream);
int odtk_gn_bwd(const void* x, int ldx, const void* y, const void* dy, int ldy, void* dx, int lddx, int N, int HW, int C, int groups,
                int dtype, const float* gamma, const float* save_mean_rstd, int relu, int accumulate, float* dgamma, float* dbeta,
                void* workspace, void* stream);

/* tf.exp on the distance outputs of the FCOS regression heads (FCOS.py:363), laid out for odtk_fcos_loss: y f32 [M][C] = exp(x rows),
 * and its chain rule dx rows = dy * y (pad columns of dx zeroed). */
int odtk_exp_rows_to_f32(const void* x, int ldx, int dtype, float* y, long long M, int C, void* stream);
int odtk_exp_rows_bwd(const float* dy, const float* y, void* dx, int lddx, int dtype, long long M, int C, void* stream);

/* conv rows <-> the f32 prediction tensors of the box-side kernels: tf.reshape + tf.concat over the pyramid levels
 * (RetinaNet.py:184-186, :321-326).  Row m of image n = m / rows_per_img is read / written at
 * y + n * y_img_stride + (m % rows_per_img) * ldy (floats); x is [M][ldx] in `dtype`; _from_f32 zeroes x's pad columns. */
int odtk_rows_to_f32(const void* x, int ldx, int dtype, float* y, int ldy, int rows_per_img, long long y_img_stride, long long M, int C,
                     void* stream);
int odtk_rows_from_f32(const float* y, int ldy, int rows_per_img, long long y_img_stride, void* x, int ldx, int dtype, long long M, int C,
                       void* stream);

/* tf.nn.l2_normalize(axis=C) * scalar gamma (SSD300.py:74-83). */
int odtk_l2norm_fwd(const void* x, void* y, int M, int C, int ld, int dtype, const float* gamma,
                    void* stream);
/* dx += (accumulate) ; dgamma[0] += sum (caller zeroes). relu_src masks like dgrad.
 * Deterministic mode (odtk_debug_set key 5): the block sums go through a small buffer per (device, scratch slot): launches in
 * flight at once must come from threads on different slots (odtk_scratch_slot), like the convolutions' scratch. */
int odtk_l2norm_bwd(const void* x, const void* dy, void* dx, int M, int C, int ld, int dtype,
                    const float* gamma, float* dgamma, int accumulate, const void* relu_src,
                    void* stream);

/* column sums: out[c] (+)= sum_m dy[m][c]  (bias gradient), f32 out. */
int odtk_colsum(const void* dy, int M, int C, int ld, int dtype, float* out, int accumulate,
                void* workspace, void* stream);

/* Fused MomentumOptimizer(0.9) + L2 weight decay over one flat f32 parameter buffer
 * (SSD300.py:149-154): g = grad*grad_scale + wd*p; m = mom*m + g; p -= lr*m.
 * l2_partial (optional, f32 [>=odtk_sgd_blocks(n)]) receives per-block sum(p_old^2)/2;
 * p_cast (optional) receives the updated parameters cast to cast_dtype (the bf16 operand copy). */
int odtk_sgd_blocks(long long n);
int odtk_sgd_momentum(float* p, float* m, const float* grad, long long n, float lr, float momentum,
                      float wd, float grad_scale, float* l2_partial, void* p_cast, int cast_dtype,
                      void* stream);
/* tf.train.AdamOptimizer (CenterNet.py:154; ApplyAdam) fused with the L2 term of the loss over one flat f32 parameter buffer:
 * g = grad*grad_scale + wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps), where the caller passes
 * lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) for step t = 1, 2, ...  l2_partial / p_cast as in odtk_sgd_momentum (same block layout). */
int odtk_adam(float* p, float* m, float* v, const float* grad, long long n, float lr_t, float beta1, float beta2, float eps,
              float wd, float grad_scale, float* l2_partial, void* p_cast, int cast_dtype, void* stream);
/* bytes of device memory at p <- 0 on `stream` (the flat gradient buffer at the top of a step: float atomics accumulate into it). */
int odtk_zero(void* p, long long bytes, void* stream);
/* out[0] = sum_{i<n} in[i] (deterministic tree). */
int odtk_sum_f32(const float* in, long long n, float* out, void* stream);
/* The scalar a training step reports, in one launch (SSD300.py:148-152: sum_i loss_i / batch + wd * sum_v ||v||^2 / 2): sum_a = sum_{i<na} a[i * stride_a]
 * (the per-image losses, a column of the loss kernel's [N][4] partials), sum_b = sum_{j<nb} b[j] (the optimizer kernel's per-block sum p^2 / 2 partials),
 * total = scale_a * sum_a + scale_b * sum_b.  sum_a / sum_b may be null.  Fixed summation order. */
int odtk_loss_total(const float* a, int na, int stride_a, const float* b, long long nb, float scale_a, float scale_b, float* sum_a, float* sum_b,
                    float* total, void* stream);
/* cast f32 -> dtype */
int odtk_cast_from_f32(const float* in, void* out, long long n, int dtype, void* stream);
/* cast dtype -> f32 (the bf16 gradient buckets of the data-parallel path are summed by RCCL as bf16 and widened back into the flat f32
 * gradient buffer; replaces nothing in the reference, which is single-device: testSSD300.py:14) */
int odtk_cast_to_f32(const void* in, int dtype, float* out, long long n, void* stream);

/* ------------------------------------------------------------------------- *
 * Box side: priors, matching, loss, NMS, decode
 * ------------------------------------------------------------------------- */
/* SSD300._get_abbox (SSD300.py:323-343) for `nlevels` levels concatenated level-major,
 * then (y, x, anchor).  prior_hw: host-computed float32 (h, w) per level/anchor exactly as
 * the reference builds its python list (level l has na[l] entries, packed).  Outputs are
 * [A][2] each; nmsbox [A][4] = (yx - hw/2, yx + hw/2) as recomputed at SSD300.py:421. */
int odtk_ssd_priors(int input_size, int nlevels, const int* fsize /*host*/, const int* na /*host*/,
                    const float* prior_hw /*host*/, float* y1x1, float* y2x2, float* yx, float* hw,
                    float* nmsbox, void* stream);

/* SSD300._compute_one_image_loss steps 1-7 (SSD300.py:347-426), one workgroup per image.
 * gt [N][P][5] = (yc, xc, h, w, cls), padded rows -1.  Outputs:
 *  ngt[N]; best[N][P] (anchor id per GT, first max); status[N][A] u8: 0 = best anchor of some
 *  GT, 1 = positive (max IoU > 0.5), 2 = negative; rgindex[N][A] (arg-max GT, first max);
 *  counts[N][4] = {num_pos (= ngt + #status1), num_neg, chosen_num_neg, 0}. */
int odtk_ssd_match(const float* y1x1, const float* y2x2, const float* hw, int A, const float* gt,
                   int N, int P, int* ngt, int* best, unsigned char* status, int* rgindex,
                   int* counts, void* stream);

/* per-row softmax cross entropy against one constant label (the background class):
 * loss[n][a] = logsumexp(pred[n][a][0:C]) - pred[n][a][label]   (SSD300.py:427-430) */
int odtk_softmax_ce_const(const float* pred, long long rows, int C, int ld, int label, float* loss,
                          void* stream);

/* tf.image.non_max_suppression (NonMaxSuppressionV3) for B independent problems in one
 * launch (SSD300.py:179, :431).  Problem b: boxes at boxes + b*box_stride (floats) [n][4];
 * score i at scores[b*score_bstride + i*score_estride]; valid[b*valid_bstride + i*valid_estride]
 * (u8, NULL = all valid) must equal `valid_value`; max_out from max_out_dev[b*max_out_stride]
 * (device ints) or, when NULL, max_out_const.  Selected ORIGINAL indices in pick order go to
 * out_idx[b*cap + j], the count to out_cnt[b].  Bit-exact vs the reference kernel for distinct
 * scores; equal scores are visited lower index first.  n <= 32768 (above 16384 a global-memory work area of the single-workgroup kernel is used). */
int odtk_nms_batched(const float* boxes, long long box_stride, const float* scores,
                     long long score_bstride, int score_estride, const unsigned char* valid,
                     long long valid_bstride, int valid_estride, int valid_value, int n, int B,
                     const int* max_out_dev, int max_out_stride, int max_out_const,
                     float iou_threshold, int* out_idx, int cap, int* out_cnt, void* stream);

/* SSD300._compute_one_image_loss steps 8-12 (SSD300.py:427-453) + the batch mean
 * (SSD300.py:148) and its gradient.  loss_parts[N][4] = {neg, pos_conf, coord, total};
 * dpred [N][A][ld] is overwritten with d(sum_i total_i * grad_scale)/dpred. */
int odtk_ssd_loss(const float* pred, int N, int A, int C, int ld, const float* yx, const float* hw,
                  const float* gt, int P, const int* ngt, const int* best,
                  const unsigned char* status, const int* rgindex, const int* counts,
                  const float* negloss, const int* sel_idx, int sel_cap, const int* sel_cnt,
                  float grad_scale, float* loss_parts, float* dpred, void* stream);

/* Inference decode (SSD300.py:157-171): softmax, drop rows whose arg-max is background,
 * decode boxes.  conf [A][C-1], boxes [A][4]; cand[A][C-1] u8 = keep && conf >= thr. */
int odtk_ssd_decode(const float* pred0, int A, int C, int ld, const float* yx, const float* hw,
                    float score_thr, float* conf, float* boxes, unsigned char* keep,
                    unsigned char* cand, void* stream);

/* ------------------------------------------------------------------------- *
 * RetinaNet box side (kernel K16): anchors, matching with the 0.4 / 0.5 ignore band,
 * softmax focal loss + smooth-L1 and their gradients.  Any number of anchors (47 961 @500x500,
 * 120 087 @800x800); same GT layout as SSD ([N][P][5] = yc, xc, h, w, class; pad rows -1; P <= 128).
 * ------------------------------------------------------------------------- */
/* RetinaNet._get_abbox (RetinaNet.py:328-355) for every pyramid level: level l has fh[l] x fw[l] cells and
 * na[l] anchors per cell whose (h, w) follow in prior_hw (host floats, level-major); centre = (i + 0.5) *
 * (input_dim / fh[l]) on BOTH axes (the reference passes data_shape[1] as input_dim, RetinaNet.py:330).
 * Outputs [A][2] each, A = sum fh*fw*na, order (level, y, x, anchor). */
int odtk_retina_anchors(int input_dim, int nlevels, const int* fh, const int* fw, const int* na,
                        const float* prior_hw, float* y1x1, float* y2x2, float* yx, float* hw, void* stream);

/* Matching (RetinaNet.py:357-417).  best[N][P]: first arg-max anchor of every GT; status[N][A]: 0 ignore
 * (0.4 <= IoU <= 0.5), 1 positive (IoU > 0.5), 2 negative (IoU < 0.4), 3 best anchor of some GT;
 * rgindex[N][A]: first arg-max GT of the anchor; counts[N][4] = {rows of the positive set, negatives, min(3 * positives, negatives), 0}.
 * Indices are bit-exact vs the reference's float32 arithmetic.  workspace: odtk_retina_match_workspace_bytes. */
long long odtk_retina_match_workspace_bytes(int A, int N, int P);
int odtk_retina_match(const float* y1x1, const float* y2x2, const float* hw, int A, const float* gt, int N,
                      int P, int* ngt, int* best, unsigned char* status, int* rgindex, int* counts,
                      void* workspace, void* stream);

/* RefineDet, two-stage loss (RefineDet.py:422-567) on top of odtk_retina_match (same matching rule), odtk_softmax_ce_const (ARM background
 * cross entropy of every anchor, 2 classes, label 1) and odtk_nms_batched (hard negatives among status == 2, budget counts[n][2], IoU 0.7).
 * arm_loc / odm_loc [N][A][4] = (ty, tx, th, tw), arm_conf [N][A][2] (class 0 = object), odm_conf [N][A][C] (background = C-1).
 * ARM: mean CE of the mined negatives, mean CE + smooth-L1 of the positive rows against the anchors.  ODM: mean CE against background of the mined
 * negatives whose ARM background LOGIT is < 0.99 (sic), mean CE + smooth-L1 of the positive rows against the ARM-REFINED anchors -- the box term
 * also differentiates the refined anchors (no stop_gradient in the reference).  loss_parts [N][8] = {arm negatives, arm positives, arm boxes,
 * odm negatives, odm positives, odm boxes, total, #odm negatives}; the four gradients (of sum_n total_n * grad_scale) are fully written. */
int odtk_refinedet_loss(const float* arm_loc, const float* arm_conf, const float* odm_loc, const float* odm_conf, int N, int A, int C,
                        const float* yx, const float* hw, const float* gt, int P, const int* ngt, const int* best,
                        const unsigned char* status, const int* rgindex, const int* counts, const float* negloss, const int* sel_idx,
                        int sel_cap, const int* sel_cnt, float grad_scale, float* loss_parts, float* d_arm_loc, float* d_arm_conf,
                        float* d_odm_loc, float* d_odm_conf, void* stream);
/* RefineDet inference decode (RefineDet.py:189-206) for one image: keep[a] = softmax(arm)[1] < 0.99 and arg-max(softmax(odm)) != background;
 * conf [A][C-1] = softmax(odm) without the background column; boxes [A][4] y1x1y2x2 decoded through both stages; cand = keep and conf >= threshold
 * (the per-class NMS that follows is odtk_nms_batched, as for SSD300). */
int odtk_refinedet_decode(const float* arm_loc, const float* arm_conf, const float* odm_loc, const float* odm_conf, int A, int C,
                          const float* yx, const float* hw, float score_threshold, float* conf, float* boxes, unsigned char* keep,
                          unsigned char* cand, void* stream);

/* Focal (softmax flavour, alpha on positives AND negatives, p clipped to [1e-8, 1], sum / #positives;
 * RetinaNet.py:457-474) + smooth-L1 on the positive rows (:441-446).  pconf [N][A][C] logits (background =
 * class C-1), pbox [N][A][4] = (ty, tx, th, tw).  loss_parts[N][2] = {focal, coord}; dconf / dbox are
 * overwritten with d(sum_i (focal_i + coord_i) * grad_scale) / d(pconf, pbox)  (grad_scale = 1 / batch). */
int odtk_retina_loss(const float* pconf, const float* pbox, int N, int A, int C, const float* yx,
                     const float* hw, const float* gt, int P, const int* ngt, const int* best,
                     const unsigned char* status, const int* rgindex, const int* counts, float alpha,
                     float gamma, float grad_scale, float* loss_parts, float* dconf, float* dbox, void* stream);
/* RetinaNet inference branch up to the per-class NMS loop (RetinaNet.py:224-238): softmax, background-arg-max mask,
 * box decode; outputs as odtk_ssd_decode (conf [A][C-1], boxes [A][4] y1,x1,y2,x2, keep [A], cand [A][C-1]). */
int odtk_retina_decode(const float* pconf, const float* pbox, int A, int C, const float* yx, const float* hw,
                       float score_thr, float* conf, float* boxes, unsigned char* keep, unsigned char* cand,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * CenterNet box side: replaces CenterNet._compute_one_image_loss / _keypoints_loss /
 * _gaussian_radius (CenterNet.py:187-270) + the batch mean (:144-152) and the inference branch (:159-185).
 * keypoints [N][H][W][C] logits, offset / size [N][H][W][2], gt [N][P][5] = yc,xc,h,w,cls px (pad rows -1).
 * loss_parts [N][4] = keypoint, offset, size, total per image; d_* = grad_scale * d(sum_n total_n)/d(input)
 * (pass grad_scale = 1/N for the reference's batch mean).  workspace: odtk_centernet_workspace_bytes. */
long long odtk_centernet_workspace_bytes(int N, int H, int W, int C);
int odtk_centernet_loss(const float* keypoints, const float* offset, const float* size, const float* gt, int N, int H,
                        int W, int C, int P, float stride, float grad_scale, float* loss_parts, float* d_keypoints,
                        float* d_offset, float* d_size, void* workspace, void* stream);
/* one image: sigmoid, arg-max class, 3x3 peak test, score > threshold, top-k (descending, lower index first).
 * scores [top_k], bbox [top_k][4] y1,x1,y2,x2 px, class_id [top_k], count [1].  H*W <= 16384. */
int odtk_centernet_decode(const float* keypoints, const float* offset, const float* size, int H, int W, int C,
                          float stride, float score_threshold, int top_k, float* scores, float* bbox, int* class_id,
                          int* count, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FCOS box side: replaces the level assignment + FCOS._compute_one_image_loss
 * (FCOS.py:153-189, :266-348) and the decode candidates (FCOS.py:197-246; feed them to odtk_nms_batched).
 * conf / reg / center: host arrays of 5 device pointers (p3..p7) to [N][H_l][W_l][C] logits, [..][4] = l,r,t,b
 * distances (> 0, i.e. after the exp), [..][1] centre-ness logits; shapes [5][2] = H_l, W_l; strides 8..128.
 * loss [N] per image; d_* as for CenterNet.  workspace: odtk_fcos_workspace_bytes. */
long long odtk_fcos_workspace_bytes(const int* shapes, int N);
int odtk_fcos_loss(const float* const* conf, const float* const* reg, const float* const* center, const int* shapes,
                   const float* gt, int N, int C, int P, float grad_scale, float* loss, float* const* d_conf,
                   float* const* d_reg, float* const* d_center, void* workspace, void* stream);
int odtk_fcos_decode_candidates(const float* const* conf, const float* const* reg, const float* const* center,
                                const int* shapes, int C, float* pconf, float* pbbox, void* stream);

/* ---------------------------------------------------------------------------------------------
 * YOLOv3 box side: replaces the per-image loss loop (YOLOv3.py:117-310, batch mean :311) and the
 * decode candidates (:320-350; feed them to odtk_nms_batched).  pred: host array of 3 device pointers (head 1 =
 * coarsest) to [N][H_l][W_l][num_priors][C+5] = class(C), yx(2), hw(2), obj(1) logits; shapes [3][2];
 * priors [3][num_priors][2] (h, w) in the units the reference pairs with each head (config priors[i] / stride[i]);
 * head_stride [3] = 32, 16, 8 (ground truth / head_stride); decode_scale [3] = 32, 32, 16 (sic, YOLOv3.py:343-348).
 * loss_parts [N][5] = coord, class, obj, no-object sums and the per-image total; d_pred as for CenterNet. */
long long odtk_yolov3_workspace_bytes(const int* shapes, int num_priors, int N);
int odtk_yolov3_loss(const float* const* pred, const int* shapes, const float* priors, const float* head_stride,
                     const float* gt, int N, int num_priors, int C, int pad, float coord_scale, float noobj_scale,
                     float obj_scale, float class_scale, float grad_scale, float* loss_parts, float* const* d_pred,
                     void* workspace, void* stream);
int odtk_yolov3_decode_candidates(const float* const* pred, const int* shapes, const float* priors,
                                  const float* decode_scale, int num_priors, int C, float* confidence, float* bbox,
                                  void* stream);

/* YOLOv2 box side: the per-image loss loop of YOLOv2.py:102-166 (batch mean :167) and the decode of :177-186 (feed the candidates to
 * odtk_nms_batched).  pred [N][H][W][num_priors][C+5] = class(C), yx(2), hw(2), obj(1) logits (f32, device); priors: HOST array [num_priors][2] (h, w) in
 * cell units as the reference's config gives them; stride 32 (ground truth pixels / stride); gt [N][pad][5] = yc, xc, h, w, class padded with -1.
 * loss_parts [N][5] = coord (yx + hw), class, objectness, no-object sums and the scaled per-image total; d_pred (fully written) = d(sum_i total_i) *
 * grad_scale.  The reference's quirks (unclamped intersections, the mangled prior box of the no-object IoU, additive decode) are reproduced. */
int odtk_yolov2_loss(const float* pred, int N, int H, int W, int num_priors, int C, const float* priors, float stride, const float* gt,
                     int pad, float coord_scale, float noobj_scale, float obj_scale, float class_scale, float grad_scale,
                     float* loss_parts, float* d_pred, void* stream);
int odtk_yolov2_decode_candidates(const float* pred, int H, int W, int num_priors, int C, const float* priors, float stride,
                                  float* confidence, float* bbox, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image augmentor: replaces utils/image_augmentor.py:87-232 -- the tf.image / tf.contrib.image ops on
 * the picture and the box arithmetic next to them.  The host turns the reference's random draws (crop offsets, flip,
 * colour-jitter and rotate draws, in the reference's order) into one plan per image; the plans live in DEVICE memory.
 * Pixels: f32 or u8 source, HWC or CHW, any size per image; output f32 [N][out_h][out_w][C] (or [N][C][out_h][out_w]).
 * Boxes: gt_in [N][P][5] = ymin, ymax, xmin, xmax, class (:76-81) with gt_count[n] valid rows; gt_out [N][pad_to][5] =
 * yc, xc, h, w, class, rows whose centre left the picture dropped, the rest -1 (:201-230).  fallback[n] = 1 when every
 * box of image n was lost: boxes and picture then come from the plain resize of the input (gt_checker_helper :263-267).
 * Call odtk_augment_boxes first when there is ground truth and hand its fallback array to odtk_augment_images (NULL
 * without ground truth).  hue needs C == 3.  workspace: odtk_augment_workspace_bytes. */
typedef struct odtk_aug_plan {
    const void* src;          /* device pointer to the source picture */
    int src_u8;               /* 1: unsigned char pixels, 0: float */
    int src_chw;              /* 1: channels_first source */
    int in_h, in_w;
    int resize;               /* 0: fill_mode 'CONSTANT' (pad only, :119-123); 1 'BILINEAR', 2 'NEAREST_NEIGHBOR', 3 'BICUBIC' (:72-76:
                                 ResizeBilinear / ResizeNearestNeighbor / ResizeBicubic of TF 1.13, align_corners=True) */
    int resize_h, resize_w;   /* resize target inside the zoom canvas (:98-113, :125-128) */
    int crop_h, crop_w;       /* :131-143 */
    int flip_td, flip_lr;     /* :148-160 */
    int has_brightness, has_contrast, has_hue, has_rotate;
    float brightness, contrast, hue;   /* :173-188 */
    float angle;              /* radians, the image angle (rotate_helper :236); boxes turn by -angle */
    float ratio_y, ratio_x;   /* box zoom ratios (:114-116, :129-133) */
} odtk_aug_plan;
long long odtk_augment_workspace_bytes(int N, int C, int out_h, int out_w);
int odtk_augment_boxes(const odtk_aug_plan* plans, const float* gt_in, const int* gt_count, int N, int P, int out_h,
                       int out_w, int pad_to, float* gt_out, int* fallback, void* stream);
int odtk_augment_images(const odtk_aug_plan* plans, const int* fallback, int N, int C, int zoom_h, int zoom_w, int out_h,
                        int out_w, float constant_value, int out_chw, float* out, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Light-Head R-CNN (last entry): what LH_RCNN.py needs beyond the convolution / batch-norm / pool / NMS entry points above.
 * csrc/lhrcnn.hip; CPU restatement: oracle/lhrcnn_ref.py (pinned on the reference's own class).
 * --------------------------------------------------------------------------------------------- */
/* The depthwise half of tf.layers.separable_conv2d (LH_RCNN.py:551-567; depth multiplier 1, stride 1, SAME, odd kh x kw -- 3x3, 1x15, 15x1):
 * y[n,h,w,c] (+)= sum_{r,s} x[n, h + r - (kh-1)/2, w + s - (kw-1)/2, c] * filter[r][s][c]; filter is the f32 master [kh][kw][C]; x / y NHWC rows of
 * pitch ldx / ldy in `dtype`.  flip != 0 mirrors the taps: with dy as x this is the INPUT gradient of the same op.  The pointwise half is a 1x1
 * odtk_conv2d_*. */
int odtk_depthwise_conv(const void* x, int ldx, const float* filter, void* y, int ldy, int N, int H, int W, int C, int kh, int kw, int flip,
                        int accumulate, int dtype, void* stream);
/* dfilter[r][s][c] += sum_{n,h,w} x[n, h + r - (kh-1)/2, w + s - (kw-1)/2, c] * dy[n,h,w,c]   (float atomics over pixel slices; kh * kw <= 16) */
int odtk_depthwise_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dfilter, int N, int H, int W, int C, int kh, int kw, int dtype,
                         void* stream);
/* LHRCNN._compute_one_image_loss, first half (LH_RCNN.py:263-389), one workgroup per image.  Anchors: the A anchors that lie inside the picture
 * (:87-97) as [A][2] arrays, anchor_row[A] = their row in the prediction tensors conf [N][A_full][2] / bbox [N][A_full][4].  gt [N][P][5] = yc, xc, h, w,
 * class, padded with -1 (the number of boxes is the index of the first smallest yc, :265).  Per image: counts[8] = {G, n_pos, n_neg, k_pos = min(n_pos,
 * 128), k_neg = min(n_neg, 256 - k_pos)}; status[A] (3 best anchor of a box, 1 IoU > 0.5, 2 IoU < 0.3, 0 neither); the POSITIVE list in the reference's
 * order -- the G best anchors in box order (duplicates kept), then the status-1 anchors in anchor order -- as pos_anchor / pos_gt / pos_label / pos_score
 * (softmax of the two RPN outputs, column 0) / pos_box (y1,x1,y2,x2 of the anchor) / pos_valid; the NEGATIVE list (status 2, anchor order) with its
 * cross entropy against label 1 as score.  Label of a best row = label[anchor index] if that index < G else 0: tf.gather with the GPU kernel's
 * out-of-range rule (:337).  cap >= A + P is the lists' row pitch.  Feed both lists to odtk_nms_batched (IoU 0.7, max_out = counts[3] / counts[4]). */
int odtk_lhrcnn_match(const float* y1x1, const float* y2x2, const float* yx, const float* hw, const int* anchor_row, int A, int A_full,
                      const float* conf, const float* gt, int N, int P, int cap, int* counts, unsigned char* status, int* pos_anchor, int* pos_gt,
                      int* pos_label, float* pos_score, float* pos_box, unsigned char* pos_valid, int* neg_anchor, float* neg_score, float* neg_box,
                      unsigned char* neg_valid, void* stream);
/* Second half (:390-440) on the NMS picks sel_pos [N][128] / cnt_pos [N], sel_neg [N][256] / cnt_neg [N] (indices into the lists): loss_parts [N][4] =
 * {mean CE of the picked negatives, mean CE of the picked positives, 10 * mean smooth-L1 of their box codes, the sum}; d_conf / d_bbox = gradient of
 * grad_scale * sum_n total_n, fully written.  The R-CNN stage's inputs (:140-152) in fixed slots of 256 rows per image (positives first): roi_prop the
 * proposal clamped to [0, img_h - 1] x [0, img_w - 1], roi_box the same divided by (img_h - 1, img_w - 1, ..), roi_img (image, -1 = empty row), roi_label
 * (num_classes - 1 = background for negatives), roi_kind (1 | 2 | 0), roi_truth ((g_yx - p_yx) / p_yx -- sic, :430 -- and log(g_hw / p_hw)),
 * roi_counts [N][2]. */
int odtk_lhrcnn_rpn_loss(const float* y1x1, const float* y2x2, const float* yx, const float* hw, const int* anchor_row, int A, int A_full,
                         const float* conf, const float* bbox, const float* gt, int N, int P, int cap, int num_classes, const int* pos_anchor,
                         const int* pos_gt, const int* pos_label, const int* neg_anchor, const int* sel_pos, const int* cnt_pos, const int* sel_neg,
                         const int* cnt_neg, float grad_scale, int img_h, int img_w, float* loss_parts, float* d_conf, float* d_bbox, float* roi_box,
                         float* roi_prop, float* roi_truth, int* roi_img, int* roi_label, int* roi_kind, int* roi_counts, void* stream);
/* tf.image.crop_and_resize(feat, boxes, box_ind, [crop, crop]) (bilinear, extrapolation value 0; :146-149, :162) on NHWC rows: out row r =
 * [crop][crop][C] (pitch ldo, pad columns untouched); box_img[r] < 0 writes a zero row.  _bwd: the image gradient into d_feat (f32 [N*H*W][ldf], zeroed
 * by the call, float atomics). */
int odtk_crop_and_resize_fwd(const void* feat, int ldf, int N, int H, int W, int C, const float* boxes, const int* box_img, int R, int crop, void* out,
                             int ldo, int dtype, void* stream);
int odtk_crop_and_resize_bwd(const void* d_out, int ldo, int N, int H, int W, int C, const float* boxes, const int* box_img, int R, int crop, float* d_feat,
                             int ldf, int dtype, void* stream);
/* :167-170 on the 256-row slots of odtk_lhrcnn_rpn_loss: softmax cross entropy of every filled row (mean over ALL filled rows of the batch) and smooth L1
 * of the positive rows against roi_truth (mean over all positives); logits [N*256][ldl] (C classes), pbbox [N*256][ldb]; loss_parts [N][2] = this
 * image's share of the two means (sum over n = the loss); d_logits / d_pbbox fully written (scaled by grad_scale). */
int odtk_lhrcnn_rcnn_loss(const float* logits, int ldl, const float* pbbox, int ldb, int N, int C, const int* roi_label, const int* roi_kind,
                          const float* roi_truth, const int* roi_counts, float grad_scale, float* loss_parts, float* d_logits, float* d_pbbox, void* stream);
/* Test mode (:134-138, :153-164, :203-236) around odtk_nms_batched: proposals (decoded, clamped) and objectness of the kept anchors of image 0;
 * the picked proposals as crop rows (cap rows, roi_img -1 beyond cnt[0]); softmax / background filter / box decode of the head's outputs with
 * cand [R][C-1] = foreground row && confidence >= threshold. */
int odtk_lhrcnn_rpn_decode(const float* yx, const float* hw, const int* anchor_row, int A, int A_full, const float* conf, const float* bbox, int img_h,
                           int img_w, float* prop, float* score, void* stream);
int odtk_lhrcnn_gather_rois(const float* prop, const int* sel, const int* cnt, int cap, int img_h, int img_w, float* roi_box, float* roi_prop, int* roi_img,
                            void* stream);
int odtk_lhrcnn_rcnn_decode(const float* logits, int ldl, const float* pbbox, int ldb, const float* roi_prop, const int* roi_img, int R, int C,
                            float score_threshold, float* conf, float* boxes, unsigned char* cand, void* stream);

/* ------------------------------------------------------------------------- *
 * Collectives (SURVEY.md 8b's export list, 8e): the gradient sum of the data-parallel step for a binder that is not PyTorch.  No reference
 * counterpart (the reference is single-device, testSSD300.py:14).  A thin layer over RCCL (ring / tree all-reduce over xGMI), bound with dlopen at
 * first use: a missing RCCL fails these calls with a message and nothing else.  One communicator per process and GPU (the current HIP device at
 * odtk_comm_init); rank 0 creates the id, the host transports its ODTK_COMM_ID_BYTES bytes to the other ranks (file, socket, MPI, a
 * torch.distributed store: not the library's business); odtk_comm_init blocks until all `world` ranks have called it.  odtk_comm_allreduce sums
 * `count` elements of ODTK_F32 or ODTK_BF16 over the ranks, send == recv allowed, asynchronously on `stream` (the caller orders it behind the
 * kernels that produce `send` by launching on the same stream or by an event wait, exactly as for any kernel of this library).
 * ------------------------------------------------------------------------- */
#define ODTK_COMM_ID_BYTES 128
typedef struct odtk_comm odtk_comm;
int odtk_comm_unique_id(void* id128);
int odtk_comm_init(const void* id128, int rank, int world, odtk_comm** comm);
int odtk_comm_info(const odtk_comm* comm, int* rank, int* world);
int odtk_comm_allreduce(odtk_comm* comm, const void* send, void* recv, long long count, int dtype, void* stream);
int odtk_comm_broadcast(odtk_comm* comm, void* buf, long long count, int dtype, int root, void* stream);
int odtk_comm_destroy(odtk_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* ODTK_H_ */
