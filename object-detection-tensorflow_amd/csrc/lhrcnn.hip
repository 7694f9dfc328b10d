// Light-Head R-CNN on the GPU (gfx950): the device side of the reference's LH_RCNN.py that the other detector classes do not need.
//   depthwise_kernel / depthwise_wgrad_kernel .. the depthwise half of tf.layers.separable_conv2d (LH_RCNN.py:551-567; 3x3 in the backbone, 1x15 / 15x1 in the
//                                                light head), forward, input gradient (mirrored taps) and filter gradient
//   lh_match_kernel .......................... LH_RCNN._compute_one_image_loss up to the candidate lists (:263-389): best anchor per box, IoU > 0.5 / < 0.3 sets,
//                                                the ORDER of the reference's concatenations (the NMS that follows breaks ties by it)
//   lh_rpn_loss_kernel ....................... :390-440 after the two tf.image.non_max_suppression calls (odtk_nms_batched): losses, their gradients, the
//                                                proposals / labels / box targets handed to the R-CNN stage (:140-152)
//   crop_resize_fwd / bwd .................... tf.image.crop_and_resize(rcnn_feat, boxes, box_ind, [7, 7]) and its image gradient (:146-149, :162)
//   lh_rcnn_loss_kernel ...................... :167-170 softmax cross entropy over all rows + smooth L1 over the positive rows, with gradients
//   lh_rpn_decode / lh_gather_rois / lh_rcnn_decode .. the test-mode branch (:134-138, :153-164, :203-236) around odtk_nms_batched
// All of it is HBM / latency bound integer-and-float bookkeeping: no MFMA; one workgroup per image for the ordered lists (a wave ballot gives the order),
// one thread per output element with the channel innermost (coalesced NHWC rows) for the pixel kernels.  The dense layers and every 1x1 / 3x3
// convolution of the model run on the implicit-GEMM kernels of conv*.hip.
// Three behaviours of the reference are reproduced as TensorFlow executes them, not as they read (oracle/lhrcnn_ref.py header): both optimizer ops on
// every step (host side), the GPU gather semantics of :337 (label 0 for an out-of-range anchor index), the centre-divided box target of :430.
#include "common.h"
#include <math.h>

namespace odtk {
namespace {

constexpr int LH_THREADS = 256;
constexpr int LH_WAVES = LH_THREADS / 64;
constexpr int LH_MAX_GT = 128;
constexpr int LH_MAX_POS = 128, LH_ROIS = 256;         // LH_RCNN.py:383-384: at most 128 positives, 256 rows per image

// ------------------------------------------------------------------------------------------------------------------ depthwise convolution
template <typename T>
__global__ void __launch_bounds__(LH_THREADS) depthwise_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ f, T* __restrict__ y, int ldy,
                                                               int N, int H, int W, int C, int kh, int kw, int flip, int accumulate) {
    const long long total = (long long)N * H * W * C;
    const long long i = (long long)blockIdx.x * LH_THREADS + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long m = i / C;
    const int w = (int)(m % W), h = (int)((m / W) % H), n = (int)(m / ((long long)W * H));
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    float acc = 0.f;
    for (int r = 0; r < kh; ++r) {
        const int hh = h + r - ph;
        if (hh < 0 || hh >= H) continue;
        for (int s = 0; s < kw; ++s) {
            const int ww = w + s - pw;
            if (ww < 0 || ww >= W) continue;
            const int fr = flip ? kh - 1 - r : r, fs = flip ? kw - 1 - s : s;
            acc += elem<T>::load(x[(((long long)n * H + hh) * W + ww) * ldx + c]) * f[((long long)fr * kw + fs) * C + c];
        }
    }
    T* o = y + m * ldy + c;
    *o = elem<T>::store(accumulate ? elem<T>::load(*o) + acc : acc);
}

// df[r][s][c] += sum_m x[m + (r - ph, s - pw)][c] * dy[m][c]: a thread owns one channel and a slice of the pixels, up to 16 taps in registers
constexpr int DW_MAX_TAPS = 16, DW_CH = 64, DW_SLICES = LH_THREADS / DW_CH;
template <typename T>
__global__ void __launch_bounds__(LH_THREADS) depthwise_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, float* __restrict__ df,
                                                                     int N, int H, int W, int C, int kh, int kw, int pix_per_block) {
    __shared__ float red[DW_SLICES][DW_MAX_TAPS][DW_CH];
    const int c = blockIdx.x * DW_CH + (threadIdx.x % DW_CH), slice = threadIdx.x / DW_CH;
    const long long M = (long long)N * H * W;
    const long long m0 = (long long)blockIdx.y * pix_per_block, m1 = min(M, m0 + pix_per_block);
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2, taps = kh * kw;
    float acc[DW_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < DW_MAX_TAPS; ++t) acc[t] = 0.f;
    if (c < C) {
        for (long long m = m0 + slice; m < m1; m += DW_SLICES) {
            const float g = elem<T>::load(dy[m * lddy + c]);
            const int w = (int)(m % W), h = (int)((m / W) % H);
#pragma unroll
            for (int t = 0; t < DW_MAX_TAPS; ++t) {
                if (t < taps) {
                    const int hh = h + t / kw - ph, ww = w + t % kw - pw;
                    if (hh >= 0 && hh < H && ww >= 0 && ww < W) acc[t] += g * elem<T>::load(x[(m + (long long)(hh - h) * W + (ww - w)) * ldx + c]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < DW_MAX_TAPS; ++t) red[slice][t][threadIdx.x % DW_CH] = acc[t];
    __syncthreads();
    if (slice == 0 && c < C) {
        for (int t = 0; t < taps; ++t) {
            float s = 0.f;
            for (int k = 0; k < DW_SLICES; ++k) s += red[k][t][threadIdx.x];
            atomicAdd(df + (long long)t * C + c, s);
        }
    }
}

// Four channels per thread (16-byte f32 / 8-byte bf16 accesses, 32-bit index arithmetic): the shapes of the model (C = 144, 288, 576, 256) all divide by 4.
// First version (one element per thread, 64-bit div / mod per element): 8.8 ms of the 84.7-ms step at 700 x 1100 x 32 for each of the two kernels
// (profiles/r03zzzz_lhrcnn_700x1100_b32_kernel_trace.md).
template <typename T> struct vec4;
template <> struct vec4<float> {
    __device__ static float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct vec4<bf16_t> {
    __device__ static float4 load(const bf16_t* p) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(bf16_to_f32((bf16_t)(u.x & 0xffffu)), bf16_to_f32((bf16_t)(u.x >> 16)), bf16_to_f32((bf16_t)(u.y & 0xffffu)), bf16_to_f32((bf16_t)(u.y >> 16)));
    }
    __device__ static void store(bf16_t* p, float4 v) {
        uint2 u;
        u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2*>(p) = u;
    }
};

template <typename T>
__global__ void __launch_bounds__(LH_THREADS) depthwise4_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ f, T* __restrict__ y, int ldy,
                                                                int M, int H, int W, int C, int kh, int kw, int flip, int accumulate) {
    const int cq = C >> 2;                                     // channel quads per pixel
    const unsigned i = blockIdx.x * LH_THREADS + threadIdx.x;
    if (i >= (unsigned)M * (unsigned)cq) return;
    const int m = (int)(i / (unsigned)cq), c = ((int)(i - (unsigned)m * (unsigned)cq)) << 2;
    const int w = m % W, h = (m / W) % H;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < kh; ++r) {
        const int hh = h + r - ph;
        if (hh < 0 || hh >= H) continue;
        for (int s = 0; s < kw; ++s) {
            const int ww = w + s - pw;
            if (ww < 0 || ww >= W) continue;
            const int fr = flip ? kh - 1 - r : r, fs = flip ? kw - 1 - s : s;
            const float4 xv = vec4<T>::load(x + (size_t)(m + (hh - h) * W + (ww - w)) * ldx + c);
            const float4 fv = *reinterpret_cast<const float4*>(f + (size_t)(fr * kw + fs) * C + c);
            acc.x += xv.x * fv.x; acc.y += xv.y * fv.y; acc.z += xv.z * fv.z; acc.w += xv.w * fv.w;
        }
    }
    T* o = y + (size_t)m * ldy + c;
    if (accumulate) {
        const float4 old = vec4<T>::load(o);
        acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    vec4<T>::store(o, acc);
}

// filter gradient, four channels per thread: threads = (channel quad, pixel slice) with all 256 lanes in use for any C; the taps are reduced one at a
// time through 4 KiB of LDS (one float4 per thread), the first slice adds the sums into dfilter
template <typename T>
__global__ void __launch_bounds__(LH_THREADS) depthwise_wgrad4_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, float* __restrict__ df,
                                                                      int M, int H, int W, int C, int kh, int kw, int pix_per_block, int cq_per_block) {
    __shared__ float4 red[LH_THREADS];
    const int slices = LH_THREADS / cq_per_block;
    const int q = threadIdx.x % cq_per_block, slice = threadIdx.x / cq_per_block;
    const int c = (blockIdx.x * cq_per_block + q) << 2;
    const bool live = slice < slices && c < C;
    const int m0 = blockIdx.y * pix_per_block, m1 = min(M, m0 + pix_per_block);
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2, taps = kh * kw;
    float4 acc[DW_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < DW_MAX_TAPS; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int m = m0 + slice; m < m1; m += slices) {
            const float4 g = vec4<T>::load(dy + (size_t)m * lddy + c);
            const int w = m % W, h = (m / W) % H;
#pragma unroll
            for (int t = 0; t < DW_MAX_TAPS; ++t) {
                if (t < taps) {
                    const int dh = t / kw - ph, dw = t % kw - pw;
                    if (h + dh >= 0 && h + dh < H && w + dw >= 0 && w + dw < W) {
                        const float4 xv = vec4<T>::load(x + (size_t)(m + dh * W + dw) * ldx + c);
                        acc[t].x += g.x * xv.x; acc[t].y += g.y * xv.y; acc[t].z += g.z * xv.z; acc[t].w += g.w * xv.w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < DW_MAX_TAPS; ++t) {
        if (t < taps) {                                   // uniform
            red[threadIdx.x] = acc[t];
            __syncthreads();
            if (slice == 0 && c < C) {
                float4 s = red[q];
                for (int k = 1; k < slices; ++k) {
                    const float4 v = red[k * cq_per_block + q];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                float* o = df + (size_t)t * C + c;
                atomicAdd(o, s.x); atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ RPN: matching and candidate lists
struct LhAnchors {
    const float *y1x1, *y2x2, *yx, *hw;     // [A][2] each: the anchors INSIDE the picture (LH_RCNN.py:87-97), in anchor order
    const int* row;                          // [A]: row of the anchor in the prediction tensors [N][A_full][..]
    int A, A_full;
};

struct LhMatchArgs {
    LhAnchors an;
    const float *conf, *gt;                  // conf [N][A_full][2], gt [N][P][5]
    int P, cap;
    int* counts;                             // [N][8] = G, n_pos, n_neg, k_pos, k_neg
    unsigned char* status;                   // [N][A]: 3 best anchor of a box, 1 positive, 2 negative, 0 neither
    int *pos_anchor, *pos_gt, *pos_label, *neg_anchor;      // [N][cap]
    float *pos_score, *neg_score, *pos_box, *neg_box;       // [N][cap], [N][cap][4]
    unsigned char *pos_valid, *neg_valid;    // [N][cap]
};

__device__ __forceinline__ float lh_iou(const float* a1, const float* a2, const float* ahw, float gy1, float gx1, float gy2, float gx2, float garea) {
    const float ih = fmaxf(fminf(a2[0], gy2) - fmaxf(a1[0], gy1), 0.f), iw = fmaxf(fminf(a2[1], gx2) - fmaxf(a1[1], gx1), 0.f);
    const float inter = ih * iw;
    return inter / (ahw[0] * ahw[1] + garea - inter + 1e-8f);
}

__global__ void __launch_bounds__(LH_THREADS) lh_match_kernel(const LhMatchArgs a) {
    __shared__ float s_g[LH_MAX_GT][5];            // y1, x1, y2, x2, area
    __shared__ int s_label[LH_MAX_GT], s_best[LH_MAX_GT];
    __shared__ float s_rv[LH_WAVES];
    __shared__ int s_ri[LH_WAVES], s_cnt[LH_WAVES][2], s_G, s_base[2];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gt = a.gt + (size_t)n * a.P * 5;
    const int A = a.an.A;
    if (tid == 0) {                                // :265 argmin of the first column: the first row holding the smallest value
        int g = 0;
        float mn = gt[0];
        for (int i = 1; i < a.P; ++i) if (gt[i * 5] < mn) { mn = gt[i * 5]; g = i; }
        s_G = g;
    }
    __syncthreads();
    const int G = s_G;
    for (int g = tid; g < G; g += LH_THREADS) {
        const float yc = gt[g * 5], xc = gt[g * 5 + 1], h = gt[g * 5 + 2], w = gt[g * 5 + 3];
        s_g[g][0] = yc - h / 2.f; s_g[g][1] = xc - w / 2.f; s_g[g][2] = yc + h / 2.f; s_g[g][3] = xc + w / 2.f; s_g[g][4] = h * w;
        s_label[g] = (int)gt[g * 5 + 4];
    }
    unsigned char* status = a.status + (size_t)n * A;
    for (int i = tid; i < A; i += LH_THREADS) status[i] = 0;
    __syncthreads();
    // best anchor of every box: first maximum over the anchors (tf.argmax)
    for (int g = 0; g < G; ++g) {
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int i = tid; i < A; i += LH_THREADS) {
            const float v = lh_iou(a.an.y1x1 + 2 * i, a.an.y2x2 + 2 * i, a.an.hw + 2 * i, s_g[g][0], s_g[g][1], s_g[g][2], s_g[g][3], s_g[g][4]);
            if (v > bv) { bv = v; bi = i; }       // ascending i per thread: a later equal value does not replace
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_rv[wave] = bv; s_ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < LH_WAVES; ++k) if (s_rv[k] > bv || (s_rv[k] == bv && s_ri[k] < bi)) { bv = s_rv[k]; bi = s_ri[k]; }
            s_best[g] = bi;
            status[bi] = 3;
        }
        __syncthreads();
    }
    // the other anchors: positive above 0.5, negative below 0.3 (:352-354); status is this block's own global memory
    int* pos_anchor = a.pos_anchor + (size_t)n * a.cap; int* pos_gt = a.pos_gt + (size_t)n * a.cap; int* pos_label = a.pos_label + (size_t)n * a.cap;
    int* neg_anchor = a.neg_anchor + (size_t)n * a.cap;
    float* pos_score = a.pos_score + (size_t)n * a.cap; float* neg_score = a.neg_score + (size_t)n * a.cap;
    float* pos_box = a.pos_box + (size_t)n * a.cap * 4; float* neg_box = a.neg_box + (size_t)n * a.cap * 4;
    const float* conf = a.conf + (size_t)n * a.an.A_full * 2;
    auto emit = [&](int slot, int i, bool positive, int g, int label) {
        const float* yx = a.an.yx + 2 * i; const float* hw = a.an.hw + 2 * i;
        float* box = (positive ? pos_box : neg_box) + (size_t)slot * 4;
        box[0] = yx[0] - hw[0] / 2.f; box[1] = yx[1] - hw[1] / 2.f; box[2] = yx[0] + hw[0] / 2.f; box[3] = yx[1] + hw[1] / 2.f;     // :400, :408
        const float z0 = conf[(size_t)a.an.row[i] * 2], z1 = conf[(size_t)a.an.row[i] * 2 + 1];
        const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m);
        if (positive) {
            pos_anchor[slot] = i; pos_gt[slot] = g; pos_label[slot] = label;
            pos_score[slot] = e0 / (e0 + e1);                                   // tf.nn.softmax(pos_pconf)[:, 0]  (:387)
        } else {
            neg_anchor[slot] = i;
            neg_score[slot] = logf(e0 + e1) - (z1 - m);                          // cross entropy against label 1 (:391)
        }
    };
    // the G best rows lead the positive list, in box order, duplicates included (:356-363); their label is tf.gather(rcnn_label, best_raindex) (:337)
    for (int g = tid; g < G; g += LH_THREADS) {
        const int bi = s_best[g];
        emit(g, bi, true, g, bi < G ? s_label[bi] : 0);
    }
    if (tid == 0) { s_base[0] = G; s_base[1] = 0; }
    __syncthreads();
    for (int base = 0; base < A; base += LH_THREADS) {
        const int i = base + tid;
        int kind = 0, arg = 0;
        if (i < A && status[i] != 3) {
            float mv = -1.f;
            for (int g = 0; g < G; ++g) {
                const float v = lh_iou(a.an.y1x1 + 2 * i, a.an.y2x2 + 2 * i, a.an.hw + 2 * i, s_g[g][0], s_g[g][1], s_g[g][2], s_g[g][3], s_g[g][4]);
                if (v > mv) { mv = v; arg = g; }
            }
            kind = mv > 0.5f ? 1 : (mv < 0.3f ? 2 : 0);
            status[i] = (unsigned char)kind;
        }
        const unsigned long long mp = __ballot(kind == 1), mn = __ballot(kind == 2);
        if (lane == 0) { s_cnt[wave][0] = __popcll(mp); s_cnt[wave][1] = __popcll(mn); }
        __syncthreads();
        int op = s_base[0], on = s_base[1];
        for (int k = 0; k < wave; ++k) { op += s_cnt[k][0]; on += s_cnt[k][1]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (kind == 1) emit(op + __popcll(mp & below), i, true, arg, s_label[arg]);
        if (kind == 2) emit(on + __popcll(mn & below), i, false, 0, 0);
        __syncthreads();
        if (tid == 0) for (int k = 0; k < LH_WAVES; ++k) { s_base[0] += s_cnt[k][0]; s_base[1] += s_cnt[k][1]; }
        __syncthreads();
    }
    const int n_pos = s_base[0], n_neg = s_base[1];
    for (int j = tid; j < a.cap; j += LH_THREADS) {
        a.pos_valid[(size_t)n * a.cap + j] = j < n_pos;
        a.neg_valid[(size_t)n * a.cap + j] = j < n_neg;
    }
    if (tid == 0) {
        const int k_pos = min(n_pos, LH_MAX_POS), k_neg = min(n_neg, LH_ROIS - k_pos);
        int* c = a.counts + n * 8;
        c[0] = G; c[1] = n_pos; c[2] = n_neg; c[3] = k_pos; c[4] = k_neg; c[5] = c[6] = c[7] = 0;
    }
}

// ------------------------------------------------------------------------------------------------------------------ RPN: loss, gradients, proposals
struct LhLossArgs {
    LhAnchors an;
    const float *conf, *bbox, *gt;           // [N][A_full][2], [N][A_full][4], [N][P][5]
    int P, cap, num_classes;
    const int *pos_anchor, *pos_gt, *pos_label, *neg_anchor;
    const int *sel_pos, *cnt_pos, *sel_neg, *cnt_neg;      // NMS picks (indices into the lists): [N][128], [N], [N][256], [N]
    float grad_scale, img_h, img_w;          // img_h = H - 1, img_w = W - 1 (self.h, self.w of the reference)
    float *loss_parts, *d_conf, *d_bbox;     // [N][4] = mined-negative CE, positive CE, 10 * box term, total
    float *roi_box, *roi_prop, *roi_truth;   // [N*256][4]: normalised clamped box, the clamped box in pixels, R-CNN box target
    int *roi_img, *roi_label, *roi_kind, *roi_counts;      // [N*256] image (-1: empty row), class, 1 positive | 2 negative | 0 empty; [N][2]
};

__device__ __forceinline__ float lh_sl1(float d) { const float ad = fabsf(d); return ad < 1.f ? 0.5f * d * d : ad - 0.5f; }
__device__ __forceinline__ float lh_sl1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

__global__ void __launch_bounds__(LH_ROIS) lh_rpn_loss_kernel(const LhLossArgs a) {
    __shared__ float s_part[3][LH_ROIS];
    const int n = blockIdx.x, j = threadIdx.x;
    const int kp = min(a.cnt_pos[n], LH_MAX_POS), kn = min(a.cnt_neg[n], LH_ROIS - kp);
    const float* gt = a.gt + (size_t)n * a.P * 5;
    const size_t slot = (size_t)n * LH_ROIS + j;
    float ce_neg = 0.f, ce_pos = 0.f, coord = 0.f;
    int kind = 0;
    if (j < kp + kn) {
        const bool positive = j < kp;
        kind = positive ? 1 : 2;
        const int li = positive ? a.sel_pos[(size_t)n * LH_MAX_POS + j] : a.sel_neg[(size_t)n * LH_ROIS + (j - kp)];
        const int i = positive ? a.pos_anchor[(size_t)n * a.cap + li] : a.neg_anchor[(size_t)n * a.cap + li];
        const size_t row = (size_t)n * a.an.A_full + a.an.row[i];
        const float z0 = a.conf[row * 2], z1 = a.conf[row * 2 + 1];
        const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m), se = e0 + e1, lse = logf(se);
        const float p0 = e0 / se, p1 = e1 / se;
        const float* p = a.bbox + row * 4;
        const float ay = a.an.yx[2 * i], ax = a.an.yx[2 * i + 1], ah = a.an.hw[2 * i], aw = a.an.hw[2 * i + 1];
        const float pr_y = ah * p[0] + ay, pr_x = aw * p[1] + ax, pr_h = expf(p[2]) * ah, pr_w = expf(p[3]) * aw;     // :424-425, :432-433
        float box[4] = {pr_y - pr_h / 2.f, pr_x - pr_w / 2.f, pr_y + pr_h / 2.f, pr_x + pr_w / 2.f};
        const float lim[4] = {a.img_h, a.img_w, a.img_h, a.img_w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            box[k] = fminf(fmaxf(box[k], 0.f), lim[k]);                                 // :140-145
            a.roi_prop[slot * 4 + k] = box[k];
            a.roi_box[slot * 4 + k] = box[k] / lim[k];
        }
        a.roi_img[slot] = n;
        a.roi_kind[slot] = kind;
        if (positive) {
            const int g = a.pos_gt[(size_t)n * a.cap + li];
            const float gy = gt[g * 5], gx = gt[g * 5 + 1], gh = gt[g * 5 + 2], gw = gt[g * 5 + 3];
            ce_pos = lse - (z0 - m);
            const float w = a.grad_scale / (float)kp;
            a.d_conf[row * 2] = (p0 - 1.f) * w; a.d_conf[row * 2 + 1] = p1 * w;
            const float t[4] = {(gy - ay) / ah, (gx - ax) / aw, logf(gh / ah), logf(gw / aw)};            // :416-417
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = p[k] - t[k];
                coord += lh_sl1(d);
                a.d_bbox[row * 4 + k] = lh_sl1_grad(d) * 10.f * w;
            }
            a.roi_label[slot] = a.pos_label[(size_t)n * a.cap + li];
            a.roi_truth[slot * 4] = (gy - pr_y) / pr_y; a.roi_truth[slot * 4 + 1] = (gx - pr_x) / pr_x;     // :430 (sic: by the proposal's centre)
            a.roi_truth[slot * 4 + 2] = logf(gh / pr_h); a.roi_truth[slot * 4 + 3] = logf(gw / pr_w);
        } else {
            ce_neg = lse - (z1 - m);
            const float w = a.grad_scale / (float)kn;
            a.d_conf[row * 2] = p0 * w; a.d_conf[row * 2 + 1] = (p1 - 1.f) * w;
            a.roi_label[slot] = a.num_classes - 1;                                      // :150: background is the LAST class
#pragma unroll
            for (int k = 0; k < 4; ++k) a.roi_truth[slot * 4 + k] = 0.f;
        }
    } else {
        a.roi_img[slot] = -1; a.roi_kind[slot] = 0; a.roi_label[slot] = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a.roi_box[slot * 4 + k] = 0.f; a.roi_prop[slot * 4 + k] = 0.f; a.roi_truth[slot * 4 + k] = 0.f; }
    }
    s_part[0][j] = ce_neg; s_part[1][j] = ce_pos; s_part[2][j] = coord;
    __syncthreads();
    if (j == 0) {                              // fixed order: the same sum on every run
        float sn = 0.f, sp = 0.f, sc = 0.f;
        for (int k = 0; k < kp; ++k) { sp += s_part[1][k]; sc += s_part[2][k]; }
        for (int k = kp; k < kp + kn; ++k) sn += s_part[0][k];
        float* lp = a.loss_parts + n * 4;
        lp[0] = sn / (float)kn; lp[1] = sp / (float)kp; lp[2] = 10.f * sc / (float)kp;   // tf.reduce_mean of an empty list is NaN here as there
        lp[3] = lp[0] + lp[1] + lp[2];
        a.roi_counts[n * 2] = kp; a.roi_counts[n * 2 + 1] = kn;
    }
}

// ------------------------------------------------------------------------------------------------------------------ crop_and_resize (bilinear, extrapolation 0)
struct CropGeom {
    float in_y, in_x;
    bool ok;
};
__device__ __forceinline__ CropGeom crop_geom(const float* box, int H, int W, int crop, int by, int bx) {
    const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
    CropGeom g;
    const float hs = crop > 1 ? (y2 - y1) * (float)(H - 1) / (float)(crop - 1) : 0.f, ws = crop > 1 ? (x2 - x1) * (float)(W - 1) / (float)(crop - 1) : 0.f;
    g.in_y = crop > 1 ? y1 * (float)(H - 1) + (float)by * hs : 0.5f * (y1 + y2) * (float)(H - 1);
    g.in_x = crop > 1 ? x1 * (float)(W - 1) + (float)bx * ws : 0.5f * (x1 + x2) * (float)(W - 1);
    g.ok = g.in_y >= 0.f && g.in_y <= (float)(H - 1) && g.in_x >= 0.f && g.in_x <= (float)(W - 1);
    return g;
}

template <typename T>
__global__ void __launch_bounds__(LH_THREADS) crop_resize_fwd_kernel(const T* __restrict__ feat, int ldf, int H, int W, int C, const float* __restrict__ boxes,
                                                                     const int* __restrict__ box_img, int crop, T* __restrict__ out, int ldo) {
    const int r = blockIdx.x;
    const int img = box_img[r];
    T* o = out + (size_t)r * ldo;
    const int total = crop * crop * C;
    for (int e = threadIdx.x; e < total; e += LH_THREADS) {
        float v = 0.f;
        if (img >= 0) {
            const int c = e % C, bin = e / C, bx = bin % crop, by = bin / crop;
            const CropGeom g = crop_geom(boxes + (size_t)r * 4, H, W, crop, by, bx);
            if (g.ok) {
                const int y0 = (int)floorf(g.in_y), y1 = (int)ceilf(g.in_y), x0 = (int)floorf(g.in_x), x1 = (int)ceilf(g.in_x);
                const float ly = g.in_y - (float)y0, lx = g.in_x - (float)x0;
                const T* base = feat + (size_t)img * H * W * ldf + c;
                const float tl = elem<T>::load(base[((size_t)y0 * W + x0) * ldf]), tr = elem<T>::load(base[((size_t)y0 * W + x1) * ldf]);
                const float bl = elem<T>::load(base[((size_t)y1 * W + x0) * ldf]), br = elem<T>::load(base[((size_t)y1 * W + x1) * ldf]);
                const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
                v = top + (bot - top) * ly;
            }
        }
        o[e] = elem<T>::store(v);
    }
}

template <typename T>
__global__ void __launch_bounds__(LH_THREADS) crop_resize_bwd_kernel(const T* __restrict__ d_out, int ldo, int H, int W, int C, const float* __restrict__ boxes,
                                                                     const int* __restrict__ box_img, int crop, float* __restrict__ d_feat, int ldf) {
    const int r = blockIdx.x;
    const int img = box_img[r];
    if (img < 0) return;
    const T* go = d_out + (size_t)r * ldo;
    const int total = crop * crop * C;
    for (int e = threadIdx.x; e < total; e += LH_THREADS) {
        const int c = e % C, bin = e / C, bx = bin % crop, by = bin / crop;
        const CropGeom g = crop_geom(boxes + (size_t)r * 4, H, W, crop, by, bx);
        if (!g.ok) continue;
        const float dv = elem<T>::load(go[e]);
        const int y0 = (int)floorf(g.in_y), y1 = (int)ceilf(g.in_y), x0 = (int)floorf(g.in_x), x1 = (int)ceilf(g.in_x);
        const float ly = g.in_y - (float)y0, lx = g.in_x - (float)x0;
        float* base = d_feat + (size_t)img * H * W * ldf + c;
        atomicAdd(base + ((size_t)y0 * W + x0) * ldf, dv * (1.f - ly) * (1.f - lx));
        atomicAdd(base + ((size_t)y0 * W + x1) * ldf, dv * (1.f - ly) * lx);
        atomicAdd(base + ((size_t)y1 * W + x0) * ldf, dv * ly * (1.f - lx));
        atomicAdd(base + ((size_t)y1 * W + x1) * ldf, dv * ly * lx);
    }
}

// ------------------------------------------------------------------------------------------------------------------ R-CNN loss (one workgroup per image slot)
struct LhRcnnArgs {
    const float *logits, *pbbox;             // [R][ldl], [R][ldb] f32
    int ldl, ldb, C, N;
    const int *roi_label, *roi_kind, *roi_counts;
    const float* roi_truth;
    float grad_scale;
    float *loss_parts, *d_logits, *d_pbbox;  // [N][2] = sum of the rows' cross entropies / all rows of the batch, sum of the positive rows' box terms / all positives
};

__global__ void __launch_bounds__(LH_ROIS) lh_rcnn_loss_kernel(const LhRcnnArgs a) {
    __shared__ float s_ce[LH_ROIS], s_box[LH_ROIS];
    __shared__ int s_tot[2];
    const int n = blockIdx.x, j = threadIdx.x;
    if (j == 0) {
        int rows = 0, pos = 0;
        for (int k = 0; k < a.N; ++k) { pos += a.roi_counts[k * 2]; rows += a.roi_counts[k * 2] + a.roi_counts[k * 2 + 1]; }
        s_tot[0] = rows; s_tot[1] = pos;
    }
    __syncthreads();
    const size_t r = (size_t)n * LH_ROIS + j;
    const int kind = a.roi_kind[r];
    float ce = 0.f, bx = 0.f;
    const float* z = a.logits + r * a.ldl;
    float* dz = a.d_logits + r * a.ldl;
    float* db = a.d_pbbox + r * a.ldb;
    if (kind != 0) {
        const int label = a.roi_label[r];
        float m = z[0];
        for (int c = 1; c < a.C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < a.C; ++c) se += expf(z[c] - m);
        ce = logf(se) - (z[label] - m);
        const float w = a.grad_scale / (float)s_tot[0];
        for (int c = 0; c < a.C; ++c) dz[c] = (expf(z[c] - m) / se - (c == label ? 1.f : 0.f)) * w;
    } else {
        for (int c = 0; c < a.C; ++c) dz[c] = 0.f;
    }
    if (kind == 1) {
        const float w = a.grad_scale / (float)s_tot[1];
        for (int k = 0; k < 4; ++k) {
            const float d = a.pbbox[r * a.ldb + k] - a.roi_truth[r * 4 + k];
            bx += lh_sl1(d);
            db[k] = lh_sl1_grad(d) * w;
        }
    } else {
        for (int k = 0; k < 4; ++k) db[k] = 0.f;
    }
    for (int c = a.C; c < a.ldl; ++c) dz[c] = 0.f;
    for (int k = 4; k < a.ldb; ++k) db[k] = 0.f;
    s_ce[j] = ce; s_box[j] = bx;
    __syncthreads();
    if (j == 0) {
        float sc = 0.f, sb = 0.f;
        for (int k = 0; k < LH_ROIS; ++k) { sc += s_ce[k]; sb += s_box[k]; }
        a.loss_parts[n * 2] = sc / (float)s_tot[0];
        a.loss_parts[n * 2 + 1] = sb / (float)s_tot[1];
    }
}

// ------------------------------------------------------------------------------------------------------------------ inference
__global__ void __launch_bounds__(LH_THREADS) lh_rpn_decode_kernel(const LhAnchors an, const float* __restrict__ conf, const float* __restrict__ bbox, float img_h,
                                                                   float img_w, float* __restrict__ prop, float* __restrict__ score) {
    const int i = blockIdx.x * LH_THREADS + threadIdx.x;
    if (i >= an.A) return;
    const size_t row = an.row[i];
    const float* p = bbox + row * 4;
    const float ah = an.hw[2 * i], aw = an.hw[2 * i + 1];
    const float y = p[0] * ah + an.yx[2 * i], x = p[1] * aw + an.yx[2 * i + 1], h = expf(p[2]) * ah, w = expf(p[3]) * aw;     // :134-136
    prop[i * 4] = fminf(fmaxf(y - h / 2.f, 0.f), img_h); prop[i * 4 + 1] = fminf(fmaxf(x - w / 2.f, 0.f), img_w);
    prop[i * 4 + 2] = fminf(fmaxf(y + h / 2.f, 0.f), img_h); prop[i * 4 + 3] = fminf(fmaxf(x + w / 2.f, 0.f), img_w);
    const float z0 = conf[row * 2], z1 = conf[row * 2 + 1], m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m);
    score[i] = e0 / (e0 + e1);
}

__global__ void __launch_bounds__(LH_THREADS) lh_gather_rois_kernel(const float* __restrict__ prop, const int* __restrict__ sel, const int* __restrict__ cnt, int cap,
                                                                    float img_h, float img_w, float* __restrict__ roi_box, float* __restrict__ roi_prop,
                                                                    int* __restrict__ roi_img) {
    const int j = blockIdx.x * LH_THREADS + threadIdx.x;
    if (j >= cap) return;
    const bool live = j < min(cnt[0], cap);
    const float lim[4] = {img_h, img_w, img_h, img_w};
    for (int k = 0; k < 4; ++k) {
        const float v = live ? prop[(size_t)sel[j] * 4 + k] : 0.f;
        roi_prop[j * 4 + k] = v;
        roi_box[j * 4 + k] = v / lim[k];
    }
    roi_img[j] = live ? 0 : -1;
}

// :203-219: softmax, rows whose arg-max is the background are dropped, boxes decoded about the (clamped) proposal; cand = kept && confidence >= threshold
__global__ void __launch_bounds__(LH_THREADS) lh_rcnn_decode_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ pbbox, int ldb,
                                                                    const float* __restrict__ roi_prop, const int* __restrict__ roi_img, int R, int C, float thr,
                                                                    float* __restrict__ conf, float* __restrict__ boxes, unsigned char* __restrict__ cand) {
    const int r = blockIdx.x * LH_THREADS + threadIdx.x;
    if (r >= R) return;
    const int nc = C - 1;
    if (roi_img[r] < 0) {
        for (int c = 0; c < nc; ++c) { conf[(size_t)r * nc + c] = 0.f; cand[(size_t)r * nc + c] = 0; }
        for (int k = 0; k < 4; ++k) boxes[r * 4 + k] = 0.f;
        return;
    }
    const float* z = logits + (size_t)r * ldl;
    float m = z[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) if (z[c] > m) { m = z[c]; arg = c; }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - m);
    const bool fg = arg < nc;                       // arg-max of the softmax = arg-max of the logits (first maximum)
    for (int c = 0; c < nc; ++c) {
        const float p = expf(z[c] - m) / se;
        conf[(size_t)r * nc + c] = p;
        cand[(size_t)r * nc + c] = fg && p >= thr;
    }
    const float* q = roi_prop + (size_t)r * 4;
    const float py = q[0] / 2.f + q[2] / 2.f, px = q[1] / 2.f + q[3] / 2.f, ph = q[2] - q[0], pw = q[3] - q[1];     // :158-159
    const float* t = pbbox + (size_t)r * ldb;
    const float y = t[0] * ph + py, x = t[1] * pw + px, h = ph * expf(t[2]), w = pw * expf(t[3]);
    boxes[r * 4] = y - h / 2.f; boxes[r * 4 + 1] = x - w / 2.f; boxes[r * 4 + 2] = y + h / 2.f; boxes[r * 4 + 3] = x + w / 2.f;
}

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" int odtk_depthwise_conv(const void* x, int ldx, const float* filter, void* y, int ldy, int N, int H, int W, int C, int kh, int kw, int flip,
                                   int accumulate, int dtype, void* stream) {
    ODTK_REQUIRE(x && filter && y, "depthwise_conv: null pointer");
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C, "depthwise_conv: N=%d H=%d W=%d C=%d ldx=%d ldy=%d out of range", N, H, W, C, ldx, ldy);
    ODTK_REQUIRE(kh > 0 && kw > 0 && (kh & 1) && (kw & 1), "depthwise_conv: %dx%d taps (odd sizes: SAME padding is symmetric)", kh, kw);
    const long long total = (long long)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    const int vchunk = dtype == ODTK_F32 ? 4 : 4;
    if ((C & 3) == 0 && ldx % vchunk == 0 && ldy % vchunk == 0 && total / 4 < (1ll << 31) && (long long)N * H * W < (1ll << 30)) {     // four channels per thread
        const long long quads = total / 4;
        const dim3 g4((unsigned)((quads + LH_THREADS - 1) / LH_THREADS));
        if (dtype == ODTK_F32)
            hipLaunchKernelGGL(depthwise4_kernel<float>, g4, dim3(LH_THREADS), 0, st, (const float*)x, ldx, filter, (float*)y, ldy, N * H * W, H, W, C, kh, kw, flip, accumulate);
        else
            hipLaunchKernelGGL(depthwise4_kernel<bf16_t>, g4, dim3(LH_THREADS), 0, st, (const bf16_t*)x, ldx, filter, (bf16_t*)y, ldy, N * H * W, H, W, C, kh, kw, flip, accumulate);
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    const dim3 grid((unsigned)((total + LH_THREADS - 1) / LH_THREADS));
    if (dtype == ODTK_F32)
        hipLaunchKernelGGL(depthwise_kernel<float>, grid, dim3(LH_THREADS), 0, st, (const float*)x, ldx, filter, (float*)y, ldy, N, H, W, C, kh, kw, flip, accumulate);
    else
        hipLaunchKernelGGL(depthwise_kernel<bf16_t>, grid, dim3(LH_THREADS), 0, st, (const bf16_t*)x, ldx, filter, (bf16_t*)y, ldy, N, H, W, C, kh, kw, flip, accumulate);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_depthwise_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dfilter, int N, int H, int W, int C, int kh, int kw, int dtype,
                                    void* stream) {
    ODTK_REQUIRE(x && dy && dfilter, "depthwise_wgrad: null pointer");
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && lddy >= C, "depthwise_wgrad: N=%d H=%d W=%d C=%d out of range", N, H, W, C);
    ODTK_REQUIRE(kh > 0 && kw > 0 && (kh & 1) && (kw & 1) && kh * kw <= DW_MAX_TAPS, "depthwise_wgrad: %dx%d taps (odd sizes, at most %d taps)", kh, kw, DW_MAX_TAPS);
    const long long M = (long long)N * H * W;
    if ((C & 3) == 0 && ldx % 4 == 0 && lddy % 4 == 0 && M < (1ll << 30)) {                       // four channels per thread, every lane busy
        const int cq = C >> 2;
        const int cq_per_block = cq >= LH_THREADS ? LH_THREADS : cq;                                // C <= 1024: one block column owns all channels of its pixels
        const int cblocks = ceil_div(cq, cq_per_block);
        const int slices = LH_THREADS / cq_per_block;
        int splits = (int)min((long long)max(1, 2048 / cblocks), (M + 64 * slices - 1) / (64 * slices));
        const int per = (int)((M + splits - 1) / splits);
        splits = (int)((M + per - 1) / per);
        hipStream_t st4 = (hipStream_t)stream;
        if (dtype == ODTK_F32)
            hipLaunchKernelGGL(depthwise_wgrad4_kernel<float>, dim3(cblocks, splits), dim3(LH_THREADS), 0, st4, (const float*)x, ldx, (const float*)dy, lddy, dfilter, (int)M, H, W,
                               C, kh, kw, per, cq_per_block);
        else
            hipLaunchKernelGGL(depthwise_wgrad4_kernel<bf16_t>, dim3(cblocks, splits), dim3(LH_THREADS), 0, st4, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, dfilter, (int)M, H,
                               W, C, kh, kw, per, cq_per_block);
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    const int ctiles = ceil_div(C, DW_CH);
    int splits = (int)min((long long)max(1, 2048 / ctiles), (M + 255) / 256);
    const int per = (int)((M + splits - 1) / splits);
    splits = (int)((M + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_F32)
        hipLaunchKernelGGL(depthwise_wgrad_kernel<float>, dim3(ctiles, splits), dim3(LH_THREADS), 0, st, (const float*)x, ldx, (const float*)dy, lddy, dfilter, N, H, W, C,
                           kh, kw, per);
    else
        hipLaunchKernelGGL(depthwise_wgrad_kernel<bf16_t>, dim3(ctiles, splits), dim3(LH_THREADS), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, dfilter, N, H, W,
                           C, kh, kw, per);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

static LhAnchors lh_anchors(const float* y1x1, const float* y2x2, const float* yx, const float* hw, const int* row, int A, int A_full) {
    LhAnchors an;
    an.y1x1 = y1x1; an.y2x2 = y2x2; an.yx = yx; an.hw = hw; an.row = row; an.A = A; an.A_full = A_full;
    return an;
}

extern "C" int odtk_lhrcnn_match(const float* y1x1, const float* y2x2, const float* yx, const float* hw, const int* anchor_row, int A, int A_full,
                                 const float* conf, const float* gt, int N, int P, int cap, int* counts, unsigned char* status, int* pos_anchor, int* pos_gt,
                                 int* pos_label, float* pos_score, float* pos_box, unsigned char* pos_valid, int* neg_anchor, float* neg_score, float* neg_box,
                                 unsigned char* neg_valid, void* stream) {
    ODTK_REQUIRE(y1x1 && y2x2 && yx && hw && anchor_row && conf && gt && counts && status && pos_anchor && pos_gt && pos_label && pos_score && pos_box && pos_valid &&
                 neg_anchor && neg_score && neg_box && neg_valid, "lhrcnn_match: null pointer");
    ODTK_REQUIRE(A > 0 && A <= A_full && N > 0 && P > 0 && P <= LH_MAX_GT && cap >= A + P, "lhrcnn_match: A=%d A_full=%d N=%d P=%d cap=%d out of range", A, A_full, N, P, cap);
    LhMatchArgs a;
    a.an = lh_anchors(y1x1, y2x2, yx, hw, anchor_row, A, A_full);
    a.conf = conf; a.gt = gt; a.P = P; a.cap = cap; a.counts = counts; a.status = status;
    a.pos_anchor = pos_anchor; a.pos_gt = pos_gt; a.pos_label = pos_label; a.neg_anchor = neg_anchor;
    a.pos_score = pos_score; a.neg_score = neg_score; a.pos_box = pos_box; a.neg_box = neg_box; a.pos_valid = pos_valid; a.neg_valid = neg_valid;
    hipLaunchKernelGGL(lh_match_kernel, dim3(N), dim3(LH_THREADS), 0, (hipStream_t)stream, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_lhrcnn_rpn_loss(const float* y1x1, const float* y2x2, const float* yx, const float* hw, const int* anchor_row, int A, int A_full,
                                    const float* conf, const float* bbox, const float* gt, int N, int P, int cap, int num_classes, const int* pos_anchor,
                                    const int* pos_gt, const int* pos_label, const int* neg_anchor, const int* sel_pos, const int* cnt_pos, const int* sel_neg,
                                    const int* cnt_neg, float grad_scale, int img_h, int img_w, float* loss_parts, float* d_conf, float* d_bbox, float* roi_box,
                                    float* roi_prop, float* roi_truth, int* roi_img, int* roi_label, int* roi_kind, int* roi_counts, void* stream) {
    ODTK_REQUIRE(yx && hw && anchor_row && conf && bbox && gt && pos_anchor && pos_gt && pos_label && neg_anchor && sel_pos && cnt_pos && sel_neg && cnt_neg &&
                 loss_parts && d_conf && d_bbox && roi_box && roi_prop && roi_truth && roi_img && roi_label && roi_kind && roi_counts, "lhrcnn_rpn_loss: null pointer");
    ODTK_REQUIRE(A > 0 && A <= A_full && N > 0 && P > 0 && P <= LH_MAX_GT && num_classes > 1 && img_h > 1 && img_w > 1, "lhrcnn_rpn_loss: argument out of range");
    hipStream_t st = (hipStream_t)stream;
    if (int e = zero_async(d_conf, (size_t)N * A_full * 2 * sizeof(float), st)) return e;
    if (int e = zero_async(d_bbox, (size_t)N * A_full * 4 * sizeof(float), st)) return e;
    LhLossArgs a;
    a.an = lh_anchors(y1x1, y2x2, yx, hw, anchor_row, A, A_full);
    a.conf = conf; a.bbox = bbox; a.gt = gt; a.P = P; a.cap = cap; a.num_classes = num_classes;
    a.pos_anchor = pos_anchor; a.pos_gt = pos_gt; a.pos_label = pos_label; a.neg_anchor = neg_anchor;
    a.sel_pos = sel_pos; a.cnt_pos = cnt_pos; a.sel_neg = sel_neg; a.cnt_neg = cnt_neg;
    a.grad_scale = grad_scale; a.img_h = (float)(img_h - 1); a.img_w = (float)(img_w - 1);
    a.loss_parts = loss_parts; a.d_conf = d_conf; a.d_bbox = d_bbox; a.roi_box = roi_box; a.roi_prop = roi_prop; a.roi_truth = roi_truth;
    a.roi_img = roi_img; a.roi_label = roi_label; a.roi_kind = roi_kind; a.roi_counts = roi_counts;
    hipLaunchKernelGGL(lh_rpn_loss_kernel, dim3(N), dim3(LH_ROIS), 0, st, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_crop_and_resize_fwd(const void* feat, int ldf, int N, int H, int W, int C, const float* boxes, const int* box_img, int R, int crop, void* out,
                                        int ldo, int dtype, void* stream) {
    ODTK_REQUIRE(feat && boxes && box_img && out, "crop_and_resize_fwd: null pointer");
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && ldf >= C && R > 0 && crop > 0 && ldo >= crop * crop * C, "crop_and_resize_fwd: argument out of range (C=%d crop=%d ldo=%d)", C, crop, ldo);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_F32)
        hipLaunchKernelGGL(crop_resize_fwd_kernel<float>, dim3(R), dim3(LH_THREADS), 0, st, (const float*)feat, ldf, H, W, C, boxes, box_img, crop, (float*)out, ldo);
    else
        hipLaunchKernelGGL(crop_resize_fwd_kernel<bf16_t>, dim3(R), dim3(LH_THREADS), 0, st, (const bf16_t*)feat, ldf, H, W, C, boxes, box_img, crop, (bf16_t*)out, ldo);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_crop_and_resize_bwd(const void* d_out, int ldo, int N, int H, int W, int C, const float* boxes, const int* box_img, int R, int crop, float* d_feat,
                                        int ldf, int dtype, void* stream) {
    ODTK_REQUIRE(d_out && boxes && box_img && d_feat, "crop_and_resize_bwd: null pointer");
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && ldf >= C && R > 0 && crop > 0 && ldo >= crop * crop * C, "crop_and_resize_bwd: argument out of range");
    hipStream_t st = (hipStream_t)stream;
    if (int e = zero_async(d_feat, (size_t)N * H * W * ldf * sizeof(float), st)) return e;
    if (dtype == ODTK_F32)
        hipLaunchKernelGGL(crop_resize_bwd_kernel<float>, dim3(R), dim3(LH_THREADS), 0, st, (const float*)d_out, ldo, H, W, C, boxes, box_img, crop, d_feat, ldf);
    else
        hipLaunchKernelGGL(crop_resize_bwd_kernel<bf16_t>, dim3(R), dim3(LH_THREADS), 0, st, (const bf16_t*)d_out, ldo, H, W, C, boxes, box_img, crop, d_feat, ldf);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_lhrcnn_rcnn_loss(const float* logits, int ldl, const float* pbbox, int ldb, int N, int C, const int* roi_label, const int* roi_kind,
                                     const float* roi_truth, const int* roi_counts, float grad_scale, float* loss_parts, float* d_logits, float* d_pbbox, void* stream) {
    ODTK_REQUIRE(logits && pbbox && roi_label && roi_kind && roi_truth && roi_counts && loss_parts && d_logits && d_pbbox, "lhrcnn_rcnn_loss: null pointer");
    ODTK_REQUIRE(N > 0 && C > 1 && ldl >= C && ldb >= 4, "lhrcnn_rcnn_loss: N=%d C=%d ldl=%d ldb=%d out of range", N, C, ldl, ldb);
    LhRcnnArgs a;
    a.logits = logits; a.pbbox = pbbox; a.ldl = ldl; a.ldb = ldb; a.C = C; a.N = N; a.roi_label = roi_label; a.roi_kind = roi_kind; a.roi_counts = roi_counts;
    a.roi_truth = roi_truth; a.grad_scale = grad_scale; a.loss_parts = loss_parts; a.d_logits = d_logits; a.d_pbbox = d_pbbox;
    hipLaunchKernelGGL(lh_rcnn_loss_kernel, dim3(N), dim3(LH_ROIS), 0, (hipStream_t)stream, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_lhrcnn_rpn_decode(const float* yx, const float* hw, const int* anchor_row, int A, int A_full, const float* conf, const float* bbox, int img_h,
                                      int img_w, float* prop, float* score, void* stream) {
    ODTK_REQUIRE(yx && hw && anchor_row && conf && bbox && prop && score, "lhrcnn_rpn_decode: null pointer");
    ODTK_REQUIRE(A > 0 && A <= A_full && img_h > 1 && img_w > 1, "lhrcnn_rpn_decode: A=%d A_full=%d out of range", A, A_full);
    const LhAnchors an = lh_anchors(nullptr, nullptr, yx, hw, anchor_row, A, A_full);
    hipLaunchKernelGGL(lh_rpn_decode_kernel, dim3(ceil_div(A, LH_THREADS)), dim3(LH_THREADS), 0, (hipStream_t)stream, an, conf, bbox, (float)(img_h - 1), (float)(img_w - 1),
                       prop, score);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_lhrcnn_gather_rois(const float* prop, const int* sel, const int* cnt, int cap, int img_h, int img_w, float* roi_box, float* roi_prop, int* roi_img,
                                       void* stream) {
    ODTK_REQUIRE(prop && sel && cnt && roi_box && roi_prop && roi_img, "lhrcnn_gather_rois: null pointer");
    ODTK_REQUIRE(cap > 0 && img_h > 1 && img_w > 1, "lhrcnn_gather_rois: cap=%d out of range", cap);
    hipLaunchKernelGGL(lh_gather_rois_kernel, dim3(ceil_div(cap, LH_THREADS)), dim3(LH_THREADS), 0, (hipStream_t)stream, prop, sel, cnt, cap, (float)(img_h - 1),
                       (float)(img_w - 1), roi_box, roi_prop, roi_img);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_lhrcnn_rcnn_decode(const float* logits, int ldl, const float* pbbox, int ldb, const float* roi_prop, const int* roi_img, int R, int C,
                                       float score_threshold, float* conf, float* boxes, unsigned char* cand, void* stream) {
    ODTK_REQUIRE(logits && pbbox && roi_prop && roi_img && conf && boxes && cand, "lhrcnn_rcnn_decode: null pointer");
    ODTK_REQUIRE(R > 0 && C > 1 && ldl >= C && ldb >= 4, "lhrcnn_rcnn_decode: R=%d C=%d ldl=%d ldb=%d out of range", R, C, ldl, ldb);
    hipLaunchKernelGGL(lh_rcnn_decode_kernel, dim3(ceil_div(R, LH_THREADS)), dim3(LH_THREADS), 0, (hipStream_t)stream, logits, ldl, pbbox, ldb, roi_prop, roi_img, R, C,
                       score_threshold, conf, boxes, cand);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
