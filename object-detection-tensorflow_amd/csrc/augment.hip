// Image augmentor on the GPU (gfx950): the step in front of the detector's hot path, SURVEY.md 8(f).2.
// Replaces utils/image_augmentor.py:87-232 of the reference (resize / pad / crop / flips / colour jitter / rotate
// of the image, the same transforms on the boxes, centre filter, [yc,xc,h,w,cls] rows padded with -1).  The host
// (odtk/augment.py) turns the random draws into one odtk_aug_plan per image; everything per pixel / per box is here.
//
// HBM-bound gather work, no MFMA: one thread per output pixel, all channels; the whole batch in one launch per stage
// (grid.y = image), sources of different sizes addressed through the plan.  Stages:
//   geometry  : resize (bilinear | nearest | bicubic per plan.resize = 1 | 2 | 3, align_corners) + constant pad + crop + flips + brightness -> stage0, channel sums
//   colour    : contrast about the per-channel mean (fixed-order reduction of the block sums) + hue -> stage1 | out
//   rotate    : bilinear sample about the image centre, zero outside -> out          (only images that drew one)
// Boxes: one wave per image, ordered compaction with ballots; sets the all-boxes-lost flag the geometry stage reads.
// CPU restatement these are tested against: oracle/augment_ref.py (pinned to the reference run on oracle/tf_shim).
#include "common.h"
#include <math.h>
#include "augment_resize.h"

namespace odtk {
namespace {

constexpr int AUG_THREADS = 256, AUG_MAXC = 4;

struct AugArgs {
    const odtk_aug_plan* plans;
    const int* fallback;
    int N, C, zoom_h, zoom_w, out_h, out_w, out_chw, nblk;
    float constant_value;
    float* stage0;      // [N][oh][ow][C]
    float* stage1;      // [N][oh][ow][C]
    float* parts;       // [N][nblk][AUG_MAXC]
    float* out;
};

__device__ __forceinline__ float src_px(const odtk_aug_plan& p, int C, int y, int x, int c) {
    const size_t i = p.src_chw ? ((size_t)c * p.in_h + y) * p.in_w + x : ((size_t)y * p.in_w + x) * C + c;
    return p.src_u8 ? (float)((const unsigned char*)p.src)[i] : ((const float*)p.src)[i];
}

__device__ __forceinline__ void store_out(const AugArgs& a, int n, int y, int x, const float* v) {
    if (a.out_chw) {
        for (int c = 0; c < a.C; ++c) a.out[(((size_t)n * a.C + c) * a.out_h + y) * a.out_w + x] = v[c];
    } else {
        float* o = a.out + (((size_t)n * a.out_h + y) * a.out_w + x) * a.C;
        for (int c = 0; c < a.C; ++c) o[c] = v[c];
    }
}

// TF ResizeBilinear: top + (bottom - top) * lerp with top = tl + (tr - tl) * xlerp
__device__ __forceinline__ void bilinear(const odtk_aug_plan& p, int C, float fy, float fx, int y1cap, int x1cap, bool ceil_upper, float* v) {
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(ceil_upper ? (int)ceilf(fy) : y0 + 1, y1cap), x1 = min(ceil_upper ? (int)ceilf(fx) : x0 + 1, x1cap);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    for (int c = 0; c < C; ++c) {
        const float tl = src_px(p, C, y0, x0, c), tr = src_px(p, C, y0, x1, c);
        const float bl = src_px(p, C, y1, x0, c), br = src_px(p, C, y1, x1, c);
        const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
        v[c] = top + (bot - top) * ly;
    }
}

// TF 1.13 ResizeNearestNeighbor / ResizeBicubic with align_corners: indices and weights in augment_resize.h
__device__ __forceinline__ void nearest_align(const odtk_aug_plan& p, int C, int ys, int xs, float sy, float sx, float* v) {
    const int y = nearest_src(ys, sy, p.in_h), x = nearest_src(xs, sx, p.in_w);
    for (int c = 0; c < C; ++c) v[c] = src_px(p, C, y, x, c);
}

// bicubic: along x first (v0 w0 + v1 w1 + v2 w2 + v3 w3 in float, no contraction), then the four row results along y
__device__ __forceinline__ void bicubic_align(const odtk_aug_plan& p, int C, int ys, int xs, float sy, float sx, float* v) {
    float wy[4], wx[4];
    int iy[4], ix[4];
    bicubic_taps((float)ys * sy, p.in_h, wy, iy);
    bicubic_taps((float)xs * sx, p.in_w, wx, ix);
    for (int c = 0; c < C; ++c) {
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a0 = src_px(p, C, iy[k], ix[0], c), a1 = src_px(p, C, iy[k], ix[1], c);
            const float a2 = src_px(p, C, iy[k], ix[2], c), a3 = src_px(p, C, iy[k], ix[3], c);
            r[k] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a0, wx[0]), __fmul_rn(a1, wx[1])), __fmul_rn(a2, wx[2])), __fmul_rn(a3, wx[3]));
        }
        v[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[0], wy[0]), __fmul_rn(r[1], wy[1])), __fmul_rn(r[2], wy[2])), __fmul_rn(r[3], wy[3]));
    }
}

__global__ void __launch_bounds__(AUG_THREADS) aug_geometry_kernel(AugArgs a) {
    __shared__ float red[AUG_THREADS / 64][AUG_MAXC];
    const int n = blockIdx.y;
    const odtk_aug_plan p = a.plans[n];
    const int px = blockIdx.x * AUG_THREADS + threadIdx.x;
    const bool live = px < a.out_h * a.out_w;
    const int y = live ? px / a.out_w : 0, x = live ? px - y * a.out_w : 0;
    float v[AUG_MAXC] = {0.f, 0.f, 0.f, 0.f};
    const bool fb = a.fallback && a.fallback[n];
    if (live) {
        if (fb) {                          // gt_checker_helper: tf.image.resize(image_copy, size), TF-1.x legacy grid
            bilinear(p, a.C, (float)y * ((float)p.in_h / (float)a.out_h), (float)x * ((float)p.in_w / (float)a.out_w), p.in_h - 1, p.in_w - 1, false, v);
            store_out(a, n, y, x, v);
        } else {
            const int ys = (p.flip_td ? a.out_h - 1 - y : y) + p.crop_h, xs = (p.flip_lr ? a.out_w - 1 - x : x) + p.crop_w;
            if (ys < p.resize_h && xs < p.resize_w) {
                if (p.resize) {
                    const float sy = resize_scale_align(p.in_h, p.resize_h), sx = resize_scale_align(p.in_w, p.resize_w);
                    if (p.resize == 2) nearest_align(p, a.C, ys, xs, sy, sx, v);
                    else if (p.resize == 3) bicubic_align(p, a.C, ys, xs, sy, sx, v);
                    else bilinear(p, a.C, (float)ys * sy, (float)xs * sx, p.in_h - 1, p.in_w - 1, true, v);
                } else {
                    for (int c = 0; c < a.C; ++c) v[c] = src_px(p, a.C, ys, xs, c);
                }
            } else {
                for (int c = 0; c < a.C; ++c) v[c] = a.constant_value;
            }
            if (p.has_brightness) for (int c = 0; c < a.C; ++c) v[c] += p.brightness;
            float* o = a.stage0 + (((size_t)n * a.out_h + y) * a.out_w + x) * a.C;
            for (int c = 0; c < a.C; ++c) o[c] = v[c];
        }
    }
    if (!p.has_contrast || fb) return;     // uniform per workgroup
    // per-channel workgroup sums for the contrast mean
    for (int c = 0; c < a.C; ++c) {
        float s = live ? v[c] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < a.C) {
        float s = 0.f;
        for (int w = 0; w < AUG_THREADS / 64; ++w) s += red[w][threadIdx.x];
        a.parts[((size_t)n * a.nblk + blockIdx.x) * AUG_MAXC + threadIdx.x] = s;
    }
}

// TF AdjustHue (CPU kernel): RGB -> (hue, min, max) -> hue + delta mod 1 -> RGB
__device__ __forceinline__ void hue_shift(float* v, float delta) {
    const float r = v[0], g = v[1], b = v[2];
    const float vmax = fmaxf(fmaxf(r, g), b), vmin = fminf(fminf(r, g), b), rng = vmax - vmin;
    const float norm = rng > 0.f ? 1.f / (6.f * rng) : 0.f;
    float h = r == vmax ? norm * (g - b) : (g == vmax ? norm * (b - r) + 2.f / 6.f : norm * (r - g) + 4.f / 6.f);
    if (!(rng > 0.f)) h = 0.f;
    if (h < 0.f) h += 1.f;
    h += delta;
    h -= floorf(h);
    const float dh = h * 6.f;
    const float fi = fminf(floorf(dh), 5.f);
    const float f = dh - fi;
    const float up = vmin + rng * f, dn = vmin + rng * (1.f - f);
    switch ((int)fi) {
        case 0: v[0] = vmax; v[1] = up;   v[2] = vmin; break;
        case 1: v[0] = dn;   v[1] = vmax; v[2] = vmin; break;
        case 2: v[0] = vmin; v[1] = vmax; v[2] = up;   break;
        case 3: v[0] = vmin; v[1] = dn;   v[2] = vmax; break;
        case 4: v[0] = up;   v[1] = vmin; v[2] = vmax; break;
        default: v[0] = vmax; v[1] = vmin; v[2] = dn;  break;
    }
}

__global__ void __launch_bounds__(AUG_THREADS) aug_colour_kernel(AugArgs a) {
    __shared__ float red[AUG_THREADS / 64][AUG_MAXC];
    __shared__ float mean[AUG_MAXC];
    const int n = blockIdx.y;
    if (a.fallback && a.fallback[n]) return;
    const odtk_aug_plan p = a.plans[n];
    if (p.has_contrast) {
        for (int c = 0; c < a.C; ++c) {
            float s = 0.f;
            for (int i = threadIdx.x; i < a.nblk; i += AUG_THREADS) s += a.parts[((size_t)n * a.nblk + i) * AUG_MAXC + c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
        }
        __syncthreads();
        if (threadIdx.x < a.C) {
            float s = 0.f;
            for (int w = 0; w < AUG_THREADS / 64; ++w) s += red[w][threadIdx.x];
            mean[threadIdx.x] = s / (float)(a.out_h * a.out_w);
        }
        __syncthreads();
    }
    const int px = blockIdx.x * AUG_THREADS + threadIdx.x;
    if (px >= a.out_h * a.out_w) return;
    const int y = px / a.out_w, x = px - y * a.out_w;
    const size_t base = (((size_t)n * a.out_h + y) * a.out_w + x) * a.C;
    float v[AUG_MAXC];
    for (int c = 0; c < a.C; ++c) v[c] = a.stage0[base + c];
    if (p.has_contrast) for (int c = 0; c < a.C; ++c) v[c] = (v[c] - mean[c]) * p.contrast + mean[c];
    if (p.has_hue && a.C == 3) hue_shift(v, p.hue);
    if (p.has_rotate) {
        for (int c = 0; c < a.C; ++c) a.stage1[base + c] = v[c];
    } else {
        store_out(a, n, y, x, v);
    }
}

// tf.contrib.image.rotate(img, ang, 'BILINEAR'): out(x, y) = in(cos x - sin y + ox, sin x + cos y + oy), taps outside = 0
__global__ void __launch_bounds__(AUG_THREADS) aug_rotate_kernel(AugArgs a) {
    const int n = blockIdx.y;
    if (a.fallback && a.fallback[n]) return;
    const odtk_aug_plan p = a.plans[n];
    if (!p.has_rotate) return;
    const int px = blockIdx.x * AUG_THREADS + threadIdx.x;
    if (px >= a.out_h * a.out_w) return;
    const int y = px / a.out_w, x = px - y * a.out_w;
    const int H = a.out_h, W = a.out_w;
    const float c = cosf(p.angle), s = sinf(p.angle);
    const float ox = ((float)(W - 1) - (c * (float)(W - 1) - s * (float)(H - 1))) * 0.5f;
    const float oy = ((float)(H - 1) - (s * (float)(W - 1) + c * (float)(H - 1))) * 0.5f;
    const float sx = c * (float)x - s * (float)y + ox, sy = s * (float)x + c * (float)y + oy;
    const float x0 = floorf(sx), y0 = floorf(sy);
    const float fx = sx - x0, fy = sy - y0;
    const float* img = a.stage1 + (size_t)n * H * W * a.C;
    float v[AUG_MAXC];
    for (int ch = 0; ch < a.C; ++ch) {
        auto tap = [&](float yy, float xx) -> float {
            if (yy < 0.f || yy > (float)(H - 1) || xx < 0.f || xx > (float)(W - 1)) return 0.f;
            return img[((size_t)(int)yy * W + (int)xx) * a.C + ch];
        };
        const float top = tap(y0, x0) * (1.f - fx) + tap(y0, x0 + 1.f) * fx;
        const float bot = tap(y0 + 1.f, x0) * (1.f - fx) + tap(y0 + 1.f, x0 + 1.f) * fx;
        v[ch] = top * (1.f - fy) + bot * fy;
    }
    store_out(a, n, y, x, v);
}

// one wave per image: transform every box, keep those whose centre stays inside, compact in order, pad with -1
__global__ void __launch_bounds__(64) aug_boxes_kernel(const odtk_aug_plan* plans, const float* gt_in, const int* gt_count, int P,
                                                       int out_h, int out_w, int pad_to, float* gt_out, int* fallback) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const odtk_aug_plan p = plans[n];
    const int G = min(gt_count[n], P);
    const float oh = (float)out_h, ow = (float)out_w;
    float* out = gt_out + (size_t)n * pad_to * 5;
    int kept = 0;
    for (int b0 = 0; b0 < G; b0 += 64) {
        const int b = b0 + lane;
        bool keep = false;
        float yc = 0.f, xc = 0.f, hh = 0.f, ww = 0.f, cls = 0.f;
        if (b < G) {
            const float* g = gt_in + ((size_t)n * P + b) * 5;
            float ymin = g[0] * p.ratio_y - (float)p.crop_h, ymax = g[1] * p.ratio_y - (float)p.crop_h;
            float xmin = g[2] * p.ratio_x - (float)p.crop_w, xmax = g[3] * p.ratio_x - (float)p.crop_w;
            cls = g[4];
            if (p.flip_td) { const float t = oh - ymin - 1.f; ymin = oh - ymax - 1.f; ymax = t; }
            if (p.flip_lr) { const float t = ow - xmin - 1.f; xmin = ow - xmax - 1.f; xmax = t; }
            if (p.has_rotate) {           // rotate_helper: corners by -ang about ((w-1)/2, (h-1)/2)
                const float ang = -p.angle, c = cosf(ang), s = sinf(ang);
                const float cx = (ow - 1.f) / 2.f, cy = (oh - 1.f) / 2.f;
                const float offx = cx * (1.f - c) + cy * s, offy = cy * (1.f - c) - cx * s;
                const float ax = xmin * c - ymin * s + offx, ay = xmin * s + ymin * c + offy;
                const float bx = xmax * c - ymax * s + offx, by = xmax * s + ymax * c + offy;
                const float cx_ = xmin * c - ymax * s + offx, cy_ = xmin * s + ymax * c + offy;
                const float dx = xmax * c - ymin * s + offx, dy = xmax * s + ymin * c + offy;
                xmin = fminf(fminf(ax, bx), fminf(cx_, dx)); xmax = fmaxf(fmaxf(ax, bx), fmaxf(cx_, dx));
                ymin = fminf(fminf(ay, by), fminf(cy_, dy)); ymax = fmaxf(fmaxf(ay, by), fmaxf(cy_, dy));
            }
            ymin = fminf(fmaxf(ymin, 0.f), oh - 1.f); ymax = fminf(fmaxf(ymax, 0.f), oh - 1.f);
            xmin = fminf(fmaxf(xmin, 0.f), ow - 1.f); xmax = fminf(fmaxf(xmax, 0.f), ow - 1.f);
            yc = (ymin + ymax) / 2.f; xc = (xmin + xmax) / 2.f; hh = ymax - ymin; ww = xmax - xmin;
            keep = yc > 0.f && yc < oh - 1.f && xc > 0.f && xc < ow - 1.f;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < pad_to) {
            float* o = out + (size_t)pos * 5;
            o[0] = yc; o[1] = xc; o[2] = hh; o[3] = ww; o[4] = cls;
        }
        kept += __popcll(m);
    }
    const bool lost = kept == 0;
    if (lost) {                           // gt_checker_helper: the un-augmented boxes scaled to the output size
        const float fy = oh / (float)p.in_h, fx = ow / (float)p.in_w;
        for (int b = lane; b < min(G, pad_to); b += 64) {
            const float* g = gt_in + ((size_t)n * P + b) * 5;
            float* o = out + (size_t)b * 5;
            o[0] = (g[0] / 2.f + g[1] / 2.f) * fy; o[1] = (g[2] / 2.f + g[3] / 2.f) * fx;
            o[2] = (g[1] - g[0]) * fy; o[3] = (g[3] - g[2]) * fx; o[4] = g[4];
        }
        kept = G;
    }
    for (int i = min(kept, pad_to) * 5 + lane; i < pad_to * 5; i += 64) out[i] = -1.f;
    if (lane == 0) fallback[n] = lost ? 1 : 0;
}

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" long long odtk_augment_workspace_bytes(int N, int C, int out_h, int out_w) {
    const long long px = (long long)out_h * out_w;
    const long long nblk = (px + AUG_THREADS - 1) / AUG_THREADS;
    return 2 * (long long)N * px * C * 4 + (long long)N * nblk * AUG_MAXC * 4;
}

extern "C" int odtk_augment_boxes(const odtk_aug_plan* plans, const float* gt_in, const int* gt_count, int N, int P, int out_h,
                                  int out_w, int pad_to, float* gt_out, int* fallback, void* stream) {
    ODTK_REQUIRE(plans && gt_in && gt_count && gt_out && fallback, "augment_boxes: null pointer");
    ODTK_REQUIRE(N > 0 && P > 0 && pad_to > 0 && out_h > 1 && out_w > 1, "augment_boxes: N=%d P=%d pad_to=%d out=%dx%d out of range", N, P, pad_to, out_h, out_w);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aug_boxes_kernel, dim3(N), dim3(64), 0, st, plans, gt_in, gt_count, P, out_h, out_w, pad_to, gt_out, fallback);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_augment_images(const odtk_aug_plan* plans, const int* fallback, int N, int C, int zoom_h, int zoom_w, int out_h,
                                   int out_w, float constant_value, int out_chw, float* out, void* workspace, void* stream) {
    ODTK_REQUIRE(plans && out && workspace, "augment_images: null pointer");
    ODTK_REQUIRE(N > 0 && C > 0 && C <= AUG_MAXC, "augment_images: N=%d C=%d (C <= %d)", N, C, AUG_MAXC);
    ODTK_REQUIRE(out_h > 1 && out_w > 1 && zoom_h >= out_h && zoom_w >= out_w, "augment_images: out %dx%d must fit zoom %dx%d", out_h, out_w, zoom_h, zoom_w);
    hipStream_t st = (hipStream_t)stream;
    AugArgs a;
    a.plans = plans; a.fallback = fallback; a.N = N; a.C = C; a.zoom_h = zoom_h; a.zoom_w = zoom_w; a.out_h = out_h; a.out_w = out_w;
    a.out_chw = out_chw; a.constant_value = constant_value; a.out = out;
    a.nblk = ceil_div(out_h * out_w, AUG_THREADS);
    const size_t img = (size_t)N * out_h * out_w * C;
    a.stage0 = (float*)workspace; a.stage1 = a.stage0 + img; a.parts = a.stage1 + img;
    const dim3 grid(a.nblk, N);
    hipLaunchKernelGGL(aug_geometry_kernel, grid, dim3(AUG_THREADS), 0, st, a);
    hipLaunchKernelGGL(aug_colour_kernel, grid, dim3(AUG_THREADS), 0, st, a);
    hipLaunchKernelGGL(aug_rotate_kernel, grid, dim3(AUG_THREADS), 0, st, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
