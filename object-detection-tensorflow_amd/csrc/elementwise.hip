// HBM-bound layers of the SSD300 path (gfx950): preprocess, max-pool, batch-norm, L2-norm,
// bias-gradient column sums, fused SGD-momentum.  All NHWC "rows x pitch"; every thread moves
// 16-byte channel chunks (8 bf16 / 4 f32), lanes run along channels so a wave reads whole lines.
//
// Reference call sites: SSD300.py:52-63 (mean subtraction), :539-547 (max_pooling2d SAME),
// :506-512 (batch_normalization), :74-83 (l2_normalize * scalar), :149-154 (MomentumOptimizer
// + l2_loss over all trainables).
#include <mutex>
#include <initializer_list>
#include "common.h"

namespace odtk {
namespace {

template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float (&f)[8]) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 pack(const float (&f)[8]) {
        uint4 v;
        v.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
        v.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
        v.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
        v.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
        return v;
    }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float (&f)[4]) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float (&f)[4]) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};

template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }
template <typename T>
__device__ __forceinline__ void st16(T* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// ------------------------------------------------------------------ preprocess
template <typename T>
__global__ void preprocess_kernel(const float* __restrict__ img, long long pixels, float m0, float m1,
                                  float m2, int ldx, T* __restrict__ x) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < pixels; i += step) {
        const float r = img[i * 3 + 0] - m0, g = img[i * 3 + 1] - m1, b = img[i * 3 + 2] - m2;
        T* o = x + i * ldx;
        o[0] = elem<T>::store(r); o[1] = elem<T>::store(g); o[2] = elem<T>::store(b);
        for (int c = 3; c < ldx; ++c) o[c] = elem<T>::store(0.f);
    }
}

// bf16 rows of 8 channels (the first layer's 16-byte pixels): FOUR pixels per thread -- three 16-byte loads of 12 consecutive floats, four 16-byte
// stores -- instead of three 4-byte loads at a 12-byte stride and eight 2-byte stores per pixel (57 us for 80 MB at batch 32, round 3).  Same
// arithmetic per element (x - mean, one rounding to bf16).
__global__ void __launch_bounds__(256) preprocess_bf16x8_kernel(const float* __restrict__ img, long long pixels, float m0, float m1, float m2,
                                                                bf16_t* __restrict__ x) {
    const long long quads = pixels >> 2;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < quads; i += step) {
        const float4* p = reinterpret_cast<const float4*>(img + i * 12);
        const float4 a = p[0], b = p[1], c = p[2];
        const float v[12] = {a.x - m0, a.y - m1, a.z - m2, a.w - m0, b.x - m1, b.y - m2, b.z - m0, b.w - m1, c.x - m2, c.y - m0, c.z - m1, c.w - m2};
        uint4* o = reinterpret_cast<uint4*>(x + i * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = make_uint4((unsigned)f32_to_bf16(v[3 * q]) | ((unsigned)f32_to_bf16(v[3 * q + 1]) << 16), (unsigned)f32_to_bf16(v[3 * q + 2]), 0u, 0u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (pixels & 3)) {            // the last pixels % 4
        const long long px = (quads << 2) + threadIdx.x;
        bf16_t* o = x + px * 8;
        o[0] = f32_to_bf16(img[px * 3] - m0); o[1] = f32_to_bf16(img[px * 3 + 1] - m1); o[2] = f32_to_bf16(img[px * 3 + 2] - m2);
        for (int c = 3; c < 8; ++c) o[c] = 0;
    }
}

// ------------------------------------------------------------------ max pool
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                   int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const long long total = (long long)N * Ho * Wo * chunks;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const int ch = (int)(i % chunks);
        long long pix = i / chunks;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float best[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) best[e] = -INFINITY;
        for (int r = 0; r < k; ++r) {
            const int h = ho * stride - pad_t + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s = 0; s < k; ++s) {
                const int w = wo * stride - pad_l + s;
                if ((unsigned)w >= (unsigned)W) continue;
                float f[KC];
                Chunk<T>::unpack(ld16(x + ((size_t)(n * H + h) * W + w) * ld + ch * KC), f);
#pragma unroll
                for (int e = 0; e < KC; ++e) best[e] = f[e] > best[e] ? f[e] : best[e];
            }
        }
        st16(y + ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC, Chunk<T>::pack(best));
    }
}

// Overlapping windows (pool5: 3x3 / stride 1) with a RECORDED arg-max: the forward pass also stores, per output chunk, 4 bits per channel = the
// window position r * k + s of the first maximum in scan order (k <= 3); the backward pass is then a gather over the <= k x k windows that contain a
// pixel which reads 4 B of codes + 16 B of dy per window -- instead of re-deriving every window's first maximum from up to k x k loads of x
// (maxpool_bwd_kernel below: pool5 at batch 32 took 72 us for a 12-MB map).
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_argmax_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned* __restrict__ arg, int N, int H,
                                                                 int W, int C, int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * Ho * Wo * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = (int)(i % (unsigned)chunks);
        unsigned pix = i / (unsigned)chunks;
        const int wo = (int)(pix % (unsigned)Wo); pix /= (unsigned)Wo;
        const int ho = (int)(pix % (unsigned)Ho);
        const int n = (int)(pix / (unsigned)Ho);
        float best[KC];
        unsigned code = 0;
#pragma unroll
        for (int e = 0; e < KC; ++e) best[e] = -INFINITY;
        for (int r = 0; r < k; ++r) {
            const int h = ho * stride - pad_t + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s_ = 0; s_ < k; ++s_) {
                const int w = wo * stride - pad_l + s_;
                if ((unsigned)w >= (unsigned)W) continue;
                float f[KC];
                Chunk<T>::unpack(ld16(x + ((size_t)(n * H + h) * W + w) * ld + ch * KC), f);
                const unsigned pos = (unsigned)(r * k + s_);
#pragma unroll
                for (int e = 0; e < KC; ++e)
                    if (f[e] > best[e]) {                     // strict >: the first maximum in scan order keeps the window
                        best[e] = f[e];
                        code = (code & ~(15u << (4 * e))) | (pos << (4 * e));
                    }
            }
        }
        st16(y + ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC, Chunk<T>::pack(best));
        arg[i] = code;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_argmax_kernel(const unsigned* __restrict__ arg, const T* __restrict__ dy, T* __restrict__ dx, int N,
                                                                 int H, int W, int C, int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * H * W * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = (int)(i % (unsigned)chunks);
        unsigned pix = i / (unsigned)chunks;
        const int w = (int)(pix % (unsigned)W); pix /= (unsigned)W;
        const int h = (int)(pix % (unsigned)H);
        const int n = (int)(pix / (unsigned)H);
        float g[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) g[e] = 0.f;
        int ho_lo = (h + pad_t - k + 1 + stride - 1);
        ho_lo = ho_lo < 0 ? 0 : ho_lo / stride;
        int ho_hi = (h + pad_t) / stride; if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        int wo_lo = (w + pad_l - k + 1 + stride - 1);
        wo_lo = wo_lo < 0 ? 0 : wo_lo / stride;
        int wo_hi = (w + pad_l) / stride; if (wo_hi > Wo - 1) wo_hi = Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const size_t o = (size_t)(n * Ho + ho) * Wo + wo;
                const unsigned code = arg[o * chunks + ch];
                const unsigned pos = (unsigned)((h - (ho * stride - pad_t)) * k + (w - (wo * stride - pad_l)));   // this pixel's position in that window
                float dv[KC];
                Chunk<T>::unpack(ld16(dy + o * ld + ch * KC), dv);
#pragma unroll
                for (int e = 0; e < KC; ++e) g[e] += ((code >> (4 * e)) & 15u) == pos ? dv[e] : 0.f;
            }
        st16(dx + ((size_t)(n * H + h) * W + w) * ld + ch * KC, Chunk<T>::pack(g));
    }
}

// gather formulation: (h,w) receives dy of window (ho,wo) iff it is the FIRST position of that
// window (row-major scan) whose value equals the window max.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                   T* __restrict__ dx, int N, int H, int W, int C, int ld, int Ho, int Wo,
                                   int k, int stride, int pad_t, int pad_l) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * H * W * chunks;           // host checks < 2^31
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = (int)(i % (unsigned)chunks);
        unsigned pix = i / (unsigned)chunks;
        const int w = (int)(pix % (unsigned)W); pix /= (unsigned)W;
        const int h = (int)(pix % (unsigned)H);
        const int n = (int)(pix / (unsigned)H);
        float xv[KC], g[KC];
        Chunk<T>::unpack(ld16(x + ((size_t)(n * H + h) * W + w) * ld + ch * KC), xv);
#pragma unroll
        for (int e = 0; e < KC; ++e) g[e] = 0.f;
        // windows containing (h, w): ho*stride - pad_t <= h <= ho*stride - pad_t + k - 1
        int ho_lo = (h + pad_t - k + 1 + stride - 1);
        ho_lo = ho_lo < 0 ? 0 : ho_lo / stride;
        int ho_hi = (h + pad_t) / stride; if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        int wo_lo = (w + pad_l - k + 1 + stride - 1);
        wo_lo = wo_lo < 0 ? 0 : wo_lo / stride;
        int wo_hi = (w + pad_l) / stride; if (wo_hi > Wo - 1) wo_hi = Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const size_t oo = ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC;
                float yv[KC], dv[KC];
                Chunk<T>::unpack(ld16(y + oo), yv);
                Chunk<T>::unpack(ld16(dy + oo), dv);
                bool first[KC];
#pragma unroll
                for (int e = 0; e < KC; ++e) first[e] = (xv[e] == yv[e]);
                const int h0 = ho * stride - pad_t, w0 = wo * stride - pad_l;
                for (int r = 0; r < k; ++r) {
                    const int hh = h0 + r;
                    if (hh > h) break;
                    if ((unsigned)hh >= (unsigned)H) continue;
                    for (int s = 0; s < k; ++s) {
                        const int ww = w0 + s;
                        if (hh == h && ww >= w) break;
                        if ((unsigned)ww >= (unsigned)W) continue;
                        float f[KC];
                        Chunk<T>::unpack(ld16(x + ((size_t)(n * H + hh) * W + ww) * ld + ch * KC), f);
#pragma unroll
                        for (int e = 0; e < KC; ++e) first[e] = first[e] && (f[e] != yv[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < KC; ++e) g[e] += first[e] ? dv[e] : 0.f;
            }
        }
        st16(dx + ((size_t)(n * H + h) * W + w) * ld + ch * KC, Chunk<T>::pack(g));
    }
}

// 2x2 / stride-2 / pad_before-0 pooling backward (pool1..pool4): the windows tile the input without
// overlap, so one thread owns one OUTPUT window chunk: it reads the four input pixels once, finds the
// FIRST maximum in window scan order (TF routes the gradient there) and writes the four dx pixels.
// No re-reads of neighbours, no 64-bit div/mod per element.
template <typename T>
__global__ void __launch_bounds__(256) maxpool2x2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             T* __restrict__ dx, int N, int H, int W, int C, int ld,
                                                             int Ho, int Wo) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * Ho * Wo * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % (unsigned)chunks;
        unsigned pix = i / (unsigned)chunks;
        const unsigned wo = pix % (unsigned)Wo; pix /= (unsigned)Wo;
        const unsigned ho = pix % (unsigned)Ho;
        const unsigned n = pix / (unsigned)Ho;
        const int h0 = (int)ho * 2, w0 = (int)wo * 2;
        float dv[KC];
        Chunk<T>::unpack(ld16(dy + ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC), dv);
        float xv[4][KC];
        bool in[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hh = h0 + (t >> 1), ww = w0 + (t & 1);
            in[t] = hh < H && ww < W;
            if (in[t]) Chunk<T>::unpack(ld16(x + ((size_t)((int)n * H + hh) * W + ww) * ld + ch * KC), xv[t]);
        }
        float g[4][KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            float m = xv[0][e];                       // (h0, w0) is always inside the image
#pragma unroll
            for (int t = 1; t < 4; ++t)
                if (in[t]) m = fmaxf(m, xv[t][e]);
            bool done = false;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool hit = !done && in[t] && xv[t][e] == m;
                g[t][e] = hit ? dv[e] : 0.f;
                done = done || hit;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hh = h0 + (t >> 1), ww = w0 + (t & 1);
            if (in[t]) st16(dx + ((size_t)((int)n * H + hh) * W + ww) * ld + ch * KC, Chunk<T>::pack(g[t]));
        }
    }
}

// 2x2 / stride-2 pooling with a recorded arg-max: the forward pass stores, per output chunk, 2 bits per channel =
// the FIRST window position (scan order) that holds the maximum; the backward pass then needs neither x nor y:
// it reads dy + 2 B of index per 16-B chunk and writes the four dx pixels (pool1: 472 MB instead of 922 MB).
template <typename T>
__global__ void __launch_bounds__(256) maxpool2x2_fwd_idx_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 unsigned short* __restrict__ idx, int N, int H, int W, int C,
                                                                 int ld, int Ho, int Wo) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * Ho * Wo * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % (unsigned)chunks;
        unsigned pix = i / (unsigned)chunks;
        const unsigned wo = pix % (unsigned)Wo; pix /= (unsigned)Wo;
        const unsigned ho = pix % (unsigned)Ho;
        const unsigned n = pix / (unsigned)Ho;
        const int h0 = (int)ho * 2, w0 = (int)wo * 2;
        float xv[4][KC];
        bool in[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hh = h0 + (t >> 1), ww = w0 + (t & 1);
            in[t] = hh < H && ww < W;
            if (in[t]) Chunk<T>::unpack(ld16(x + ((size_t)((int)n * H + hh) * W + ww) * ld + ch * KC), xv[t]);
        }
        float best[KC];
        unsigned code = 0;
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            float m = xv[0][e];                       // (h0, w0) is always inside the image
            int am = 0;
#pragma unroll
            for (int t = 1; t < 4; ++t)
                if (in[t] && xv[t][e] > m) { m = xv[t][e]; am = t; }      // strict >: the first maximum wins
            best[e] = m;
            code |= (unsigned)am << (2 * e);
        }
        st16(y + ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC, Chunk<T>::pack(best));
        idx[i] = (unsigned short)code;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool2x2_bwd_idx_kernel(const unsigned short* __restrict__ idx, const T* __restrict__ dy,
                                                                 T* __restrict__ dx, int N, int H, int W, int C, int ld,
                                                                 int Ho, int Wo) {
    constexpr int KC = Chunk<T>::N;
    const int chunks = C / KC;
    const unsigned total = (unsigned)N * Ho * Wo * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % (unsigned)chunks;
        unsigned pix = i / (unsigned)chunks;
        const unsigned wo = pix % (unsigned)Wo; pix /= (unsigned)Wo;
        const unsigned ho = pix % (unsigned)Ho;
        const unsigned n = pix / (unsigned)Ho;
        const int h0 = (int)ho * 2, w0 = (int)wo * 2;
        float dv[KC];
        Chunk<T>::unpack(ld16(dy + ((size_t)(n * Ho + ho) * Wo + wo) * ld + ch * KC), dv);
        const unsigned code = idx[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hh = h0 + (t >> 1), ww = w0 + (t & 1);
            if (hh >= H || ww >= W) continue;
            float g[KC];
#pragma unroll
            for (int e = 0; e < KC; ++e) g[e] = ((code >> (2 * e)) & 3u) == (unsigned)t ? dv[e] : 0.f;
            st16(dx + ((size_t)((int)n * H + hh) * W + ww) * ld + ch * KC, Chunk<T>::pack(g));
        }
    }
}

// ------------------------------------------------------------------ column reductions
// Block = 32 row-lanes x 8 chunk-lanes; grid (column groups of 8 chunks, row splits).
// Deterministic: partials to ws[slot][split][C], reduced in a fixed order by the consumer.
constexpr int RED_ROWS = 32;

// Narrow bf16 maps (8, 16 or 32 channels: one 16-byte chunk lane out of 8, 4 or 2 would have work -- DLA-34's and DarkNet-53's first layers, the LARGEST maps
// of their steps): the block is 256 / CPR row lanes x CPR chunk lanes instead of 32 x 8, so every lane streams (round 5, late).  bn_lane_cpr() = chunk lanes
// per row for this launch; rl = tid / cpr, cl = tid % cpr, row stride 256 / cpr.  With cpr = 8 everything below is what it was, bit for bit.
template <typename T>
__device__ __forceinline__ int bn_lane_cpr(int C, int ld) {
    return (sizeof(T) == 2 && gridDim.x == 1 && ld == C && (C == 8 || C == 16 || C == 32)) ? C / 8 : 8;
}
// tree over the row lanes of a [256 / cpr][cpr][NV] block; the sums of chunk lane cl end up in every thread with that cl (same order as the 32 x 8 form for cpr = 8)
template <int NV>
__device__ __forceinline__ void block_rowlane_reduce_cpr(float (&v)[NV], float* sm /*[256][NV]*/, int cl, int cpr) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < NV; ++e) sm[tid * NV + e] = v[e];
    __syncthreads();
    for (int s = 128; s >= cpr; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int e = 0; e < NV; ++e) sm[tid * NV + e] += sm[(tid + s) * NV + e];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = sm[cl * NV + e];
    __syncthreads();
}

template <int NV>
__device__ __forceinline__ void block_rowlane_reduce(float (&v)[NV], float* sm /*[32][8][NV]*/, int rl, int cl) {
    // sm layout [rl][cl][NV]
#pragma unroll
    for (int e = 0; e < NV; ++e) sm[(rl * 8 + cl) * NV + e] = v[e];
    __syncthreads();
    for (int s = RED_ROWS / 2; s > 0; s >>= 1) {
        if (rl < s) {
#pragma unroll
            for (int e = 0; e < NV; ++e) sm[(rl * 8 + cl) * NV + e] += sm[((rl + s) * 8 + cl) * NV + e];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = sm[cl * NV + e];
    __syncthreads();
}

// BN stats: sum(z - shift), sum((z - shift)^2), shift = z[0][c]
template <typename T>
__global__ void __launch_bounds__(256) bn_stats_kernel(const T* __restrict__ z, int M, int C, int ldz,
                                                       int rows_per_split, float* __restrict__ ws) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cpr = bn_lane_cpr<T>(C, ldz), rstride = 256 / cpr;
    const int cl = threadIdx.x & (cpr - 1), rl = threadIdx.x / cpr;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    const int split = blockIdx.y, nsplit = gridDim.y;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        float sh[KC];
        Chunk<T>::unpack(ld16(z + c0), sh);
        const int m0 = split * rows_per_split;
        int m1 = m0 + rows_per_split; if (m1 > M) m1 = M;
#pragma unroll 4
        for (int m = m0 + rl; m < m1; m += rstride) {
            float f[KC];
            Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                const float d = f[e] - sh[e];
                acc[e] += d;
                acc[KC + e] += d * d;
            }
        }
    }
    block_rowlane_reduce_cpr<2 * KC>(acc, sm, cl, cpr);
    if (rl == 0 && c0 < C) {
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            if (c0 + e < C) {
                ws[((size_t)0 * nsplit + split) * C + c0 + e] = acc[e];
                ws[((size_t)1 * nsplit + split) * C + c0 + e] = acc[KC + e];
            }
        }
    }
}

__device__ __forceinline__ long long out_off(int m, int rows_per_img, long long img_stride, int ld) {
    const int n = m / rows_per_img;
    return (long long)n * img_stride + (long long)(m - n * rows_per_img) * ld;
}

// per-channel scale/offset from the partial sums (training) or the moving stats (inference);
// one thread per channel.  fin[0*C + c] = scale, fin[1*C + c] = offset.
// Sum the [2][nsplit][C] partials of 32 channels with 8 split-lanes per channel (fixed order ->
// deterministic), result valid in the threads with sl == 0.  Block = 256 threads.
// (round 5, measured and NOT kept: 32 split lanes per channel in 1 024-thread workgroups -- YOLOv3 b8 811 -> 816 images/s, within the box noise, and the changed
//  summation order of the statistics flipped near-zero Adam updates in CenterNet's exact-engine step test)
constexpr int FIN_CH = 32, FIN_SL = 8;
__device__ __forceinline__ void reduce_partials(const float* __restrict__ ws, int nsplit, int C, int c, int sl,
                                                float& s1, float& s2) {
    __shared__ float sm[2][FIN_SL][FIN_CH];
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        // four partial rows per trip with their eight loads in flight together (one row per trip was up to 128 dependent L2 round trips for the narrow layers
        // with ~1 000 row splits: 20-40 us per launch in RetinaNet's / YOLOv3's steps); the summation order is fixed, so the result stays deterministic
        const float* w1 = ws + c;
        const float* w2 = ws + (size_t)nsplit * C + c;
        float b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
        int s = sl;
        for (; s + 3 * FIN_SL < nsplit; s += 4 * FIN_SL) {
            float v1[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v1[u] = w1[(size_t)(s + u * FIN_SL) * C];
                v2[u] = w2[(size_t)(s + u * FIN_SL) * C];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { b1[u] += v1[u]; b2[u] += v2[u]; }
        }
        for (; s < nsplit; s += FIN_SL) {
            b1[0] += w1[(size_t)s * C];
            b2[0] += w2[(size_t)s * C];
        }
        a1 = (b1[0] + b1[1]) + (b1[2] + b1[3]);
        a2 = (b2[0] + b2[1]) + (b2[2] + b2[3]);
    }
    const int cl = threadIdx.x & (FIN_CH - 1);
    sm[0][sl][cl] = a1;
    sm[1][sl][cl] = a2;
    __syncthreads();
    s1 = 0.f; s2 = 0.f;
    if (sl == 0) {
#pragma unroll
        for (int k = 0; k < FIN_SL; ++k) { s1 += sm[0][k][cl]; s2 += sm[1][k][cl]; }
    }
}

// per-channel scale/offset from the partial sums (training) or the moving stats (inference).
// fin[0*C + c] = scale, fin[1*C + c] = offset.  Grid = ceil(C / 32) blocks of 256 threads.
template <typename T>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_finalize_kernel(const T* __restrict__ z, int M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ mmean,
                                   float* __restrict__ mvar, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, int training, const float* __restrict__ ws,
                                   int nsplit, float* __restrict__ fin) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), sl = threadIdx.x / FIN_CH;
    float s1 = 0.f, s2 = 0.f;
    if (training) reduce_partials(ws, nsplit, C, c, sl, s1, s2);
    if (c >= C || sl != 0) return;
    float mean, var;
    if (training) {
        const float d = s1 / (float)M;
        mean = elem<T>::load(z[c]) + d;
        var = fmaxf(s2 / (float)M - d * d, 0.f);
        save_mean[c] = mean;
        save_invstd[c] = rsqrtf(var + 1e-3f);
        const float unb = var * ((float)M / (float)(M > 1 ? M - 1 : 1));
        mmean[c] = mmean[c] * 0.99f + mean * (1.f - 0.99f);
        mvar[c] = mvar[c] * 0.99f + unb * (1.f - 0.99f);
    } else {
        mean = mmean[c]; var = mvar[c];
    }
    const float sc = rsqrtf(var + 1e-3f) * gamma[c];
    fin[c] = sc;
    fin[C + c] = beta[c] - mean * sc;
}

template <typename T, typename TY>
__global__ void __launch_bounds__(256) bn_apply_kernel(
    const T* __restrict__ z, int M, int C, int ldz, int relu, TY* __restrict__ y, int ldy, int rows_per_img,
    long long y_img_stride, int vec_ok, const float* __restrict__ fin, int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = bn_lane_cpr<T>(C, ldz), rstride = 256 / cpr;
    const int cl = threadIdx.x & (cpr - 1), rl = threadIdx.x / cpr;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    if (c0 >= C) return;
    float sc[KC], of[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        sc[e] = c0 + e < C ? fin[c0 + e] : 0.f;
        of[e] = c0 + e < C ? fin[C + c0 + e] : 0.f;
    }
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
    const bool full = c0 + KC <= C;
#pragma unroll 2
    for (int m = m0 + rl; m < m1; m += rstride) {
        float f[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            f[e] = f[e] * sc[e] + of[e];
            if (relu == 1) f[e] = fmaxf(f[e], 0.f);
            else if (relu == 2) f[e] = f[e] > 0.f ? f[e] : 0.1f * f[e];      // tf.nn.leaky_relu(x, 0.1)
        }
        TY* yp = y + out_off(m, rows_per_img, y_img_stride, ldy) + c0;
        if (full && vec_ok) {
            if (sizeof(TY) == 2) {
                float f8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f8[e] = f[e % KC];
                st16(reinterpret_cast<bf16_t*>(yp), Chunk<bf16_t>::pack(f8));   // only reached when KC == 8
            } else {
#pragma unroll
                for (int q = 0; q < KC / 4; ++q)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yp) + 4 * q) =
                        make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < KC; ++e)
                if (c0 + e < C) yp[e] = elem<TY>::store(f[e]);
        }
    }
}

// Training-mode apply with the finalize step folded in (round 2 experiment, OFF by default: odtk_debug_set(4, -2)).  One launch less per batch norm, but
// measured SLOWER: YOLOv3 416 x 416 batch 8 12.10 ms/step against 11.01 with the separate finalize kernel, SSD300 equal -- every one of the apply pass's
// workgroups re-reduces the partials, which costs more than the 2-16 workgroup finalize launch it saves.  Every workgroup reduces the row-split partials of ITS 8 chunks of channels (32 row lanes
// take the splits, then the LDS tree of the statistics kernel), derives scale / offset, and the workgroups of the first row block also write save_mean /
// save_invstd and update the moving statistics.
template <typename T, typename TY>
__global__ void __launch_bounds__(256) bn_apply_fin_kernel(
    const T* __restrict__ z, int M, int C, int ldz, int relu, TY* __restrict__ y, int ldy, int rows_per_img, long long y_img_stride, int vec_ok,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mmean, float* __restrict__ mvar,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, const float* __restrict__ ws, int nsplit, int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C)
        for (int s = rl; s < nsplit; s += RED_ROWS) {
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                if (c0 + e < C) {
                    acc[e] += ws[((size_t)0 * nsplit + s) * C + c0 + e];
                    acc[KC + e] += ws[((size_t)1 * nsplit + s) * C + c0 + e];
                }
            }
        }
    block_rowlane_reduce<2 * KC>(acc, sm, rl, cl);
    if (c0 >= C) return;
    float sh[KC], sc[KC], of[KC];
    Chunk<T>::unpack(ld16(z + c0), sh);
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        sc[e] = 0.f; of[e] = 0.f;
        const int c = c0 + e;
        if (c >= C) continue;
        const float d = acc[e] / (float)M;
        const float mean = sh[e] + d;
        const float var = fmaxf(acc[KC + e] / (float)M - d * d, 0.f);
        const float inv = rsqrtf(var + 1e-3f);
        sc[e] = inv * gamma[c];
        of[e] = beta[c] - mean * sc[e];
        if (blockIdx.y == 0 && rl == 0) {
            save_mean[c] = mean;
            save_invstd[c] = inv;
            const float unb = var * ((float)M / (float)(M > 1 ? M - 1 : 1));
            mmean[c] = mmean[c] * 0.99f + mean * (1.f - 0.99f);
            mvar[c] = mvar[c] * 0.99f + unb * (1.f - 0.99f);
        }
    }
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
    const bool full = c0 + KC <= C;
#pragma unroll 2
    for (int m = m0 + rl; m < m1; m += RED_ROWS) {
        float f[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            f[e] = f[e] * sc[e] + of[e];
            if (relu == 1) f[e] = fmaxf(f[e], 0.f);
            else if (relu == 2) f[e] = f[e] > 0.f ? f[e] : 0.1f * f[e];
        }
        TY* yp = y + out_off(m, rows_per_img, y_img_stride, ldy) + c0;
        if (full && vec_ok) {
            if (sizeof(TY) == 2) {
                float f8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f8[e] = f[e % KC];
                st16(reinterpret_cast<bf16_t*>(yp), Chunk<bf16_t>::pack(f8));   // only reached when KC == 8
            } else {
#pragma unroll
                for (int q = 0; q < KC / 4; ++q)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yp) + 4 * q) =
                        make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < KC; ++e)
                if (c0 + e < C) yp[e] = elem<TY>::store(f[e]);
        }
    }
}

// dy / y operands of the BN backward kernels: one 16-byte load per chunk when the (y, dy) layout allows it (bf16 rows with a
// 16-byte-aligned pitch: every extra layer), element loads otherwise (the heads read d(pred) rows of 25 floats).
template <typename T, typename TY>
__device__ __forceinline__ void bn_load_dy(const TY* __restrict__ y, const TY* __restrict__ dy, long long oo, int c0, int C,
                                           int relu, int vec_ok, float (&d)[Chunk<T>::N]) {
    constexpr int KC = Chunk<T>::N;
    if (sizeof(TY) == 2 && KC == 8 && vec_ok && c0 + KC <= C) {
        float yv[8], dv[8];
        Chunk<bf16_t>::unpack(ld16(reinterpret_cast<const bf16_t*>(dy) + oo), dv);
        if (relu) Chunk<bf16_t>::unpack(ld16(reinterpret_cast<const bf16_t*>(y) + oo), yv);
#pragma unroll
        for (int e = 0; e < KC; ++e) d[e] = (relu && !(yv[e % 8] > 0.f)) ? (relu == 2 ? 0.1f * dv[e % 8] : 0.f) : dv[e % 8];
        return;
    }
    if (sizeof(TY) == 4 && KC == 4 && vec_ok && c0 + KC <= C) {     // f32 engines: one 16-byte load per operand (four 4-byte loads each before round 4)
        const float4 dv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy) + oo);
        float dd[4] = {dv.x, dv.y, dv.z, dv.w};
        if (relu) {
            const float4 yv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(y) + oo);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(yy[e] > 0.f)) dd[e] = relu == 2 ? 0.1f * dd[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < KC; ++e) d[e] = dd[e % 4];
        return;
    }
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        d[e] = 0.f;
        if (c0 + e >= C) continue;
        float v = elem<TY>::load(dy[oo + e]);
        if (relu && !(elem<TY>::load(y[oo + e]) > 0.f)) v = relu == 2 ? 0.1f * v : 0.f;
        d[e] = v;
    }
}

// BN backward partials: sum(dy'), sum(dy' * xhat)
template <typename T, typename TY>
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(
    const T* __restrict__ z, const TY* __restrict__ y, const TY* __restrict__ dy, int M, int C, int ldz, int ldy,
    int rows_per_img, long long y_img_stride, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, int relu, int vec_ok, int rows_per_split, float* __restrict__ ws) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cpr = bn_lane_cpr<T>(C, ldz), rstride = 256 / cpr;
    const int cl = threadIdx.x & (cpr - 1), rl = threadIdx.x / cpr;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    const int split = blockIdx.y, nsplit = gridDim.y;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        float mu[KC], iv[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            mu[e] = c0 + e < C ? save_mean[c0 + e] : 0.f;
            iv[e] = c0 + e < C ? save_invstd[c0 + e] : 0.f;
        }
        const int m0 = split * rows_per_split;
        int m1 = m0 + rows_per_split; if (m1 > M) m1 = M;
#pragma unroll 2
        for (int m = m0 + rl; m < m1; m += rstride) {
            float f[KC];
            Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
            const long long oo = out_off(m, rows_per_img, y_img_stride, ldy) + c0;
            float d[KC];
            bn_load_dy<T, TY>(y, dy, oo, c0, C, relu, vec_ok, d);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                if (c0 + e >= C) continue;
                acc[e] += d[e];
                acc[KC + e] += d[e] * ((f[e] - mu[e]) * iv[e]);
            }
        }
    }
    block_rowlane_reduce_cpr<2 * KC>(acc, sm, cl, cpr);
    if (rl == 0 && c0 < C) {
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            if (c0 + e < C) {
                ws[((size_t)0 * nsplit + split) * C + c0 + e] = acc[e];
                ws[((size_t)1 * nsplit + split) * C + c0 + e] = acc[KC + e];
            }
        }
    }
}

// fin[0*C+c] = mean(dy'), fin[1*C+c] = mean(dy' * xhat); also emits dbeta / dgamma
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_bwd_finalize_kernel(const float* __restrict__ ws, int nsplit, int C, int M,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ fin) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), sl = threadIdx.x / FIN_CH;
    float s1, s2;
    reduce_partials(ws, nsplit, C, c, sl, s1, s2);
    if (c >= C || sl != 0) return;
    dbeta[c] = s1;
    dgamma[c] = s2;
    fin[c] = s1 / (float)M;
    fin[C + c] = s2 / (float)M;
}

template <typename T, typename TY>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(
    const T* __restrict__ z, const TY* __restrict__ y, const TY* __restrict__ dy, int M, int C, int ldz, int ldy,
    int rows_per_img, long long y_img_stride, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu, int vec_ok, T* __restrict__ dz,
    const float* __restrict__ fin, int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = bn_lane_cpr<T>(C, ldz), rstride = 256 / cpr;
    const int cl = threadIdx.x & (cpr - 1), rl = threadIdx.x / cpr;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    if (c0 >= ldz) return;
    float mu[KC], iv[KC], gs[KC], k1[KC], k2[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int c = c0 + e;
        mu[e] = 0.f; iv[e] = 0.f; gs[e] = 0.f; k1[e] = 0.f; k2[e] = 0.f;
        if (c < C) {
            mu[e] = save_mean[c]; iv[e] = save_invstd[c];
            gs[e] = gamma[c] * iv[e];
            k1[e] = fin[c];
            k2[e] = fin[C + c];
        }
    }
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
#pragma unroll 2
    for (int m = m0 + rl; m < m1; m += rstride) {
        float f[KC], o[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
        const long long oo = out_off(m, rows_per_img, y_img_stride, ldy) + c0;
        float d[KC];
        bn_load_dy<T, TY>(y, dy, oo, c0, C, relu, vec_ok, d);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            o[e] = 0.f;
            if (c0 + e >= C) continue;
            const float xh = (f[e] - mu[e]) * iv[e];
            o[e] = gs[e] * (d[e] - k1[e] - xh * k2[e]);
        }
        st16(dz + (size_t)m * ldz + c0, Chunk<T>::pack(o));
    }
}

// backward apply with the finalize step folded in (see bn_apply_fin_kernel): the row-split partials of sum(dy') and sum(dy' * xhat) are reduced by every
// workgroup for its own channels; the workgroups of the first row block write dgamma / dbeta.
template <typename T, typename TY>
__global__ void __launch_bounds__(256) bn_bwd_apply_fin_kernel(
    const T* __restrict__ z, const TY* __restrict__ y, const TY* __restrict__ dy, int M, int C, int ldz, int ldy,
    int rows_per_img, long long y_img_stride, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu, int vec_ok, T* __restrict__ dz,
    const float* __restrict__ ws, int nsplit, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C)
        for (int s = rl; s < nsplit; s += RED_ROWS) {
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                if (c0 + e < C) {
                    acc[e] += ws[((size_t)0 * nsplit + s) * C + c0 + e];
                    acc[KC + e] += ws[((size_t)1 * nsplit + s) * C + c0 + e];
                }
            }
        }
    block_rowlane_reduce<2 * KC>(acc, sm, rl, cl);
    if (c0 >= ldz) return;
    float mu[KC], iv[KC], gs[KC], k1[KC], k2[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int c = c0 + e;
        mu[e] = 0.f; iv[e] = 0.f; gs[e] = 0.f; k1[e] = 0.f; k2[e] = 0.f;
        if (c < C) {
            mu[e] = save_mean[c]; iv[e] = save_invstd[c];
            gs[e] = gamma[c] * iv[e];
            k1[e] = acc[e] / (float)M;
            k2[e] = acc[KC + e] / (float)M;
            if (blockIdx.y == 0 && rl == 0) { dbeta[c] = acc[e]; dgamma[c] = acc[KC + e]; }
        }
    }
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
#pragma unroll 2
    for (int m = m0 + rl; m < m1; m += RED_ROWS) {
        float f[KC], o[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
        const long long oo = out_off(m, rows_per_img, y_img_stride, ldy) + c0;
        float d[KC];
        bn_load_dy<T, TY>(y, dy, oo, c0, C, relu, vec_ok, d);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            o[e] = 0.f;
            if (c0 + e >= C) continue;
            const float xh = (f[e] - mu[e]) * iv[e];
            o[e] = gs[e] * (d[e] - k1[e] - xh * k2[e]);
        }
        st16(dz + (size_t)m * ldz + c0, Chunk<T>::pack(o));
    }
}

// ------------------------------------------------------------------ batch norm of SMALL maps in one launch
// The extra layers and the heads of the small feature maps (10 x 10 and below at batch 32: <= 3200 rows) spend their time in
// launch and dependency latency, not in bandwidth: stats -> finalize -> apply are three dependent launches of 2..16
// workgroups each (5-10 us apiece in the profile, r02i timeline).  Up to 1024 rows (5 x 5 and below at batch 32; measured: at
// 3200 rows one workgroup per 64 channels streams too slowly and the step LOSES 3 %) ONE launch does all three: a
// 512-thread workgroup per 8 chunks of channels (64 row lanes x 8 chunk lanes) sums its columns, finalizes them in
// registers and walks the rows a second time (L2 hits) to apply.  Same formulas as the three-kernel path; the
// summation order differs (one block instead of row splits), so results agree to f32 round-off, not bit for bit.

// CL chunk lanes (8 channels each in bf16) x 512 / CL row lanes.  CL = 8 is the round-2 shape (one workgroup per 64 channels, 64 row lanes: a 800-row map
// is 13 dependent iterations per pass on 2-4 workgroups -- 20-60 us per launch in the round-2 timeline); CL = 1 gives one workgroup per 16-byte channel
// chunk and 512 row lanes: 16-64 workgroups, 2 iterations.
template <int NV, int CL>
__device__ __forceinline__ void block_rowlane_reduce_small(float (&v)[NV], float* sm /*[512 / CL][CL][NV]*/, int rl, int cl) {
    constexpr int RL = 512 / CL;
#pragma unroll
    for (int e = 0; e < NV; ++e) sm[(rl * CL + cl) * NV + e] = v[e];
    __syncthreads();
    for (int s = RL / 2; s > 0; s >>= 1) {
        if (rl < s) {
#pragma unroll
            for (int e = 0; e < NV; ++e) sm[(rl * CL + cl) * NV + e] += sm[((rl + s) * CL + cl) * NV + e];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = sm[cl * NV + e];
}

template <typename T, typename TY, int CL>
__global__ void __launch_bounds__(512) bn_fwd_small_kernel(
    const T* __restrict__ z, int M, int C, int ldz, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ mmean, float* __restrict__ mvar, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    int relu, TY* __restrict__ y, int ldy, int rows_per_img, long long y_img_stride, int vec_ok) {
    constexpr int KC = Chunk<T>::N;
    constexpr int SM_ROWS = 512 / CL;
    __shared__ float sm[512 * 2 * KC];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c0 = (blockIdx.x * CL + cl) * KC;
    float acc[2 * KC], sh[KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int e = 0; e < KC; ++e) sh[e] = 0.f;
    if (c0 < C) {
        Chunk<T>::unpack(ld16(z + c0), sh);
#pragma unroll 8
        for (int m = rl; m < M; m += SM_ROWS) {
            float f[KC];
            Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                const float d = f[e] - sh[e];
                acc[e] += d;
                acc[KC + e] += d * d;
            }
        }
    }
    block_rowlane_reduce_small<2 * KC, CL>(acc, sm, rl, cl);
    if (c0 >= C) return;
    float sc[KC], of[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        sc[e] = 0.f; of[e] = 0.f;
        const int c = c0 + e;
        if (c >= C) continue;
        const float d = acc[e] / (float)M;
        const float mean = sh[e] + d;
        const float var = fmaxf(acc[KC + e] / (float)M - d * d, 0.f);
        const float inv = rsqrtf(var + 1e-3f);
        sc[e] = inv * gamma[c];
        of[e] = beta[c] - mean * sc[e];
        if (rl == 0) {
            save_mean[c] = mean;
            save_invstd[c] = inv;
            const float unb = var * ((float)M / (float)(M > 1 ? M - 1 : 1));
            mmean[c] = mmean[c] * 0.99f + mean * (1.f - 0.99f);
            mvar[c] = mvar[c] * 0.99f + unb * (1.f - 0.99f);
        }
    }
    const bool full = c0 + KC <= C;
#pragma unroll 4
    for (int m = rl; m < M; m += SM_ROWS) {
        float f[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            f[e] = f[e] * sc[e] + of[e];
            if (relu == 1) f[e] = fmaxf(f[e], 0.f);
            else if (relu == 2) f[e] = f[e] > 0.f ? f[e] : 0.1f * f[e];
        }
        TY* yp = y + out_off(m, rows_per_img, y_img_stride, ldy) + c0;
        if (full && vec_ok) {
            if (sizeof(TY) == 2) {
                float f8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f8[e] = f[e % KC];
                st16(reinterpret_cast<bf16_t*>(yp), Chunk<bf16_t>::pack(f8));   // only reached when KC == 8
            } else {
#pragma unroll
                for (int q = 0; q < KC / 4; ++q)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yp) + 4 * q) =
                        make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < KC; ++e)
                if (c0 + e < C) yp[e] = elem<TY>::store(f[e]);
        }
    }
}

template <typename T, typename TY, int CL>
__global__ void __launch_bounds__(512) bn_bwd_small_kernel(
    const T* __restrict__ z, const TY* __restrict__ y, const TY* __restrict__ dy, int M, int C, int ldz, int ldy,
    int rows_per_img, long long y_img_stride, const float* __restrict__ gamma, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, int relu, int vec_ok, T* __restrict__ dz, float* __restrict__ dgamma,
    float* __restrict__ dbeta) {
    constexpr int KC = Chunk<T>::N;
    constexpr int SM_ROWS = 512 / CL;
    __shared__ float sm[512 * 2 * KC];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c0 = (blockIdx.x * CL + cl) * KC;
    float acc[2 * KC], mu[KC], iv[KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        mu[e] = c0 + e < C ? save_mean[c0 + e] : 0.f;
        iv[e] = c0 + e < C ? save_invstd[c0 + e] : 0.f;
    }
    if (c0 < C) {
#pragma unroll 4
        for (int m = rl; m < M; m += SM_ROWS) {
            float f[KC], d[KC];
            Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
            bn_load_dy<T, TY>(y, dy, out_off(m, rows_per_img, y_img_stride, ldy) + c0, c0, C, relu, vec_ok, d);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                if (c0 + e >= C) continue;
                acc[e] += d[e];
                acc[KC + e] += d[e] * ((f[e] - mu[e]) * iv[e]);
            }
        }
    }
    block_rowlane_reduce_small<2 * KC, CL>(acc, sm, rl, cl);
    if (c0 >= ldz) return;
    float gs[KC], k1[KC], k2[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int c = c0 + e;
        gs[e] = 0.f; k1[e] = 0.f; k2[e] = 0.f;
        if (c >= C) continue;
        gs[e] = gamma[c] * iv[e];
        k1[e] = acc[e] / (float)M;
        k2[e] = acc[KC + e] / (float)M;
        if (rl == 0) { dbeta[c] = acc[e]; dgamma[c] = acc[KC + e]; }
    }
#pragma unroll 4
    for (int m = rl; m < M; m += SM_ROWS) {
        float f[KC], o[KC], d[KC];
        Chunk<T>::unpack(ld16(z + (size_t)m * ldz + c0), f);
        bn_load_dy<T, TY>(y, dy, out_off(m, rows_per_img, y_img_stride, ldy) + c0, c0, C, relu, vec_ok, d);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            o[e] = 0.f;
            if (c0 + e >= C) continue;
            const float xh = (f[e] - mu[e]) * iv[e];
            o[e] = gs[e] * (d[e] - k1[e] - xh * k2[e]);
        }
        st16(dz + (size_t)m * ldz + c0, Chunk<T>::pack(o));
    }
}

// generic column sum (bias gradient)
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, int M, int C, int ld,
                                                     int rows_per_split, float* __restrict__ ws) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * KC];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    const int split = blockIdx.y;
    float acc[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        const int m0 = split * rows_per_split;
        int m1 = m0 + rows_per_split; if (m1 > M) m1 = M;
        for (int m = m0 + rl; m < m1; m += RED_ROWS) {
            float f[KC];
            Chunk<T>::unpack(ld16(x + (size_t)m * ld + c0), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) acc[e] += f[e];
        }
    }
    block_rowlane_reduce<KC>(acc, sm, rl, cl);
    if (rl == 0 && c0 < C) {
#pragma unroll
        for (int e = 0; e < KC; ++e)
            if (c0 + e < C) ws[(size_t)split * C + c0 + e] = acc[e];
    }
}
// 32 columns x 8 row groups per workgroup: the partial rows are summed eight at a time (up to 1024 of them for narrow layers: one thread per column
// walked them as 1024 dependent loads, 45 us a launch in RetinaNet's step), then across the groups through LDS in a fixed order
__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ ws, int nsplit, int C, float* __restrict__ out,
                                                              int accumulate) {
    __shared__ float part[8][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C)
        for (int i = rg; i < nsplit; i += 8) s += ws[(size_t)i * C + c];
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = part[0][cl];
#pragma unroll
        for (int r = 1; r < 8; ++r) t += part[r][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ------------------------------------------------------------------ L2 normalise (one wave per row)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int C,
                                                         int ld, const float* __restrict__ gamma) {
    constexpr int KC = Chunk<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = C / KC;
    const float g = gamma[0];
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        float ss = 0.f;
        for (int ch = lane; ch < chunks; ch += 64) {
            float f[KC];
            Chunk<T>::unpack(ld16(x + (size_t)m * ld + ch * KC), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) ss += f[e] * f[e];
        }
        ss = wave_sum(ss);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        for (int ch = lane; ch < chunks; ch += 64) {
            float f[KC];
            Chunk<T>::unpack(ld16(x + (size_t)m * ld + ch * KC), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) f[e] = f[e] * inv * g;
            st16(y + (size_t)m * ld + ch * KC, Chunk<T>::pack(f));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         T* __restrict__ dx, int M, int C, int ld,
                                                         const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                         int accumulate, const T* __restrict__ relu_src, float* __restrict__ part) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sdg[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = C / KC;
    const float g = gamma[0];
    float dg = 0.f;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        float ss = 0.f, sd = 0.f;
        for (int ch = lane; ch < chunks; ch += 64) {
            float f[KC], d[KC];
            Chunk<T>::unpack(ld16(x + (size_t)m * ld + ch * KC), f);
            Chunk<T>::unpack(ld16(dy + (size_t)m * ld + ch * KC), d);
#pragma unroll
            for (int e = 0; e < KC; ++e) { ss += f[e] * f[e]; sd += f[e] * d[e]; }
        }
        ss = wave_sum(ss);
        sd = wave_sum(sd);
        const bool clamped = !(ss > 1e-12f);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        dg += sd * inv;
        const float k = clamped ? 0.f : g * inv * inv * inv * sd;
        for (int ch = lane; ch < chunks; ch += 64) {
            float f[KC], d[KC], o[KC];
            const size_t off = (size_t)m * ld + ch * KC;
            Chunk<T>::unpack(ld16(x + off), f);
            Chunk<T>::unpack(ld16(dy + off), d);
#pragma unroll
            for (int e = 0; e < KC; ++e) o[e] = g * inv * d[e] - k * f[e];
            if (accumulate) {
                float p[KC];
                Chunk<T>::unpack(ld16(dx + off), p);
#pragma unroll
                for (int e = 0; e < KC; ++e) o[e] += p[e];
            }
            if (relu_src) {
                float r[KC];
                Chunk<T>::unpack(ld16(relu_src + off), r);
#pragma unroll
                for (int e = 0; e < KC; ++e) if (!(r[e] > 0.f)) o[e] = 0.f;
            }
            st16(dx + off, Chunk<T>::pack(o));
        }
    }
    if (lane == 0) sdg[wave] = dg;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (sdg[0] + sdg[1]) + (sdg[2] + sdg[3]);
        if (part == nullptr) {
            atomicAdd(dgamma, v);
        } else {
            // deterministic mode (the default since round 6): this block's sum goes to part[block] with a plain store; l2norm_dgamma_reduce_kernel adds the blocks in
            // order.  (Round 5 had the last block to arrive do it behind a device-scope fence + ticket: 211 us per launch against 48 with the atomic -- the fence in
            // 1 024 workgroups costs more than the second launch; found in round 6's end-of-round trace.)
            part[blockIdx.x] = v;
        }
    }
}

// dgamma += part[0] + part[1] + ... in a fixed order: 256 threads take consecutive runs, the run sums meet in LDS and are added in thread order
__global__ void __launch_bounds__(256) l2norm_dgamma_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ dgamma) {
    __shared__ float sm[256];
    const int per = (n + 255) / 256, b0 = threadIdx.x * per;
    float t = 0.f;
    for (int i = b0; i < b0 + per && i < n; ++i) t += part[i];
    sm[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
        for (int i = 0; i < 256; ++i) sum += sm[i];
        *dgamma += sum;
    }
}

// ------------------------------------------------------------------ optimizer
constexpr int SGD_THREADS = 256;
constexpr int SGD_PER_BLOCK = SGD_THREADS * 4 * 8;   // 8 float4 per thread

template <typename TC>
__global__ void __launch_bounds__(SGD_THREADS) sgd_kernel(float* __restrict__ p, float* __restrict__ m,
                                                          const float* __restrict__ g, long long n, float lr,
                                                          float mom, float wd, float gscale,
                                                          float* __restrict__ l2_partial, TC* __restrict__ pc) {
    __shared__ float sm[SGD_THREADS / 64];
    const long long base = (long long)blockIdx.x * SGD_PER_BLOCK;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long long i = base + ((long long)it * SGD_THREADS + threadIdx.x) * 4;
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            ss += pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
            mv.x = mom * mv.x + (gv.x * gscale + wd * pv.x); pv.x -= lr * mv.x;
            mv.y = mom * mv.y + (gv.y * gscale + wd * pv.y); pv.y -= lr * mv.y;
            mv.z = mom * mv.z + (gv.z * gscale + wd * pv.z); pv.z -= lr * mv.z;
            mv.w = mom * mv.w + (gv.w * gscale + wd * pv.w); pv.w -= lr * mv.w;
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            if (pc) {
                pc[i] = elem<TC>::store(pv.x); pc[i + 1] = elem<TC>::store(pv.y);
                pc[i + 2] = elem<TC>::store(pv.z); pc[i + 3] = elem<TC>::store(pv.w);
            }
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) {
                float pv = p[j], mv = m[j];
                ss += pv * pv;
                mv = mom * mv + (g[j] * gscale + wd * pv);
                pv -= lr * mv;
                p[j] = pv; m[j] = mv;
                if (pc) pc[j] = elem<TC>::store(pv);
            }
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0 && l2_partial) l2_partial[blockIdx.x] = 0.5f * ((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

__global__ void __launch_bounds__(1024) sum_kernel(const float* __restrict__ in, long long n, float* __restrict__ out) {
    __shared__ float sm[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += in[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += sm[i];
        out[0] = t;
    }
}

// the step's reported loss in ONE launch: sum_a = sum_i a[i * stride_a], sum_b = sum_j b[j], total = scale_a * sum_a + scale_b * sum_b (fixed order)
__global__ void __launch_bounds__(1024) loss_total_kernel(const float* __restrict__ a, int na, int stride_a, const float* __restrict__ b, long long nb,
                                                          float scale_a, float scale_b, float* __restrict__ sum_a, float* __restrict__ sum_b,
                                                          float* __restrict__ total) {
    __shared__ float sm[32];
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x; i < na; i += 1024) sa += a[(size_t)i * stride_a];
    for (long long i = threadIdx.x; i < nb; i += 1024) sb += b[i];
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = sa; sm[16 + (threadIdx.x >> 6)] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int i = 0; i < 16; ++i) { ta += sm[i]; tb += sm[16 + i]; }
        if (sum_a) sum_a[0] = ta;
        if (sum_b) sum_b[0] = tb;
        total[0] = scale_a * ta + scale_b * tb;
    }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = elem<T>::store(in[i]);
}

// 8 elements per thread and pass: two float4 <-> one 16-byte bf16 chunk (the gradient buckets of the data-parallel path: 26 M elements)
__global__ void cast_f32_to_bf16_x8_kernel(const float4* __restrict__ in, uint4* __restrict__ out, long long n8) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n8; i += step) {
        const float4 a = in[2 * i], b = in[2 * i + 1];
        uint4 o;
        o.x = (unsigned)f32_to_bf16(a.x) | ((unsigned)f32_to_bf16(a.y) << 16);
        o.y = (unsigned)f32_to_bf16(a.z) | ((unsigned)f32_to_bf16(a.w) << 16);
        o.z = (unsigned)f32_to_bf16(b.x) | ((unsigned)f32_to_bf16(b.y) << 16);
        o.w = (unsigned)f32_to_bf16(b.z) | ((unsigned)f32_to_bf16(b.w) << 16);
        out[i] = o;
    }
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = elem<T>::load(in[i]);
}
__global__ void cast_bf16_to_f32_x8_kernel(const uint4* __restrict__ in, float4* __restrict__ out, long long n8) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n8; i += step) {
        const uint4 v = in[i];
        out[2 * i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                                 __uint_as_float(v.y & 0xffff0000u));
        out[2 * i + 1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16),
                                     __uint_as_float(v.w & 0xffff0000u));
    }
}

inline int grid_for(long long total, int threads, int cap = 8192) {
    long long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// rows of partials the batch-norm workspace holds per sum: 256 at >= 256 channels, more below (the same 2 x 256 x 256 floats)
inline int bn_ws_rows(int C) {
    const int cpad = (C + 63) / 64 * 64;
    return cpad < 256 ? 256 * (256 / cpad) : 256;
}
struct RedPlan { int colgroups, nsplit, rows_per_split; };
inline RedPlan red_plan(int M, int C, int kc) {
    RedPlan p;
    p.colgroups = ceil_div(C, 8 * kc);
    int want = 1024 / p.colgroups;
    if (want < 1) want = 1;
    int maxs = ceil_div(M, 4 * RED_ROWS);
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    // the partials live in the workspace: 2 x bn_ws_rows(C) x ceil64(C) floats (odtk_bn_workspace_bytes) = 256 splits from 256 channels on, up to 1 024 for
    // narrow layers (round 5: with ONE column group 256 splits are one workgroup per CU -- too few loads in flight to stream; Darknet's and DLA's first layers
    // are 1.4 - 4.2 M rows x 16 - 64 channels)
    int cap = bn_ws_rows(C) * (((C + 63) / 64) * 64) / C;
    if (cap > 1024) cap = 1024;
    if (want > cap) want = cap;
    p.rows_per_split = ceil_div(M, want);
    p.nsplit = ceil_div(M, p.rows_per_split);
    return p;
}
// the same plan with at most `maxsplit` row splits: what the finalize-in-apply launches use (every apply workgroup re-reduces the partials of its 64
// channels as a prologue: 32 splits = 16 KB from L2, ~1.5 us; the 169-256 splits of the plain plan made that path SLOWER than three launches)
inline RedPlan red_plan_capped(int M, int C, int kc, int maxsplit) {
    RedPlan p = red_plan(M, C, kc);
    if (p.nsplit > maxsplit) {
        p.rows_per_split = ceil_div(M, maxsplit);
        p.nsplit = ceil_div(M, p.rows_per_split);
    }
    return p;
}

}  // namespace
}  // namespace odtk

using namespace odtk;
namespace odtk { namespace cv { bool get_wgrad_deterministic(); int get_scratch_slot(); } }

static int g_bn_small_rows = 1024;        // odtk_debug_set key 4 (value >= 0).  (1 408, so that YOLOv3's 13 x 13 maps at 8 images -- 1 352 rows x 512 / 1 024 channels --
                                          // take the single launch, was measured SLOWER there: 11.24 vs 11.13 ms/step, gpurun r03o)
static bool g_bn_auto_two = true;         // odtk_debug_set key 4, value -5: never pick the two-launch path by itself (round-2 behaviour; A/B); -6: back
static bool g_bn_small_wide = false;      // odtk_debug_set key 4, value -3: the single-launch kernels in their 64-channel shape only (A/B); -4: back
static int g_bn_rpb = 0;                  // rows per workgroup of the apply passes; odtk_debug_set key 4, values -20 .. -23: 128 / 512 / 1 024 / by shape (A/B)
// 256 rows per workgroup of an apply pass; 1 024 for the 8- / 16-channel bf16 maps, whose blocks are 256 / 128 row lanes (one or two rows per lane and
// workgroup otherwise: DLA-34's 512 x 512 x 16 backward 238 -> 199 us, forward 86 -> 81; every other shape is flat from 128 to 1 024, gpurun r05n)
static inline int bn_rpb(int C, int ldz, int dtype) {
    if (g_bn_rpb) return g_bn_rpb;
    return (dtype == ODTK_BF16 && ldz == C && (C == 8 || C == 16)) ? 1024 : 256;
}
static bool g_bn_three_kernels = true;    // odtk_debug_set key 4, value -2: statistics + apply-with-finalize (two launches; A/B, tests); -1: back to three
namespace odtk { void set_bn_small_rows(int rows) { if (rows <= -20 && rows >= -23) g_bn_rpb = rows == -20 ? 128 : rows == -21 ? 512 : rows == -22 ? 1024 : 0; else if (rows == -1) g_bn_three_kernels = true; else if (rows == -2) g_bn_three_kernels = false; else if (rows == -3) g_bn_small_wide = true; else if (rows == -4) g_bn_small_wide = false; else if (rows == -5) g_bn_auto_two = false; else if (rows == -6) g_bn_auto_two = true; else g_bn_small_rows = rows; } }

#define DT_SWITCH(dtype, T, ...)                                         \
    if ((dtype) == ODTK_BF16) { typedef bf16_t T; __VA_ARGS__ }          \
    else if ((dtype) == ODTK_F32) { typedef float T; __VA_ARGS__ }       \
    else { set_error("bad dtype %d", (int)(dtype)); return ODTK_ERR_ARG; }

extern "C" int odtk_preprocess(const float* images, long long pixels, const float* mean3, int ldx, int dtype,
                               void* x, void* stream) {
    ODTK_REQUIRE(images && x && mean3 && ldx >= 3, "preprocess: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_BF16 && ldx == 8 && ((uintptr_t)images & 15) == 0 && ((uintptr_t)x & 15) == 0) {
        hipLaunchKernelGGL(preprocess_bf16x8_kernel, dim3(grid_for((pixels + 3) / 4, 256)), dim3(256), 0, st, images, pixels, mean3[0], mean3[1], mean3[2],
                           (bf16_t*)x);
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(preprocess_kernel<T>, dim3(grid_for(pixels, 256)), dim3(256), 0, st,
                                           images, pixels, mean3[0], mean3[1], mean3[2], ldx, (T*)x);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

static int pool_check(int C, int ld, int dtype) {
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(C % kc == 0 && ld % kc == 0, "pool: C=%d ld=%d must be multiples of %d", C, ld, kc);
    return ODTK_OK;
}

extern "C" int odtk_maxpool_fwd(const void* x, void* y, int N, int H, int W, int C, int ld, int Ho, int Wo,
                                int k, int stride, int pad_t, int pad_l, int dtype, void* stream) {
    ODTK_REQUIRE(x && y, "maxpool_fwd: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    const long long total = (long long)N * Ho * Wo * (C / kc);
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                                           (const T*)x, (T*)y, N, H, W, C, ld, Ho, Wo, k, stride, pad_t, pad_l);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_maxpool_bwd(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W,
                                int C, int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                                int dtype, void* stream) {
    ODTK_REQUIRE(x && y && dy && dx, "maxpool_bwd: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    if (k == 2 && stride == 2 && pad_t == 0 && pad_l == 0 && (long long)N * H * W * (C / kc) < (1ll << 31)) {
        const long long tot_o = (long long)N * Ho * Wo * (C / kc);
        DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool2x2_bwd_kernel<T>, dim3(grid_for(tot_o, 256, 65536)), dim3(256), 0, st,
                                               (const T*)x, (const T*)dy, (T*)dx, N, H, W, C, ld, Ho, Wo);)
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    const long long total = (long long)N * H * W * (C / kc);
    ODTK_REQUIRE(total < (1ll << 31), "maxpool_bwd: tensor too large (%lld chunks)", total);
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                                           (const T*)x, (const T*)y, (const T*)dy, (T*)dx, N, H, W, C, ld, Ho, Wo, k,
                                           stride, pad_t, pad_l);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_maxpool_fwd_argmax(const void* x, void* y, void* arg, int N, int H, int W, int C, int ld, int Ho, int Wo, int k, int stride,
                                       int pad_t, int pad_l, int dtype, void* stream) {
    ODTK_REQUIRE(x && y && arg, "maxpool_fwd_argmax: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(k >= 1 && k <= 3 && stride >= 1, "maxpool_fwd_argmax: window %d unsupported (4-bit position codes: k <= 3)", k);
    const long long tot_o = (long long)N * Ho * Wo * (C / kc);
    ODTK_REQUIRE(tot_o < (1ll << 31) && (long long)N * H * W * (C / kc) < (1ll << 31), "maxpool_fwd_argmax: tensor too large");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_fwd_argmax_kernel<T>, dim3(grid_for(tot_o, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)x, (T*)y, (unsigned*)arg, N, H, W, C, ld, Ho, Wo, k, stride, pad_t, pad_l);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_maxpool_bwd_argmax(const void* arg, const void* dy, void* dx, int N, int H, int W, int C, int ld, int Ho, int Wo, int k,
                                       int stride, int pad_t, int pad_l, int dtype, void* stream) {
    ODTK_REQUIRE(arg && dy && dx, "maxpool_bwd_argmax: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(k >= 1 && k <= 3 && stride >= 1, "maxpool_bwd_argmax: window %d unsupported", k);
    const long long tot = (long long)N * H * W * (C / kc);
    ODTK_REQUIRE(tot < (1ll << 31) && (long long)N * Ho * Wo * (C / kc) < (1ll << 31), "maxpool_bwd_argmax: tensor too large");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_bwd_argmax_kernel<T>, dim3(grid_for(tot, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const unsigned*)arg, (const T*)dy, (T*)dx, N, H, W, C, ld, Ho, Wo, k, stride, pad_t, pad_l);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_maxpool2x2_fwd_idx(const void* x, void* y, void* idx, int N, int H, int W, int C, int ld, int Ho, int Wo,
                                       int dtype, void* stream) {
    ODTK_REQUIRE(x && y && idx, "maxpool2x2_fwd_idx: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    const long long tot_o = (long long)N * Ho * Wo * (C / kc);
    ODTK_REQUIRE(Ho == (H + 1) / 2 && Wo == (W + 1) / 2 && (long long)N * H * W * (C / kc) < (1ll << 31), "maxpool2x2_fwd_idx: bad geometry");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool2x2_fwd_idx_kernel<T>, dim3(grid_for(tot_o, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)x, (T*)y, (unsigned short*)idx, N, H, W, C, ld, Ho, Wo);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_maxpool2x2_bwd_idx(const void* idx, const void* dy, void* dx, int N, int H, int W, int C, int ld, int Ho,
                                       int Wo, int dtype, void* stream) {
    ODTK_REQUIRE(idx && dy && dx, "maxpool2x2_bwd_idx: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    const long long tot_o = (long long)N * Ho * Wo * (C / kc);
    ODTK_REQUIRE(Ho == (H + 1) / 2 && Wo == (W + 1) / 2 && (long long)N * H * W * (C / kc) < (1ll << 31), "maxpool2x2_bwd_idx: bad geometry");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool2x2_bwd_idx_kernel<T>, dim3(grid_for(tot_o, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const unsigned short*)idx, (const T*)dy, (T*)dx, N, H, W, C, ld, Ho, Wo);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" long long odtk_bn_workspace_bytes(int M, int C) {
    (void)M;
    return (long long)(2 * bn_ws_rows(C) + 2) * (long long)((C + 63) / 64 * 64) * sizeof(float);
}

extern "C" int odtk_bn_fwd(const void* z, int M, int C, int ldz, int dtype, const float* gamma,
                           const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                           float* save_invstd, int training, int relu, void* y, int y_dtype, int ldy,
                           int rows_per_img, long long y_img_stride, void* workspace, void* stream) {
    ODTK_REQUIRE(z && y && gamma && beta && moving_mean && moving_var, "bn_fwd: null pointer");
    ODTK_REQUIRE(workspace, "bn_fwd: workspace required");
    ODTK_REQUIRE(!training || (save_mean && save_invstd), "bn_fwd: training needs save buffers");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_fwd: ldz=%d must be a multiple of %d", ldz, kc);
    ODTK_REQUIRE(!(y_dtype == ODTK_BF16 && dtype == ODTK_F32), "bn_fwd: f32 in / bf16 out unsupported");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    const size_t ysz = dtype_size(y_dtype);
    const int vec_ok = ((size_t)ldy * ysz) % 16 == 0 && ((size_t)y_img_stride * ysz) % 16 == 0 &&
                       ((uintptr_t)y % 16) == 0;
    if (training && M <= g_bn_small_rows) {              // small map: statistics, finalize and apply in one launch
        // few column groups (C <= 512): one workgroup per 16-byte channel chunk, 512 row lanes (odtk_debug_set key 4, value -3 / -4: the 64-channel shape, A/B)
#define BN_SMALL(T, TY)                                                                                                      \
    do { if (!g_bn_small_wide)                                                                          \
        hipLaunchKernelGGL((bn_fwd_small_kernel<T, TY, 1>), dim3(ceil_div(C, kc)), dim3(512), 0, st, (const T*)z, M, C, ldz, gamma, beta, \
                           moving_mean, moving_var, save_mean, save_invstd, relu, (TY*)y, ldy, rows_per_img, y_img_stride, vec_ok); \
    else                                                                                                                     \
        hipLaunchKernelGGL((bn_fwd_small_kernel<T, TY, 8>), dim3(pl.colgroups), dim3(512), 0, st, (const T*)z, M, C, ldz, gamma, beta, \
                           moving_mean, moving_var, save_mean, save_invstd, relu, (TY*)y, ldy, rows_per_img, y_img_stride, vec_ok); } while (0)
        if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) BN_SMALL(bf16_t, bf16_t);
        else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) BN_SMALL(bf16_t, float);
        else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) BN_SMALL(float, float);
        else ODTK_REQUIRE(false, "bn_fwd: bad dtype");
#undef BN_SMALL
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    float* ws = (float*)workspace;
    float* fin = ws + (size_t)2 * bn_ws_rows(C) * ((C + 63) / 64 * 64);
    if (training) {
        // Two launches (statistics; apply with the finalize folded into its prologue) where <= 32 row splits still fill the chip for the statistics pass
        // (>= 128 workgroups: the 256-1 024-channel layers of DarkNet-53 at 8 images: the finalize launch alone was 8-10 us of latency, x 150 per step)
        const RedPlan plc = red_plan_capped(M, C, kc, 32);
        // Measured (gpurun r03o): SSD300's conv6 / conv7 (11 552 rows x 1 024 channels) gain 0.6 % of the step; DarkNet-53's 5 408 x 512 and 21 632 x 256 maps at
        // 8 images LOSE 1 % of theirs -- hence the size condition on top.
        const bool two = !g_bn_three_kernels || (g_bn_auto_two && plc.colgroups * plc.nsplit >= 128 && M >= 8192 && C >= 512);
        const RedPlan pl = two ? plc : red_plan(M, C, kc);
        DT_SWITCH(dtype, T, hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(pl.colgroups, pl.nsplit), dim3(256), 0, st,
                                               (const T*)z, M, C, ldz, pl.rows_per_split, ws);)
        if (two) {                                       // statistics, then apply with the finalize folded in
            const int rpb = pl.rows_per_split < bn_rpb(C, ldz, dtype) ? pl.rows_per_split : bn_rpb(C, ldz, dtype);
            dim3 gridf(pl.colgroups, ceil_div(M, rpb));
#define BN_APPLY_FIN(T, TY)                                                                                                  \
    hipLaunchKernelGGL((bn_apply_fin_kernel<T, TY>), gridf, dim3(256), 0, st, (const T*)z, M, C, ldz, relu, (TY*)y, ldy,     \
                       rows_per_img, y_img_stride, vec_ok, gamma, beta, moving_mean, moving_var, save_mean, save_invstd, ws, \
                       pl.nsplit, rpb)
            if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) BN_APPLY_FIN(bf16_t, bf16_t);
            else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) BN_APPLY_FIN(bf16_t, float);
            else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) BN_APPLY_FIN(float, float);
            else ODTK_REQUIRE(false, "bn_fwd: bad dtype");
#undef BN_APPLY_FIN
            ODTK_LAUNCH_CHECK();
            return ODTK_OK;
        }
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, st, (const T*)z, M,
                                           C, gamma, beta, moving_mean, moving_var, save_mean, save_invstd, training, ws,
                                           pl.nsplit, fin);)
    // the apply pass is elementwise: many short workgroups keep more loads in flight than one long one per CU (the statistics
    // pass keeps <= 256 row splits because its partials live in the workspace)
    const int rows_per_block = pl.rows_per_split < bn_rpb(C, ldz, dtype) ? pl.rows_per_split : bn_rpb(C, ldz, dtype);
    dim3 grid(pl.colgroups, ceil_div(M, rows_per_block));
#define BN_APPLY(T, TY)                                                                                        \
    hipLaunchKernelGGL((bn_apply_kernel<T, TY>), grid, dim3(256), 0, st, (const T*)z, M, C, ldz, relu, (TY*)y, \
                       ldy, rows_per_img, y_img_stride, vec_ok, fin, rows_per_block)
    if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) BN_APPLY(bf16_t, bf16_t);
    else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) BN_APPLY(bf16_t, float);
    else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) BN_APPLY(float, float);
    else ODTK_REQUIRE(false, "bn_fwd: bad dtype");
#undef BN_APPLY
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_bn_bwd(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype,
                           int y_dtype, int ldy, int rows_per_img, long long y_img_stride,
                           const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                           void* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
    ODTK_REQUIRE(z && dy && dz && gamma && save_mean && save_invstd && dgamma && dbeta && workspace,
                 "bn_bwd: null pointer");
    ODTK_REQUIRE(!relu || y, "bn_bwd: relu needs y");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_bwd: ldz=%d must be a multiple of %d", ldz, kc);
    hipStream_t st = (hipStream_t)stream;
    const RedPlan plc = red_plan_capped(M, C, kc, 32);   // (see odtk_bn_fwd: two launches where <= 32 row splits still fill the chip)
    const bool two = !g_bn_three_kernels || (g_bn_auto_two && plc.colgroups * plc.nsplit >= 128 && M >= 8192 && C >= 512);
    const RedPlan pl = two ? plc : red_plan(M, C, kc);
    float* ws = (float*)workspace;
    // the apply pass is elementwise: many short workgroups keep more loads in flight than one long one per CU (the statistics
    // pass keeps <= 256 row splits because its partials live in the workspace)
    const int rows_per_block = pl.rows_per_split < bn_rpb(C, ldz, dtype) ? pl.rows_per_split : bn_rpb(C, ldz, dtype);
    float* fin = ws + (size_t)2 * bn_ws_rows(C) * ((C + 63) / 64 * 64);
    dim3 g1(pl.colgroups, pl.nsplit);
    dim3 g2(ceil_div(ldz, 8 * kc), ceil_div(M, rows_per_block));
    const size_t ysz = y_dtype == ODTK_BF16 ? 2 : 4;
    const int vec_ok = ((size_t)ldy * ysz) % 16 == 0 && ((size_t)y_img_stride * ysz) % 16 == 0 &&
                       ((uintptr_t)dy) % 16 == 0 && (!relu || ((uintptr_t)y) % 16 == 0);
    if (M <= g_bn_small_rows) {                          // small map: sums, finalize and apply in one launch
#define BN_BWD_SMALL(T, TY)                                                                                                   \
    do { if (!g_bn_small_wide)                                                                           \
        hipLaunchKernelGGL((bn_bwd_small_kernel<T, TY, 1>), dim3(ceil_div(ldz, kc)), dim3(512), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, \
                           C, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, vec_ok, (T*)dz, dgamma, dbeta); \
    else                                                                                                                      \
        hipLaunchKernelGGL((bn_bwd_small_kernel<T, TY, 8>), dim3(g2.x), dim3(512), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, \
                           C, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, vec_ok, (T*)dz, dgamma, dbeta); } while (0)
        if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) { BN_BWD_SMALL(bf16_t, bf16_t); }
        else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) { BN_BWD_SMALL(bf16_t, float); }
        else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) { BN_BWD_SMALL(float, float); }
        else ODTK_REQUIRE(false, "bn_bwd: bad dtype");
#undef BN_BWD_SMALL
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (two) {                                           // sums, then apply with the finalize folded in
#define BN_BWD_FIN(T, TY)                                                                                                       \
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, TY>), g1, dim3(256), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, C, ldz,   \
                       ldy, rows_per_img, y_img_stride, save_mean, save_invstd, relu, vec_ok, pl.rows_per_split, ws);             \
    hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<T, TY>), g2, dim3(256), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, C,    \
                       ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, vec_ok, (T*)dz, ws, pl.nsplit,  \
                       dgamma, dbeta, rows_per_block)
        if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) { BN_BWD_FIN(bf16_t, bf16_t); }
        else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) { BN_BWD_FIN(bf16_t, float); }
        else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) { BN_BWD_FIN(float, float); }
        else ODTK_REQUIRE(false, "bn_bwd: bad dtype");
#undef BN_BWD_FIN
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
#define BN_BWD(T, TY)                                                                                             \
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, TY>), g1, dim3(256), 0, st, (const T*)z, (const TY*)y,             \
                       (const TY*)dy, M, C, ldz, ldy, rows_per_img, y_img_stride, save_mean, save_invstd, relu,   \
                       vec_ok, pl.rows_per_split, ws);                                                            \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, st, ws, pl.nsplit, C, M,     \
                       dgamma, dbeta, fin);                                                                       \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TY>), g2, dim3(256), 0, st, (const T*)z, (const TY*)y,             \
                       (const TY*)dy, M, C, ldz, ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd,  \
                       relu, vec_ok, (T*)dz, fin, rows_per_block)
    if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) { BN_BWD(bf16_t, bf16_t); }
    else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) { BN_BWD(bf16_t, float); }
    else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) { BN_BWD(float, float); }
    else ODTK_REQUIRE(false, "bn_bwd: bad dtype");
#undef BN_BWD
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_l2norm_fwd(const void* x, void* y, int M, int C, int ld, int dtype, const float* gamma,
                               void* stream) {
    ODTK_REQUIRE(x && y && gamma, "l2norm_fwd: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    hipStream_t st = (hipStream_t)stream;
    int grid = ceil_div(M, 4); if (grid > 4096) grid = 4096;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(l2norm_fwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)x, (T*)y, M, C,
                                           ld, gamma);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_l2norm_bwd(const void* x, const void* dy, void* dx, int M, int C, int ld, int dtype,
                               const float* gamma, float* dgamma, int accumulate, const void* relu_src,
                               void* stream) {
    ODTK_REQUIRE(x && dy && dx && gamma && dgamma, "l2norm_bwd: null pointer");
    if (int e = pool_check(C, ld, dtype)) return e;
    hipStream_t st = (hipStream_t)stream;
    int grid = ceil_div(M, 4); if (grid > 1024) grid = 1024;
    // deterministic mode (key 5, the default): per-block partial sums in a small buffer per (device, scratch slot), added in block order by a second, tiny launch --
    // like the convolutions' scratch, launches that are in flight at the same time must come from threads on different slots (odtk_scratch_slot; round-5
    // advisory).  The buffer is allocated on first use (not inside a stream capture: run one eager step first).
    float* part = nullptr;
    if (cv::get_wgrad_deterministic()) {
        static float* s_part[16][4] = {{nullptr}};
        static std::mutex s_mutex;
        int dev = 0;
        ODTK_CHECK_HIP(hipGetDevice(&dev));
        ODTK_REQUIRE(dev >= 0 && dev < 16, "l2norm_bwd: device index %d unsupported", dev);
        const int slot = cv::get_scratch_slot() & 3;
        std::lock_guard<std::mutex> lock(s_mutex);
        if (!s_part[dev][slot]) {
            void* p = nullptr;                       // (inside a stream capture hipMalloc fails and says so: run one eager step first)
            ODTK_CHECK_HIP(hipMalloc(&p, 1024 * sizeof(float)));
            s_part[dev][slot] = (float*)p;
        }
        part = s_part[dev][slot];
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(l2norm_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)x, (const T*)dy,
                                           (T*)dx, M, C, ld, gamma, dgamma, accumulate, (const T*)relu_src, part);)
    if (part) hipLaunchKernelGGL(l2norm_dgamma_reduce_kernel, dim3(1), dim3(256), 0, st, part, grid, dgamma);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_colsum(const void* dy, int M, int C, int ld, int dtype, float* out, int accumulate,
                           void* workspace, void* stream) {
    ODTK_REQUIRE(dy && out && workspace, "colsum: null pointer");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ld % kc == 0 && ld >= C, "colsum: ld=%d must be a multiple of %d", ld, kc);
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    float* ws = (float*)workspace;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, dim3(pl.colgroups, pl.nsplit), dim3(256), 0, st,
                                           (const T*)dy, M, C, ld, pl.rows_per_split, ws);)
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, st, ws, pl.nsplit, C, out,
                       accumulate);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_sgd_blocks(long long n) { return (int)((n + SGD_PER_BLOCK - 1) / SGD_PER_BLOCK); }

extern "C" int odtk_sgd_momentum(float* p, float* m, const float* grad, long long n, float lr, float momentum,
                                 float wd, float grad_scale, float* l2_partial, void* p_cast, int cast_dtype,
                                 void* stream) {
    ODTK_REQUIRE(p && m && grad && n > 0, "sgd: bad argument");
    ODTK_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)grad % 16) == 0,
                 "sgd: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = odtk_sgd_blocks(n);
    if (p_cast && cast_dtype == ODTK_F32)
        hipLaunchKernelGGL(sgd_kernel<float>, dim3(blocks), dim3(SGD_THREADS), 0, st, p, m, grad, n, lr, momentum, wd,
                           grad_scale, l2_partial, (float*)p_cast);
    else
        hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(blocks), dim3(SGD_THREADS), 0, st, p, m, grad, n, lr, momentum, wd,
                           grad_scale, l2_partial, (bf16_t*)p_cast);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_sum_f32(const float* in, long long n, float* out, void* stream) {
    ODTK_REQUIRE(in && out && n >= 0, "sum: bad argument");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, n, out);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_loss_total(const float* a, int na, int stride_a, const float* b, long long nb, float scale_a, float scale_b, float* sum_a,
                               float* sum_b, float* total, void* stream) {
    ODTK_REQUIRE(a && b && total && na >= 0 && nb >= 0 && stride_a >= 1, "loss_total: bad argument");
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, na, stride_a, b, nb, scale_a, scale_b, sum_a, sum_b, total);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_cast_from_f32(const float* in, void* out, long long n, int dtype, void* stream) {
    ODTK_REQUIRE(in && out, "cast: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_BF16 && n >= 8 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        const long long n8 = n / 8;
        hipLaunchKernelGGL(cast_f32_to_bf16_x8_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const float4*)in, (uint4*)out, n8);
        if (n % 8)
            hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(1), dim3(64), 0, st, in + n8 * 8, (bf16_t*)out + n8 * 8, n % 8);
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(cast_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, in, (T*)out, n);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_cast_to_f32(const void* in, int dtype, float* out, long long n, void* stream) {
    ODTK_REQUIRE(in && out && n >= 0, "cast: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return ODTK_OK;
    if (dtype == ODTK_BF16 && n >= 8 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        const long long n8 = n / 8;
        hipLaunchKernelGGL(cast_bf16_to_f32_x8_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const uint4*)in, (float4*)out, n8);
        if (n % 8)
            hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(1), dim3(64), 0, st, (const bf16_t*)in + n8 * 8, out + n8 * 8, n % 8);
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(cast_to_f32_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const T*)in, out, n);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Batch norm over the GLOBAL batch of a data-parallel job (SURVEY.md 8e option B, "sync-BN"): the single-call entry points
// above split at the two places where the replicas have to exchange per-channel numbers.
//   forward : odtk_bn_moments (local mean / biased variance)  -> all-gather of [2C] per rank (host, RCCL)
//             -> odtk_bn_fwd_given (combines the replicas' moments with the parallel-variance formula, then apply)
//   backward: odtk_bn_bwd_sums (local sum dy', sum dy' * xhat = this replica's dbeta / dgamma) -> all-reduce of [2C]
//             -> odtk_bn_bwd_given (dz with the global means)
// With equal local batches the result equals one device running the concatenated batch (tests/test_gpu_dist.py).
namespace odtk {
namespace {

template <typename T>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_moments_kernel(const T* __restrict__ z, int M, int C, const float* __restrict__ ws, int nsplit,
                                                         float* __restrict__ mean, float* __restrict__ var) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), sl = threadIdx.x / FIN_CH;
    float s1, s2;
    reduce_partials(ws, nsplit, C, c, sl, s1, s2);
    if (c >= C || sl != 0) return;
    const float d = s1 / (float)M;
    mean[c] = elem<T>::load(z[c]) + d;
    var[c] = fmaxf(s2 / (float)M - d * d, 0.f);
}

// moments [W][2][C] of W replicas with `rows` rows each -> statistics of the union, scale / offset, moving-statistics update
__global__ void bn_combine_kernel(const float* __restrict__ moments, int W, int C, long long rows, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ mmean, float* __restrict__ mvar,
                                  float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean = 0.f;
    for (int w = 0; w < W; ++w) mean += moments[((size_t)w * 2 + 0) * C + c];
    mean /= (float)W;
    float var = 0.f;
    for (int w = 0; w < W; ++w) {
        const float dm = moments[((size_t)w * 2 + 0) * C + c] - mean;
        var += moments[((size_t)w * 2 + 1) * C + c] + dm * dm;
    }
    var /= (float)W;
    const float n = (float)rows * (float)W;
    save_mean[c] = mean;
    save_invstd[c] = rsqrtf(var + 1e-3f);
    mmean[c] = mmean[c] * 0.99f + mean * (1.f - 0.99f);
    mvar[c] = mvar[c] * 0.99f + var * (n / (n > 1.f ? n - 1.f : 1.f)) * (1.f - 0.99f);
    const float sc = rsqrtf(var + 1e-3f) * gamma[c];
    fin[c] = sc;
    fin[C + c] = beta[c] - mean * sc;
}

__global__ void bn_scale_sums_kernel(const float* __restrict__ sums, int n, float inv_count, float* __restrict__ fin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fin[i] = sums[i] * inv_count;
}

}  // namespace
}  // namespace odtk

extern "C" int odtk_bn_moments(const void* z, int M, int C, int ldz, int dtype, float* mean, float* var, void* workspace, void* stream) {
    ODTK_REQUIRE(z && mean && var && workspace, "bn_moments: null pointer");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_moments: ldz=%d must be a multiple of %d", ldz, kc);
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    float* ws = (float*)workspace;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(pl.colgroups, pl.nsplit), dim3(256), 0, st, (const T*)z, M, C, ldz,
                                           pl.rows_per_split, ws);
              hipLaunchKernelGGL(bn_moments_kernel<T>, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, st, (const T*)z, M, C, ws, pl.nsplit, mean, var);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_bn_fwd_given(const void* z, int M, int C, int ldz, int dtype, const float* gamma, const float* beta,
                                 const float* moments, int replicas, float* moving_mean, float* moving_var, float* save_mean,
                                 float* save_invstd, int relu, void* y, int y_dtype, int ldy, int rows_per_img, long long y_img_stride,
                                 void* workspace, void* stream) {
    ODTK_REQUIRE(z && y && gamma && beta && moments && moving_mean && moving_var && save_mean && save_invstd && workspace, "bn_fwd_given: null pointer");
    ODTK_REQUIRE(replicas >= 1, "bn_fwd_given: replicas=%d", replicas);
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_fwd_given: ldz=%d must be a multiple of %d", ldz, kc);
    ODTK_REQUIRE(!(y_dtype == ODTK_BF16 && dtype == ODTK_F32), "bn_fwd_given: f32 in / bf16 out unsupported");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    float* ws = (float*)workspace;
    float* fin = ws + (size_t)2 * bn_ws_rows(C) * ((C + 63) / 64 * 64);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, moments, replicas, C, (long long)M, gamma, beta, moving_mean,
                       moving_var, save_mean, save_invstd, fin);
    const size_t ysz = dtype_size(y_dtype);
    const int vec_ok = ((size_t)ldy * ysz) % 16 == 0 && ((size_t)y_img_stride * ysz) % 16 == 0 && ((uintptr_t)y % 16) == 0;
    // the apply pass is elementwise: many short workgroups keep more loads in flight than one long one per CU (the statistics
    // pass keeps <= 256 row splits because its partials live in the workspace)
    const int rows_per_block = pl.rows_per_split < bn_rpb(C, ldz, dtype) ? pl.rows_per_split : bn_rpb(C, ldz, dtype);
    dim3 grid(pl.colgroups, ceil_div(M, rows_per_block));
#define BN_APPLY(T, TY)                                                                                        \
    hipLaunchKernelGGL((bn_apply_kernel<T, TY>), grid, dim3(256), 0, st, (const T*)z, M, C, ldz, relu, (TY*)y, \
                       ldy, rows_per_img, y_img_stride, vec_ok, fin, rows_per_block)
    if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) BN_APPLY(bf16_t, bf16_t);
    else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) BN_APPLY(bf16_t, float);
    else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) BN_APPLY(float, float);
    else ODTK_REQUIRE(false, "bn_fwd_given: bad dtype");
#undef BN_APPLY
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_bn_bwd_sums(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype, int y_dtype, int ldy,
                                int rows_per_img, long long y_img_stride, const float* save_mean, const float* save_invstd, int relu,
                                float* sums, void* workspace, void* stream) {
    ODTK_REQUIRE(z && dy && save_mean && save_invstd && sums && workspace, "bn_bwd_sums: null pointer");
    ODTK_REQUIRE(!relu || y, "bn_bwd_sums: relu needs y");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_bwd_sums: ldz=%d must be a multiple of %d", ldz, kc);
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    float* ws = (float*)workspace;
    float* fin = ws + (size_t)2 * bn_ws_rows(C) * ((C + 63) / 64 * 64);
    dim3 g1(pl.colgroups, pl.nsplit);
    const size_t ysz = y_dtype == ODTK_BF16 ? 2 : 4;
    const int vec_ok = ((size_t)ldy * ysz) % 16 == 0 && ((size_t)y_img_stride * ysz) % 16 == 0 && ((uintptr_t)dy) % 16 == 0 &&
                       (!relu || ((uintptr_t)y) % 16 == 0);
#define BN_SUMS(T, TY)                                                                                                      \
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, TY>), g1, dim3(256), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, C, ldz, \
                       ldy, rows_per_img, y_img_stride, save_mean, save_invstd, relu, vec_ok, pl.rows_per_split, ws);        \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, st, ws, pl.nsplit, C, M, sums + C, sums, fin)
    if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) { BN_SUMS(bf16_t, bf16_t); }
    else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) { BN_SUMS(bf16_t, float); }
    else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) { BN_SUMS(float, float); }
    else ODTK_REQUIRE(false, "bn_bwd_sums: bad dtype");
#undef BN_SUMS
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_bn_bwd_given(const void* z, const void* y, const void* dy, int M, int C, int ldz, int dtype, int y_dtype, int ldy,
                                 int rows_per_img, long long y_img_stride, const float* gamma, const float* save_mean,
                                 const float* save_invstd, int relu, const float* sums_global, long long count, void* dz, void* workspace,
                                 void* stream) {
    ODTK_REQUIRE(z && dy && dz && gamma && save_mean && save_invstd && sums_global && workspace, "bn_bwd_given: null pointer");
    ODTK_REQUIRE(!relu || y, "bn_bwd_given: relu needs y");
    ODTK_REQUIRE(count >= M, "bn_bwd_given: count=%lld < local rows %d", count, M);
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(ldz % kc == 0 && ldz >= C, "bn_bwd_given: ldz=%d must be a multiple of %d", ldz, kc);
    hipStream_t st = (hipStream_t)stream;
    const RedPlan pl = red_plan(M, C, kc);
    float* ws = (float*)workspace;
    // the apply pass is elementwise: many short workgroups keep more loads in flight than one long one per CU (the statistics
    // pass keeps <= 256 row splits because its partials live in the workspace)
    const int rows_per_block = pl.rows_per_split < bn_rpb(C, ldz, dtype) ? pl.rows_per_split : bn_rpb(C, ldz, dtype);
    float* fin = ws + (size_t)2 * bn_ws_rows(C) * ((C + 63) / 64 * 64);
    dim3 g2(ceil_div(ldz, 8 * kc), ceil_div(M, rows_per_block));
    const size_t ysz = y_dtype == ODTK_BF16 ? 2 : 4;
    const int vec_ok = ((size_t)ldy * ysz) % 16 == 0 && ((size_t)y_img_stride * ysz) % 16 == 0 && ((uintptr_t)dy) % 16 == 0 &&
                       (!relu || ((uintptr_t)y) % 16 == 0);
    hipLaunchKernelGGL(bn_scale_sums_kernel, dim3(ceil_div(2 * C, 256)), dim3(256), 0, st, sums_global, 2 * C, 1.f / (float)count, fin);
#define BN_GIVEN(T, TY)                                                                                                       \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TY>), g2, dim3(256), 0, st, (const T*)z, (const TY*)y, (const TY*)dy, M, C, ldz,   \
                       ldy, rows_per_img, y_img_stride, gamma, save_mean, save_invstd, relu, vec_ok, (T*)dz, fin, rows_per_block)
    if (dtype == ODTK_BF16 && y_dtype == ODTK_BF16) { BN_GIVEN(bf16_t, bf16_t); }
    else if (dtype == ODTK_BF16 && y_dtype == ODTK_F32) { BN_GIVEN(bf16_t, float); }
    else if (dtype == ODTK_F32 && y_dtype == ODTK_F32) { BN_GIVEN(float, float); }
    else ODTK_REQUIRE(false, "bn_bwd_given: bad dtype");
#undef BN_GIVEN
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Residual sum / pitched copy / nearest 2x up-sampling: the glue between the convolutions of the residual and
// pyramid detectors (YOLOv3.py:489-491 `conv = conv + conv2`, :411-412 resize_nearest_neighbor + concat).
// HBM-bound, 16 bytes per lane, rows addressed through their own pitch so that a channel slice of a concat buffer
// is an ordinary operand.
namespace odtk {
namespace {

// y[m, :C] = a[m, :C] (+ b[m, :C])
template <typename T>
__global__ void __launch_bounds__(256) add2d_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, T* __restrict__ y,
                                                    int ldy, long long M, int C) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = C / KC;
    const long long total = M * cpr, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / cpr;
        const int c = (int)(i - m * cpr) * KC;
        float fa[KC];
        Chunk<T>::unpack(ld16(a + m * lda + c), fa);
        if (b) {
            float fb[KC];
            Chunk<T>::unpack(ld16(b + m * ldb + c), fb);
#pragma unroll
            for (int e = 0; e < KC; ++e) fa[e] += fb[e];
        }
        st16(y + m * ldy + c, Chunk<T>::pack(fa));
    }
}

// y[n, 2h + i, 2w + j, :C] = x[n, h, w, :C]
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int H, int W,
                                                             int C) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = C / KC;
    const long long total = (long long)N * 2 * H * 2 * W * cpr, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr) * KC;
        const int wo = (int)(row % (2 * W));
        const long long r2 = row / (2 * W);
        const int ho = (int)(r2 % (2 * H)), n = (int)(r2 / (2 * H));
        st16(y + row * ldy + c, ld16(x + (((long long)n * H + (ho >> 1)) * W + (wo >> 1)) * ldx + c));
    }
}

// dx[n, h, w] (+)= sum of the four dy it was copied to (f32 sum, one rounding)
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int H,
                                                             int W, int C, int accumulate) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = C / KC;
    const long long total = (long long)N * H * W * cpr, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr) * KC;
        const int w = (int)(row % W);
        const long long r2 = row / W;
        const int h = (int)(r2 % H), n = (int)(r2 / H);
        float s[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) s[e] = 0.f;
        if (accumulate) Chunk<T>::unpack(ld16(dx + row * lddx + c), s);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f[KC];
            Chunk<T>::unpack(ld16(dy + (((long long)n * 2 * H + 2 * h + (q >> 1)) * 2 * W + 2 * w + (q & 1)) * lddy + c), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) s[e] += f[e];
        }
        st16(dx + row * lddx + c, Chunk<T>::pack(s));
    }
}

int glue_check(const char* who, int C, int dtype, std::initializer_list<int> pitches, std::initializer_list<const void*> ptrs) {
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(dtype == ODTK_BF16 || dtype == ODTK_F32, "%s: bad dtype %d", who, dtype);
    ODTK_REQUIRE(C > 0 && C % kc == 0, "%s: C=%d must be a multiple of %d", who, C, kc);
    for (int ld : pitches) ODTK_REQUIRE(ld >= C && ld % kc == 0, "%s: pitch %d must be >= C and a multiple of %d", who, ld, kc);
    for (const void* p : ptrs) ODTK_REQUIRE(p && ((uintptr_t)p) % 16 == 0, "%s: operands must be non-null and 16-byte aligned", who);
    return ODTK_OK;
}

}  // namespace
}  // namespace odtk

extern "C" int odtk_add2d(const void* a, int lda, const void* b, int ldb, void* y, int ldy, long long M, int C, int dtype, void* stream) {
    if (int e = glue_check("add2d", C, dtype, {lda, ldy, b ? ldb : C}, {a, y, b ? b : a})) return e;
    ODTK_REQUIRE(M > 0, "add2d: M=%lld", M);
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(add2d_kernel<T>, dim3(grid_for(M * (C / kc), 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)a, lda, (const T*)b, ldb, (T*)y, ldy, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream) {
    if (int e = glue_check("upsample2x_fwd", C, dtype, {ldx, ldy}, {x, y})) return e;
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0, "upsample2x_fwd: bad geometry");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3(grid_for((long long)N * 4 * H * W * (C / kc), 256, 65536)), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, N, H, W, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int dtype, int accumulate,
                                   void* stream) {
    if (int e = glue_check("upsample2x_bwd", C, dtype, {lddy, lddx}, {dy, dx})) return e;
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0, "upsample2x_bwd: bad geometry");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(grid_for((long long)N * H * W * (C / kc), 256, 65536)), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)dy, lddy, (T*)dx, lddx, N, H, W, C, accumulate);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// tf.image.resize_bilinear(x, size) on feature maps, TF-1.x grid (align_corners=False, no half-pixel centres: src = dst * in/out),
// forward and backward: the top-down path of the RetinaNet / FCOS pyramids (RetinaNet.py:309, FCOS.py:373).  NHWC rows with their
// own pitch, 16 bytes of channels per lane.  Backward is a gather over the (few) output pixels that read an input pixel -- no atomics:
// for an up-scaling by s every input pixel is touched by at most (ceil(s) + 1)^2 outputs, found by inverting the index map.
namespace odtk {
namespace {

template <typename T>
__global__ void __launch_bounds__(256) resize_bilinear_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int H,
                                                                  int W, int Ho, int Wo, int C, int accumulate, const float sy, const float sx) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = C / KC;
    const long long total = (long long)N * Ho * Wo * cpr, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr) * KC;
        const int xo = (int)(row % Wo);
        const long long r2 = row / Wo;
        const int yo = (int)(r2 % Ho), n = (int)(r2 / Ho);
        const float fy = (float)yo * sy, fx = (float)xo * sx;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        float tl[KC], tr[KC], bl[KC], br[KC], o[KC];
        const T* base = x + (long long)n * H * W * ldx + c;
        Chunk<T>::unpack(ld16(base + ((long long)y0 * W + x0) * ldx), tl);
        Chunk<T>::unpack(ld16(base + ((long long)y0 * W + x1) * ldx), tr);
        Chunk<T>::unpack(ld16(base + ((long long)y1 * W + x0) * ldx), bl);
        Chunk<T>::unpack(ld16(base + ((long long)y1 * W + x1) * ldx), br);
        if (accumulate) Chunk<T>::unpack(ld16(y + row * ldy + c), o);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float top = tl[e] + (tr[e] - tl[e]) * lx, bot = bl[e] + (br[e] - bl[e]) * lx;
            const float v = top + (bot - top) * ly;
            o[e] = accumulate ? o[e] + v : v;
        }
        st16(y + row * ldy + c, Chunk<T>::pack(o));
    }
}

// weight of input index `in` in output index `out` along one axis (0 when it is not one of the two taps)
__device__ __forceinline__ float bilinear_tap_weight(int out, int in, float scale, int in_size) {
    const float f = (float)out * scale;
    const int i0 = (int)floorf(f), i1 = min(i0 + 1, in_size - 1);
    const float l = f - (float)i0;
    float w = 0.f;
    if (in == i0) w += 1.f - l;
    if (in == i1) w += l;
    return w;
}

template <typename T>
__global__ void __launch_bounds__(256) resize_bilinear_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int H,
                                                                  int W, int Ho, int Wo, int C, int accumulate, const float sy, const float sx,
                                                                  const T* __restrict__ relu_src) {
    constexpr int KC = Chunk<T>::N;
    const int cpr = C / KC;
    const long long total = (long long)N * H * W * cpr, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr) * KC;
        const int xi = (int)(row % W);
        const long long r2 = row / W;
        const int yi = (int)(r2 % H), n = (int)(r2 / H);
        // outputs whose taps can include (yi, xi): floor(out * s) in {yi - 1, yi}  <=>  out in ((yi - 1) / s, (yi + 1) / s)
        // (one more candidate on either side: the weights below re-derive the forward taps, a wider range only costs a test)
        const int ya = max((int)floorf((float)(yi - 1) / sy) - 1, 0), yb = min((int)ceilf((float)(yi + 1) / sy) + 1, Ho - 1);
        const int xa = max((int)floorf((float)(xi - 1) / sx) - 1, 0), xb = min((int)ceilf((float)(xi + 1) / sx) + 1, Wo - 1);
        float s[KC], acc0[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) { s[e] = 0.f; acc0[e] = 0.f; }
        if (accumulate) Chunk<T>::unpack(ld16(dx + row * lddx + c), acc0);
        for (int yo = ya; yo <= yb; ++yo) {
            const float wy = bilinear_tap_weight(yo, yi, sy, H);
            if (wy == 0.f) continue;
            for (int xo = xa; xo <= xb; ++xo) {
                const float w = wy * bilinear_tap_weight(xo, xi, sx, W);
                if (w == 0.f) continue;
                float f[KC];
                Chunk<T>::unpack(ld16(dy + (((long long)n * Ho + yo) * Wo + xo) * lddy + c), f);
#pragma unroll
                for (int e = 0; e < KC; ++e) s[e] += w * f[e];
            }
        }
        if (relu_src) {                                    // the input is a bias + ReLU activation whose gradient buffer holds d(pre-activation)
            float r[KC];
            Chunk<T>::unpack(ld16(relu_src + row * lddx + c), r);
#pragma unroll
            for (int e = 0; e < KC; ++e) if (!(r[e] > 0.f)) s[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < KC; ++e) s[e] += acc0[e];
        st16(dx + row * lddx + c, Chunk<T>::pack(s));
    }
}

// dst[m][dst_off + c] (+)= src[m][src_off + c], c < C: tf.concat / its gradient when the pieces do NOT start on 16-byte channel boundaries
// (PFPNetR's 512 + 85 + 85 + 85 features).  Element granular; `relu_src` (same rows as dst) zeroes the copy where the ReLU output is <= 0.
template <typename T>
__global__ void __launch_bounds__(256) copy_channels_kernel(const T* __restrict__ src, int lds, int src_off, T* __restrict__ dst, int ldd, int dst_off,
                                                            long long M, int C, int accumulate, const T* __restrict__ relu_src) {
    const long long total = M * C, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        float v = elem<T>::load(src[m * lds + src_off + c]);
        T* d = dst + m * ldd + dst_off + c;
        if (relu_src && !(elem<T>::load(relu_src[m * ldd + dst_off + c]) > 0.f)) v = 0.f;
        if (accumulate) v += elem<T>::load(*d);
        *d = elem<T>::store(v);
    }
}

}  // namespace
}  // namespace odtk

static float resize_scale(int in, int out, int align_corners) {          // CalculateResizeScale (tensorflow/core/kernels/image_resizer_state.h)
    return (align_corners && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out;
}

extern "C" int odtk_resize_bilinear2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                                         int align_corners, int accumulate, void* stream) {
    if (int e = glue_check("resize_bilinear_fwd", C, dtype, {ldx, ldy}, {x, y})) return e;
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_bilinear_fwd: bad geometry");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(resize_bilinear_fwd_kernel<T>, dim3(grid_for((long long)N * Ho * Wo * (C / kc), 256, 65536)), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, N, H, W, Ho, Wo, C, accumulate,
                                           resize_scale(H, Ho, align_corners), resize_scale(W, Wo, align_corners));)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_resize_bilinear2_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                                         int align_corners, int accumulate, const void* relu_src, void* stream) {
    if (int e = glue_check("resize_bilinear_bwd", C, dtype, {lddy, lddx}, {dy, dx})) return e;
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_bilinear_bwd: bad geometry");
    ODTK_REQUIRE(!relu_src || ((uintptr_t)relu_src % 16) == 0, "resize_bilinear_bwd: relu_src must be 16-byte aligned");
    const int kc = dtype == ODTK_BF16 ? 8 : 4;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(resize_bilinear_bwd_kernel<T>, dim3(grid_for((long long)N * H * W * (C / kc), 256, 65536)), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)dy, lddy, (T*)dx, lddx, N, H, W, Ho, Wo, C, accumulate,
                                           resize_scale(H, Ho, align_corners), resize_scale(W, Wo, align_corners), (const T*)relu_src);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_resize_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                                        int accumulate, void* stream) {
    return odtk_resize_bilinear2_fwd(x, ldx, y, ldy, N, H, W, Ho, Wo, C, dtype, 0, accumulate, stream);
}

extern "C" int odtk_resize_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                                        int accumulate, void* stream) {
    return odtk_resize_bilinear2_bwd(dy, lddy, dx, lddx, N, H, W, Ho, Wo, C, dtype, 0, accumulate, nullptr, stream);
}

extern "C" int odtk_copy_channels(const void* src, int lds, int src_off, void* dst, int ldd, int dst_off, long long M, int C, int dtype,
                                  int accumulate, const void* relu_src, void* stream) {
    ODTK_REQUIRE(src && dst && M > 0 && C > 0 && src_off >= 0 && dst_off >= 0 && src_off + C <= lds && dst_off + C <= ldd,
                 "copy_channels: bad argument (C=%d, offsets %d / %d, pitches %d / %d)", C, src_off, dst_off, lds, ldd);
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(grid_for(M * C, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)src, lds, src_off, (T*)dst, ldd, dst_off, M, C, accumulate, (const T*)relu_src);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// conv rows <-> the f32 prediction tensors of the box-side kernels: a subnet's last convolution leaves [N*H*W][ld] rows in the
// compute dtype, the loss / decode kernels read pconf [N][A][classes] / pbbox [N][A][4] with all levels of one image back to back
// (RetinaNet.py:184-186, :321-326).  Row m of image n = m / rows_per_img lands at y + n * y_img_stride + (m % rows_per_img) * ldy.
namespace odtk {
namespace {

template <typename T>
__global__ void __launch_bounds__(256) rows_to_f32_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows_per_img,
                                                          long long y_img_stride, long long M, int C) {
    const long long total = M * C, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        const long long n = m / rows_per_img;
        y[n * y_img_stride + (m - n * rows_per_img) * ldy + c] = elem<T>::load(x[m * ldx + c]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) rows_from_f32_kernel(const float* __restrict__ y, int ldy, int rows_per_img, long long y_img_stride,
                                                            T* __restrict__ x, int ldx, long long M, int C) {
    const long long total = M * ldx, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / ldx;
        const int c = (int)(i - m * ldx);
        const long long n = m / rows_per_img;
        x[i] = elem<T>::store(c < C ? y[n * y_img_stride + (m - n * rows_per_img) * ldy + c] : 0.f);      // pad columns zeroed
    }
}

}  // namespace
}  // namespace odtk

extern "C" int odtk_rows_to_f32(const void* x, int ldx, int dtype, float* y, int ldy, int rows_per_img, long long y_img_stride, long long M,
                                int C, void* stream) {
    ODTK_REQUIRE(x && y && M > 0 && C > 0 && ldx >= C && ldy >= C && rows_per_img > 0, "rows_to_f32: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(rows_to_f32_kernel<T>, dim3(grid_for(M * C, 256, 65536)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                           ldx, y, ldy, rows_per_img, y_img_stride, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_rows_from_f32(const float* y, int ldy, int rows_per_img, long long y_img_stride, void* x, int ldx, int dtype, long long M,
                                  int C, void* stream) {
    ODTK_REQUIRE(x && y && M > 0 && C > 0 && ldx >= C && ldy >= C && rows_per_img > 0, "rows_from_f32: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(rows_from_f32_kernel<T>, dim3(grid_for(M * ldx, 256, 65536)), dim3(256), 0, (hipStream_t)stream, y, ldy,
                                           rows_per_img, y_img_stride, (T*)x, ldx, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// tf.contrib.layers.group_norm(groups, epsilon 1e-6) (+ ReLU) on NHWC rows, forward and backward: the normalisation of the reference's
// FCOS (FCOS.py:438-446: EVERY norm of that model).  Statistics are per SAMPLE and group over H x W x (C / groups) elements, so one
// workgroup owns one (sample, group): it reduces, then applies -- no partial sums, no second launch; the rows it re-reads come from L2.
// Backward: the same ownership for dx; dgamma / dbeta need a sum over the samples, taken in a fixed order from per-sample partials.
namespace odtk {
namespace {

constexpr int GN_THREADS = 256;

__device__ __forceinline__ float gn_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GN_THREADS / 64; ++w) t += red[w];
    return t;
}

// grid (groups, N).  x, y [N*HW][ld]; save [N][groups][2] = mean, rstd
template <typename T>
__global__ void __launch_bounds__(GN_THREADS) gn_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int HW, int C, int groups,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                            float* __restrict__ save) {
    __shared__ float red[GN_THREADS / 64];
    const int g = blockIdx.x, n = blockIdx.y, cg = C / groups, c0 = g * cg;
    const long long row0 = (long long)n * HW;
    const int total = HW * cg;
    const float shift = elem<T>::load(x[row0 * ldx + c0]);               // sums about the first element: no cancellation for large means
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < total; i += GN_THREADS) {
        const int r = i / cg, c = i - r * cg;
        const float d = elem<T>::load(x[(row0 + r) * ldx + c0 + c]) - shift;
        s1 += d; s2 += d * d;
    }
    s1 = gn_block_sum(s1, red);
    s2 = gn_block_sum(s2, red);
    const float dm = s1 / (float)total;
    const float mean = shift + dm, var = fmaxf(s2 / (float)total - dm * dm, 0.f);
    const float rstd = rsqrtf(var + 1e-6f);
    if (threadIdx.x == 0 && save) { save[((size_t)n * groups + g) * 2] = mean; save[((size_t)n * groups + g) * 2 + 1] = rstd; }
    for (int i = threadIdx.x; i < total; i += GN_THREADS) {
        const int r = i / cg, c = i - r * cg;
        float v = (elem<T>::load(x[(row0 + r) * ldx + c0 + c]) - mean) * rstd * gamma[c0 + c] + beta[c0 + c];
        if (relu) v = fmaxf(v, 0.f);
        y[(row0 + r) * ldy + c0 + c] = elem<T>::store(v);
    }
}

// grid (groups, N).  dx [N*HW][lddx] (accumulate: dx += ...); part [N][2][C] = this sample's dgamma, dbeta contributions
template <typename T>
__global__ void __launch_bounds__(GN_THREADS) gn_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ y, const T* __restrict__ dy, int ldy,
                                                            T* __restrict__ dx, int lddx, int HW, int C, int groups,
                                                            const float* __restrict__ gamma, const float* __restrict__ save, int relu,
                                                            int accumulate, float* __restrict__ part) {
    __shared__ float red[GN_THREADS / 64];
    const int g = blockIdx.x, n = blockIdx.y, cg = C / groups, c0 = g * cg;
    const long long row0 = (long long)n * HW;
    const int total = HW * cg;
    const float mean = save[((size_t)n * groups + g) * 2], rstd = save[((size_t)n * groups + g) * 2 + 1];
    // per-channel sums (dgamma, dbeta of this sample): thread t owns channel t % cg of rows t / cg, t / cg + rows_per_pass, ...
    float a1 = 0.f, a2 = 0.f;                                            // group sums of dy' * gamma and dy' * gamma * xhat
    float sg = 0.f, sb = 0.f;                                            // this thread's share of dgamma / dbeta (fast path below)
    const bool fixed_channel = GN_THREADS % cg == 0;                     // then thread t only ever sees channel t % cg
    for (int i = threadIdx.x; i < total; i += GN_THREADS) {
        const int r = i / cg, c = i - r * cg;
        float d = elem<T>::load(dy[(row0 + r) * ldy + c0 + c]);
        if (relu && !(elem<T>::load(y[(row0 + r) * ldy + c0 + c]) > 0.f)) d = 0.f;
        const float xh = (elem<T>::load(x[(row0 + r) * ldx + c0 + c]) - mean) * rstd;
        const float dg = d * gamma[c0 + c];
        a1 += dg; a2 += dg * xh;
        sg += d * xh; sb += d;
    }
    a1 = gn_block_sum(a1, red);
    a2 = gn_block_sum(a2, red);
    const float m1 = a1 / (float)total, m2 = a2 / (float)total;
    for (int i = threadIdx.x; i < total; i += GN_THREADS) {
        const int r = i / cg, c = i - r * cg;
        float d = elem<T>::load(dy[(row0 + r) * ldy + c0 + c]);
        if (relu && !(elem<T>::load(y[(row0 + r) * ldy + c0 + c]) > 0.f)) d = 0.f;
        const float xh = (elem<T>::load(x[(row0 + r) * ldx + c0 + c]) - mean) * rstd;
        float v = rstd * (d * gamma[c0 + c] - m1 - xh * m2);
        T* o = dx + (row0 + r) * lddx + c0 + c;
        if (accumulate) v += elem<T>::load(*o);
        *o = elem<T>::store(v);
    }
    // dgamma / dbeta partials of this sample
    if (fixed_channel) {                                                 // fold the threads that share a channel, in thread order
        __shared__ float s_g[GN_THREADS], s_b[GN_THREADS];
        s_g[threadIdx.x] = sg; s_b[threadIdx.x] = sb;
        __syncthreads();
        if ((int)threadIdx.x < cg) {
            float tg = 0.f, tb = 0.f;
            for (int j = threadIdx.x; j < GN_THREADS; j += cg) { tg += s_g[j]; tb += s_b[j]; }
            part[((size_t)n * 2 + 0) * C + c0 + threadIdx.x] = tg;
            part[((size_t)n * 2 + 1) * C + c0 + threadIdx.x] = tb;
        }
        return;
    }
    for (int c = threadIdx.x; c < cg; c += GN_THREADS) {                 // general case: one thread per channel walks the rows
        float sg = 0.f, sb = 0.f;
        for (int r = 0; r < HW; ++r) {
            float d = elem<T>::load(dy[(row0 + r) * ldy + c0 + c]);
            if (relu && !(elem<T>::load(y[(row0 + r) * ldy + c0 + c]) > 0.f)) d = 0.f;
            sg += d * (elem<T>::load(x[(row0 + r) * ldx + c0 + c]) - mean) * rstd;
            sb += d;
        }
        part[((size_t)n * 2 + 0) * C + c0 + c] = sg;
        part[((size_t)n * 2 + 1) * C + c0 + c] = sb;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Group norm on 16-byte channel chunks (round 2).  The kernels above give one workgroup a (sample, group) pair and walk its HW x (C / groups) elements
// one 2-byte load at a time -- with 2..8 channels per group that is 4..16 useful bytes per pixel row and an integer division per element: the FCOS
// step spent 79 % of its time there (gn_bwd 38 ms, gn_fwd 11 ms of 61.5 ms at 512 x 512, batch 16).  Here rows are read as whole chunks (8 bf16 / 4 f32
// channels per lane, 8 chunk lanes x 32 row lanes per workgroup, the batch-norm kernels' shape), per-CHANNEL sums are reduced over row splits, a small
// finalize kernel folds the channels of a group (in double: the channels carry different shifts), and the apply pass is elementwise.
// ws: [N][splits][2][Cp] channel sums; st: [N][groups][2] mean / rstd (forward) or m1 / m2 (backward).
template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_stats_kernel(const T* __restrict__ x, int ldx, int HW, int C, int Cp, int rows_per_split,
                                                             float* __restrict__ ws) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    const int split = blockIdx.y, nsplit = gridDim.y, n = blockIdx.z;
    const T* xs = x + (size_t)n * HW * ldx;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        float sh[KC];
        Chunk<T>::unpack(ld16(xs + c0), sh);                            // the sample's first row: the same shift in every split
        const int m0 = split * rows_per_split;
        int m1 = m0 + rows_per_split; if (m1 > HW) m1 = HW;
#pragma unroll 4
        for (int m = m0 + rl; m < m1; m += RED_ROWS) {
            float f[KC];
            Chunk<T>::unpack(ld16(xs + (size_t)m * ldx + c0), f);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                const float d = f[e] - sh[e];
                acc[e] += d;
                acc[KC + e] += d * d;
            }
        }
    }
    block_rowlane_reduce<2 * KC>(acc, sm, rl, cl);
    if (rl == 0 && c0 < C) {
        float* w = ws + ((size_t)(n * nsplit + split) * 2) * Cp + c0;
#pragma unroll
        for (int e = 0; e < KC; ++e) { w[e] = acc[e]; w[Cp + e] = acc[KC + e]; }
    }
}

// one workgroup per sample, one thread per group: mean / rstd from the channel sums
template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_finalize_kernel(const T* __restrict__ x, int ldx, int HW, int C, int Cp, int groups, int nsplit,
                                                                const float* __restrict__ ws, float* __restrict__ st) {
    const int n = blockIdx.x, cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double sx = 0.0, sxx = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            double s1 = 0.0, s2 = 0.0;
            for (int s = 0; s < nsplit; ++s) {
                const float* w = ws + ((size_t)(n * nsplit + s) * 2) * Cp + c;
                s1 += (double)w[0]; s2 += (double)w[Cp];
            }
            const double sh = (double)elem<T>::load(x[(size_t)n * HW * ldx + c]);
            sx += s1 + (double)HW * sh;
            sxx += s2 + 2.0 * sh * s1 + (double)HW * sh * sh;
        }
        const double cnt = (double)HW * cg;
        const double mean = sx / cnt;
        double var = sxx / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        st[((size_t)n * groups + g) * 2] = (float)mean;
        st[((size_t)n * groups + g) * 2 + 1] = rsqrtf((float)var + 1e-6f);
    }
}

// The same two finalize steps with 64 channels x 4 split lanes per workgroup, grid (C / 64, N), for group sizes that divide 64 (every FCOS layer: 2 .. 64
// channels per group): the one-thread-per-group loops above walked up to 64 channels x 128 splits serially and took 3.7 + 3.6 ms of a 24.8 ms FCOS step.
template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_finalize64_kernel(const T* __restrict__ x, int ldx, int HW, int C, int Cp, int groups, int nsplit,
                                                                  const float* __restrict__ ws, float* __restrict__ st) {
    __shared__ float s_a[4][64], s_b[4][64];
    __shared__ double s_x[64], s_xx[64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, sl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, cg = C / groups;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int s = sl; s < nsplit; s += 4) {
            const float* w = ws + ((size_t)(n * nsplit + s) * 2) * Cp + c;
            a += w[0]; b += w[Cp];
        }
    s_a[sl][cl] = a; s_b[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
        const double s1 = ((double)s_a[0][cl] + (double)s_a[1][cl]) + ((double)s_a[2][cl] + (double)s_a[3][cl]);
        const double s2 = ((double)s_b[0][cl] + (double)s_b[1][cl]) + ((double)s_b[2][cl] + (double)s_b[3][cl]);
        const double sh = (double)elem<T>::load(x[(size_t)n * HW * ldx + c]);
        s_x[cl] = s1 + (double)HW * sh;
        s_xx[cl] = s2 + 2.0 * sh * s1 + (double)HW * sh * sh;
    }
    __syncthreads();
    const int gl = threadIdx.x;                             // group inside this block
    if (gl < 64 / cg && blockIdx.x * 64 + gl * cg < C) {
        double sx = 0.0, sxx = 0.0;
        for (int j = 0; j < cg; ++j) { sx += s_x[gl * cg + j]; sxx += s_xx[gl * cg + j]; }
        const double cnt = (double)HW * cg;
        const double mean = sx / cnt;
        double var = sxx / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const int g = (blockIdx.x * 64) / cg + gl;
        st[((size_t)n * groups + g) * 2] = (float)mean;
        st[((size_t)n * groups + g) * 2 + 1] = rsqrtf((float)var + 1e-6f);
    }
}

__global__ void __launch_bounds__(256) gn_chunk_bwd_finalize64_kernel(int HW, int C, int Cp, int groups, int nsplit, const float* __restrict__ ws,
                                                                      const float* __restrict__ gamma, float* __restrict__ part, float* __restrict__ st) {
    __shared__ float s_a[4][64], s_b[4][64], s_gb[64], s_gg[64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, sl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, cg = C / groups;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int s = sl; s < nsplit; s += 4) {
            const float* w = ws + ((size_t)(n * nsplit + s) * 2) * Cp + c;
            a += w[0]; b += w[Cp];
        }
    s_a[sl][cl] = a; s_b[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
        const float sb = (s_a[0][cl] + s_a[1][cl]) + (s_a[2][cl] + s_a[3][cl]);
        const float sg = (s_b[0][cl] + s_b[1][cl]) + (s_b[2][cl] + s_b[3][cl]);
        part[((size_t)n * 2 + 0) * C + c] = sg;
        part[((size_t)n * 2 + 1) * C + c] = sb;
        s_gb[cl] = gamma[c] * sb;
        s_gg[cl] = gamma[c] * sg;
    }
    __syncthreads();
    const int gl = threadIdx.x;
    if (gl < 64 / cg && blockIdx.x * 64 + gl * cg < C) {
        float a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < cg; ++j) { a1 += s_gb[gl * cg + j]; a2 += s_gg[gl * cg + j]; }
        const float cnt = (float)HW * (float)cg;
        const int g = (blockIdx.x * 64) / cg + gl;
        st[((size_t)n * groups + g) * 2] = a1 / cnt;
        st[((size_t)n * groups + g) * 2 + 1] = a2 / cnt;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_apply_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int HW, int C, int groups,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                             const float* __restrict__ st, int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    if (c0 >= C) return;
    const int n = blockIdx.z, cg = C / groups;
    float sc[KC], of[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int g = (c0 + e) / cg;
        const float mean = st[((size_t)n * groups + g) * 2], rstd = st[((size_t)n * groups + g) * 2 + 1];
        sc[e] = rstd * gamma[c0 + e];
        of[e] = beta[c0 + e] - mean * sc[e];
    }
    const size_t base = (size_t)n * HW;
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block; if (m1 > HW) m1 = HW;
    for (int m = m0 + rl; m < m1; m += RED_ROWS) {
        float f[KC];
        Chunk<T>::unpack(ld16(x + (base + m) * ldx + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            f[e] = f[e] * sc[e] + of[e];
            if (relu) f[e] = fmaxf(f[e], 0.f);
        }
        st16(y + (base + m) * ldy + c0, Chunk<T>::pack(f));
    }
}

// backward: per-channel sums of dy' (ReLU-masked) and dy' * xhat over a row split
template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_bwd_stats_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ y, const T* __restrict__ dy, int ldy,
                                                                 int HW, int C, int Cp, int groups, const float* __restrict__ save, int relu,
                                                                 int rows_per_split, float* __restrict__ ws) {
    constexpr int KC = Chunk<T>::N;
    __shared__ float sm[RED_ROWS * 8 * 2 * KC];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    const int split = blockIdx.y, nsplit = gridDim.y, n = blockIdx.z, cg = C / groups;
    float acc[2 * KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        float mu[KC], rs[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const int g = (c0 + e) / cg;
            mu[e] = save[((size_t)n * groups + g) * 2]; rs[e] = save[((size_t)n * groups + g) * 2 + 1];
        }
        const size_t base = (size_t)n * HW;
        const int m0 = split * rows_per_split;
        int m1 = m0 + rows_per_split; if (m1 > HW) m1 = HW;
#pragma unroll 2
        for (int m = m0 + rl; m < m1; m += RED_ROWS) {
            float f[KC], d[KC], yy[KC];
            Chunk<T>::unpack(ld16(x + (base + m) * ldx + c0), f);
            Chunk<T>::unpack(ld16(dy + (base + m) * ldy + c0), d);
            if (relu) Chunk<T>::unpack(ld16(y + (base + m) * ldy + c0), yy);
#pragma unroll
            for (int e = 0; e < KC; ++e) {
                const float dd = (relu && !(yy[e] > 0.f)) ? 0.f : d[e];
                acc[e] += dd;
                acc[KC + e] += dd * ((f[e] - mu[e]) * rs[e]);
            }
        }
    }
    block_rowlane_reduce<2 * KC>(acc, sm, rl, cl);
    if (rl == 0 && c0 < C) {
        float* w = ws + ((size_t)(n * nsplit + split) * 2) * Cp + c0;
#pragma unroll
        for (int e = 0; e < KC; ++e) { w[e] = acc[e]; w[Cp + e] = acc[KC + e]; }
    }
}

// one workgroup per sample: channel totals -> part [N][2][C] (dgamma, dbeta shares of this sample); group means m1 = mean(dy' gamma), m2 = mean(dy' gamma xhat)
__global__ void __launch_bounds__(256) gn_chunk_bwd_finalize_kernel(int HW, int C, int Cp, int groups, int nsplit, const float* __restrict__ ws,
                                                                    const float* __restrict__ gamma, float* __restrict__ part, float* __restrict__ st) {
    const int n = blockIdx.x, cg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sb = 0.f, sg = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float* w = ws + ((size_t)(n * nsplit + s) * 2) * Cp + c;
            sb += w[0]; sg += w[Cp];
        }
        part[((size_t)n * 2 + 0) * C + c] = sg;
        part[((size_t)n * 2 + 1) * C + c] = sb;
    }
    for (int g = threadIdx.x; g < groups; g += 256) {      // (re-summed from ws: no read-back of this block's own global stores)
        float a1 = 0.f, a2 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            float sb = 0.f, sg = 0.f;
            for (int s = 0; s < nsplit; ++s) {
                const float* w = ws + ((size_t)(n * nsplit + s) * 2) * Cp + c;
                sb += w[0]; sg += w[Cp];
            }
            a1 += gamma[c] * sb;
            a2 += gamma[c] * sg;
        }
        const float cnt = (float)HW * (float)cg;
        st[((size_t)n * groups + g) * 2] = a1 / cnt;
        st[((size_t)n * groups + g) * 2 + 1] = a2 / cnt;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_chunk_bwd_apply_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ y, const T* __restrict__ dy, int ldy,
                                                                 T* __restrict__ dx, int lddx, int HW, int C, int groups, const float* __restrict__ gamma,
                                                                 const float* __restrict__ save, const float* __restrict__ st, int relu, int accumulate,
                                                                 int rows_per_block) {
    constexpr int KC = Chunk<T>::N;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * KC;
    if (c0 >= C) return;
    const int n = blockIdx.z, cg = C / groups;
    float mu[KC], rs[KC], gm[KC], m1[KC], m2[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int g = (c0 + e) / cg;
        mu[e] = save[((size_t)n * groups + g) * 2]; rs[e] = save[((size_t)n * groups + g) * 2 + 1];
        m1[e] = st[((size_t)n * groups + g) * 2]; m2[e] = st[((size_t)n * groups + g) * 2 + 1];
        gm[e] = gamma[c0 + e];
    }
    const size_t base = (size_t)n * HW;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block; if (r1 > HW) r1 = HW;
    for (int m = r0 + rl; m < r1; m += RED_ROWS) {
        float f[KC], d[KC], yy[KC], o[KC];
        Chunk<T>::unpack(ld16(x + (base + m) * ldx + c0), f);
        Chunk<T>::unpack(ld16(dy + (base + m) * ldy + c0), d);
        if (relu) Chunk<T>::unpack(ld16(y + (base + m) * ldy + c0), yy);
        if (accumulate) Chunk<T>::unpack(ld16(dx + (base + m) * lddx + c0), o);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float dd = (relu && !(yy[e] > 0.f)) ? 0.f : d[e];
            const float xh = (f[e] - mu[e]) * rs[e];
            const float v = rs[e] * (dd * gm[e] - m1[e] - xh * m2[e]);
            o[e] = accumulate ? o[e] + v : v;
        }
        st16(dx + (base + m) * lddx + c0, Chunk<T>::pack(o));
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Group norm of a SMALL map in one launch (round 3).  A group's statistics are local to one sample, so -- unlike batch norm -- no grid-wide reduction is
// needed: a 512-thread workgroup owns ONE group (CL = cg / KC chunk lanes x 512 / CL row lanes; cg < KC: the 8 / cg groups of one chunk) of one sample,
// walks its HW rows twice (statistics, then apply: the second pass hits L2) and folds the channels of the group in LDS.  Grid (C / (KC CL), N): 512
// workgroups for FCOS's 256-channel heads at 16 images.  The split-row path above is three launches forward and three backward of 5-10 us each whatever the
// map size; FCOS at 512 x 512 has ~130 group norms per step (~900 launches, 6.2 of 17.9 ms).  (A first version with 64-channel blocks and 32 row lanes --
// 64 workgroups -- was 9 % SLOWER than the split path at <= 1 024 pixels and 2.7x slower at <= 16 384: few long workgroups lose to many short launches.)
template <typename T, int CL>
__global__ void __launch_bounds__(512) gn_small_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int HW, int C, int groups,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                           float* __restrict__ st) {
    constexpr int KC = Chunk<T>::N, CB = KC * CL, RL = 512 / CL;
    __shared__ float sm[512 * 2 * KC];
    __shared__ double s_x[CB], s_xx[CB];
    __shared__ float s_mean[KC], s_rstd[KC];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c0 = (blockIdx.x * CL + cl) * KC;               // (C is a multiple of CB: always < C)
    const int n = blockIdx.y, cg = C / groups;
    const T* xs = x + (size_t)n * HW * ldx;
    float acc[2 * KC], sh[KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
    Chunk<T>::unpack(ld16(xs + c0), sh);
#pragma unroll 4
    for (int m = rl; m < HW; m += RL) {
        float f[KC];
        Chunk<T>::unpack(ld16(xs + (size_t)m * ldx + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float d = f[e] - sh[e];
            acc[e] += d;
            acc[KC + e] += d * d;
        }
    }
    block_rowlane_reduce_small<2 * KC, CL>(acc, sm, rl, cl);
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const double s1 = (double)acc[e], s2 = (double)acc[KC + e], shd = (double)sh[e];
            s_x[cl * KC + e] = s1 + (double)HW * shd;
            s_xx[cl * KC + e] = s2 + 2.0 * shd * s1 + (double)HW * shd * shd;
        }
    }
    __syncthreads();
    const int ngl = CB / cg;                                  // groups of this workgroup: 1, or KC / cg when a group is smaller than a chunk
    if ((int)threadIdx.x < ngl) {
        const int gl = threadIdx.x;
        double sx = 0.0, sxx = 0.0;
        for (int j = 0; j < cg; ++j) { sx += s_x[gl * cg + j]; sxx += s_xx[gl * cg + j]; }
        const double cnt = (double)HW * cg;
        const double mean = sx / cnt;
        double var = sxx / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = rsqrtf((float)var + 1e-6f);
        s_mean[gl] = (float)mean; s_rstd[gl] = rstd;
        if (st) {
            const int g = (blockIdx.x * CB) / cg + gl;
            st[((size_t)n * groups + g) * 2] = (float)mean;
            st[((size_t)n * groups + g) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    float sc[KC], of[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int gl2 = (cl * KC + e) / cg;
        sc[e] = s_rstd[gl2] * gamma[c0 + e];
        of[e] = beta[c0 + e] - s_mean[gl2] * sc[e];
    }
    T* ys = y + (size_t)n * HW * ldy;
#pragma unroll 4
    for (int m = rl; m < HW; m += RL) {
        float f[KC];
        Chunk<T>::unpack(ld16(xs + (size_t)m * ldx + c0), f);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            f[e] = f[e] * sc[e] + of[e];
            if (relu) f[e] = fmaxf(f[e], 0.f);
        }
        st16(ys + (size_t)m * ldy + c0, Chunk<T>::pack(f));
    }
}

// backward of the same: channel sums of dy' and dy' xhat (-> this sample's dgamma / dbeta shares in `part`), the group means m1 / m2 folded in LDS, then dx
template <typename T, int CL>
__global__ void __launch_bounds__(512) gn_small_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ y, const T* __restrict__ dy, int ldy,
                                                           T* __restrict__ dx, int lddx, int HW, int C, int groups, const float* __restrict__ gamma,
                                                           const float* __restrict__ save, int relu, int accumulate, float* __restrict__ part) {
    constexpr int KC = Chunk<T>::N, CB = KC * CL, RL = 512 / CL;
    __shared__ float sm[512 * 2 * KC];
    __shared__ float s_gb[CB], s_gg[CB], s_m1[KC], s_m2[KC];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c0 = (blockIdx.x * CL + cl) * KC;
    const int n = blockIdx.y, cg = C / groups;
    const size_t base = (size_t)n * HW;
    float acc[2 * KC], mu[KC], rs[KC], gm[KC];
#pragma unroll
    for (int e = 0; e < 2 * KC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int g = (c0 + e) / cg;
        mu[e] = save[((size_t)n * groups + g) * 2]; rs[e] = save[((size_t)n * groups + g) * 2 + 1];
        gm[e] = gamma[c0 + e];
    }
#pragma unroll 2
    for (int m = rl; m < HW; m += RL) {
        float f[KC], d[KC], yy[KC];
        Chunk<T>::unpack(ld16(x + (base + m) * ldx + c0), f);
        Chunk<T>::unpack(ld16(dy + (base + m) * ldy + c0), d);
        if (relu) Chunk<T>::unpack(ld16(y + (base + m) * ldy + c0), yy);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float dd = (relu && !(yy[e] > 0.f)) ? 0.f : d[e];
            acc[e] += dd;
            acc[KC + e] += dd * ((f[e] - mu[e]) * rs[e]);
        }
    }
    block_rowlane_reduce_small<2 * KC, CL>(acc, sm, rl, cl);
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float sb = acc[e], sg = acc[KC + e];
            part[((size_t)n * 2 + 0) * C + c0 + e] = sg;
            part[((size_t)n * 2 + 1) * C + c0 + e] = sb;
            s_gb[cl * KC + e] = gm[e] * sb;
            s_gg[cl * KC + e] = gm[e] * sg;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < CB / cg) {
        const int gl = threadIdx.x;
        float a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < cg; ++j) { a1 += s_gb[gl * cg + j]; a2 += s_gg[gl * cg + j]; }
        const float cnt = (float)HW * (float)cg;
        s_m1[gl] = a1 / cnt; s_m2[gl] = a2 / cnt;
    }
    __syncthreads();
    float m1[KC], m2[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) { const int gl2 = (cl * KC + e) / cg; m1[e] = s_m1[gl2]; m2[e] = s_m2[gl2]; }
#pragma unroll 2
    for (int m = rl; m < HW; m += RL) {
        float f[KC], d[KC], yy[KC], o[KC];
        Chunk<T>::unpack(ld16(x + (base + m) * ldx + c0), f);
        Chunk<T>::unpack(ld16(dy + (base + m) * ldy + c0), d);
        if (relu) Chunk<T>::unpack(ld16(y + (base + m) * ldy + c0), yy);
        if (accumulate) Chunk<T>::unpack(ld16(dx + (base + m) * lddx + c0), o);
#pragma unroll
        for (int e = 0; e < KC; ++e) {
            const float dd = (relu && !(yy[e] > 0.f)) ? 0.f : d[e];
            const float xh = (f[e] - mu[e]) * rs[e];
            const float v = rs[e] * (dd * gm[e] - m1[e] - xh * m2[e]);
            o[e] = accumulate ? o[e] + v : v;
        }
        st16(dx + (base + m) * lddx + c0, Chunk<T>::pack(o));
    }
}

__global__ void gn_param_grad_kernel(const float* __restrict__ part, int N, int C, float* __restrict__ dgamma, float* __restrict__ dbeta, int acc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sg = 0.f, sb = 0.f;
    for (int n = 0; n < N; ++n) { sg += part[((size_t)n * 2 + 0) * C + c]; sb += part[((size_t)n * 2 + 1) * C + c]; }
    dgamma[c] = acc ? dgamma[c] + sg : sg;          // acc: a norm whose gamma / beta serve several activations (FCOS heads shared over the levels)
    dbeta[c] = acc ? dbeta[c] + sb : sb;
}

}  // namespace
}  // namespace odtk

extern "C" long long odtk_gn_workspace_bytes(int N, int C) { return (long long)N * 2 * C * sizeof(float); }

// per-device scratch of the chunked group-norm path: channel sums per row split + the per-(sample, group) statistics.  Grown on demand, never freed or
// moved once handed out (captured graphs keep pointing into it); launches are stream-ordered by the caller.
static float* g_gn_scratch[16];
static size_t g_gn_scratch_bytes[16];
static int gn_scratch(size_t bytes, float** out) {
    int dev = 0;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    ODTK_REQUIRE(dev >= 0 && dev < 16, "gn: device index %d unsupported", dev);
    if (g_gn_scratch_bytes[dev] < bytes) {
        size_t want = g_gn_scratch_bytes[dev] ? 2 * g_gn_scratch_bytes[dev] : ((size_t)8 << 20);
        if (want < bytes) want = bytes;
        void* p = nullptr;
        ODTK_CHECK_HIP(hipMalloc(&p, want));              // fails inside a stream capture: run one eager step first
        g_gn_scratch[dev] = (float*)p; g_gn_scratch_bytes[dev] = want;
    }
    *out = g_gn_scratch[dev];
    return ODTK_OK;
}

static int g_gn_small_rows = 1024;      // odtk_debug_set key 7: group norms of maps with at most this many pixels per sample run as one launch (0 = never)
namespace odtk {
void set_gn_small_rows(int rows) { g_gn_small_rows = rows < 0 ? 0 : rows; }
}  // namespace odtk
struct GnPlan { int kc, colgroups, Cp, nsplit, rows_per_split, rows_per_block; bool chunked; };
static GnPlan gn_plan(int N, int HW, int C, int dtype, std::initializer_list<int> pitches, std::initializer_list<const void*> ptrs) {
    GnPlan p;
    p.kc = dtype == ODTK_BF16 ? 8 : 4;
    p.chunked = C % p.kc == 0;
    for (int ld : pitches) p.chunked = p.chunked && ld % p.kc == 0;
    for (const void* q : ptrs) p.chunked = p.chunked && ((uintptr_t)q % 16) == 0;
    p.colgroups = ceil_div(C, 8 * p.kc);
    p.Cp = p.colgroups * 8 * p.kc;
    int want = ceil_div(1024, N * p.colgroups);          // enough workgroups to fill the chip
    const int maxs = ceil_div(HW, 4 * RED_ROWS);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 128) want = 128;
    p.rows_per_split = ceil_div(HW, want);
    p.nsplit = ceil_div(HW, p.rows_per_split);
    p.rows_per_block = p.rows_per_split < 256 ? p.rows_per_split : 256;
    return p;
}

// chunk lanes of the one-launch kernels for this shape (0 = not eligible): one group per workgroup (cg = kc * CL) or the kc / cg groups of one chunk
static int gn_small_cl(const GnPlan& pl, int HW, int C, int groups) {
    if (!pl.chunked || HW > g_gn_small_rows) return 0;
    const int cg = C / groups;
    if (cg <= pl.kc) return pl.kc % cg == 0 ? 1 : 0;
    if (cg % pl.kc) return 0;
    const int cl = cg / pl.kc;
    return (cl == 2 || cl == 4 || cl == 8) ? cl : 0;
}

extern "C" int odtk_gn_fwd(const void* x, int ldx, void* y, int ldy, int N, int HW, int C, int groups, int dtype, const float* gamma,
                           const float* beta, int relu, float* save_mean_rstd, void* stream) {
    ODTK_REQUIRE(x && y && gamma && beta, "gn_fwd: null pointer");
    ODTK_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && ldx >= C && ldy >= C, "gn_fwd: N=%d HW=%d C=%d groups=%d", N, HW, C, groups);
    hipStream_t st = (hipStream_t)stream;
    const GnPlan pl = gn_plan(N, HW, C, dtype, {ldx, ldy}, {x, y});
    if (const int cl = gn_small_cl(pl, HW, C, groups)) {       // small map: statistics + apply in ONE launch
#define ODTK_GN_SMALL_FWD(CLV)                                                                                                                      \
    DT_SWITCH(dtype, T, hipLaunchKernelGGL((gn_small_fwd_kernel<T, CLV>), dim3(C / (pl.kc * CLV), N), dim3(512), 0, st, (const T*)x, ldx, (T*)y, ldy, HW, C, \
                                           groups, gamma, beta, relu, save_mean_rstd);)
        if (cl == 1) { ODTK_GN_SMALL_FWD(1) } else if (cl == 2) { ODTK_GN_SMALL_FWD(2) } else if (cl == 4) { ODTK_GN_SMALL_FWD(4) } else { ODTK_GN_SMALL_FWD(8) }
#undef ODTK_GN_SMALL_FWD
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (pl.chunked) {
        float* scr = nullptr;
        const size_t wsn = (size_t)N * pl.nsplit * 2 * pl.Cp, stn = (size_t)N * groups * 2;
        if (int e = gn_scratch((wsn + stn) * sizeof(float), &scr)) return e;
        float* stat = save_mean_rstd ? save_mean_rstd : scr + wsn;
        const bool fin64 = 64 % (C / groups) == 0;             // whole groups inside a 64-channel block
        DT_SWITCH(dtype, T,
                  hipLaunchKernelGGL(gn_chunk_stats_kernel<T>, dim3(pl.colgroups, pl.nsplit, N), dim3(256), 0, st, (const T*)x, ldx, HW, C, pl.Cp,
                                     pl.rows_per_split, scr);
                  if (fin64) hipLaunchKernelGGL(gn_chunk_finalize64_kernel<T>, dim3(ceil_div(C, 64), N), dim3(256), 0, st, (const T*)x, ldx, HW, C, pl.Cp, groups,
                                                pl.nsplit, scr, stat);
                  else hipLaunchKernelGGL(gn_chunk_finalize_kernel<T>, dim3(N), dim3(256), 0, st, (const T*)x, ldx, HW, C, pl.Cp, groups, pl.nsplit, scr, stat);
                  hipLaunchKernelGGL(gn_chunk_apply_kernel<T>, dim3(pl.colgroups, ceil_div(HW, pl.rows_per_block), N), dim3(256), 0, st, (const T*)x, ldx,
                                     (T*)y, ldy, HW, C, groups, gamma, beta, relu, stat, pl.rows_per_block);)
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(gn_fwd_kernel<T>, dim3(groups, N), dim3(GN_THREADS), 0, st, (const T*)x, ldx, (T*)y, ldy, HW, C,
                                           groups, gamma, beta, relu, save_mean_rstd);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_gn_bwd(const void* x, int ldx, const void* y, const void* dy, int ldy, void* dx, int lddx, int N, int HW, int C, int groups,
                           int dtype, const float* gamma, const float* save_mean_rstd, int relu, int accumulate, float* dgamma, float* dbeta,
                           void* workspace, void* stream) {
    ODTK_REQUIRE(x && dy && dx && gamma && save_mean_rstd && dgamma && dbeta && workspace, "gn_bwd: null pointer");
    ODTK_REQUIRE(!relu || y, "gn_bwd: relu needs y");
    ODTK_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && ldx >= C && ldy >= C && lddx >= C, "gn_bwd: N=%d HW=%d C=%d groups=%d", N, HW, C, groups);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    const GnPlan pl = gn_plan(N, HW, C, dtype, {ldx, ldy, lddx}, {x, dy, dx, relu ? y : x});
    if (const int cl = gn_small_cl(pl, HW, C, groups)) {
#define ODTK_GN_SMALL_BWD(CLV)                                                                                                                      \
    DT_SWITCH(dtype, T, hipLaunchKernelGGL((gn_small_bwd_kernel<T, CLV>), dim3(C / (pl.kc * CLV), N), dim3(512), 0, st, (const T*)x, ldx, (const T*)y,       \
                                           (const T*)dy, ldy, (T*)dx, lddx, HW, C, groups, gamma, save_mean_rstd, relu, accumulate & 1, part);)
        if (cl == 1) { ODTK_GN_SMALL_BWD(1) } else if (cl == 2) { ODTK_GN_SMALL_BWD(2) } else if (cl == 4) { ODTK_GN_SMALL_BWD(4) } else { ODTK_GN_SMALL_BWD(8) }
#undef ODTK_GN_SMALL_BWD
    } else if (pl.chunked) {
        float* scr = nullptr;
        const size_t wsn = (size_t)N * pl.nsplit * 2 * pl.Cp, stn = (size_t)N * groups * 2;
        if (int e = gn_scratch((wsn + stn) * sizeof(float), &scr)) return e;
        float* stat = scr + wsn;
        const bool fin64 = 64 % (C / groups) == 0;
        DT_SWITCH(dtype, T,
                  hipLaunchKernelGGL(gn_chunk_bwd_stats_kernel<T>, dim3(pl.colgroups, pl.nsplit, N), dim3(256), 0, st, (const T*)x, ldx, (const T*)y,
                                     (const T*)dy, ldy, HW, C, pl.Cp, groups, save_mean_rstd, relu, pl.rows_per_split, scr);
                  if (fin64) hipLaunchKernelGGL(gn_chunk_bwd_finalize64_kernel, dim3(ceil_div(C, 64), N), dim3(256), 0, st, HW, C, pl.Cp, groups, pl.nsplit, scr,
                                                gamma, part, stat);
                  else hipLaunchKernelGGL(gn_chunk_bwd_finalize_kernel, dim3(N), dim3(256), 0, st, HW, C, pl.Cp, groups, pl.nsplit, scr, gamma, part, stat);
                  hipLaunchKernelGGL(gn_chunk_bwd_apply_kernel<T>, dim3(pl.colgroups, ceil_div(HW, pl.rows_per_block), N), dim3(256), 0, st, (const T*)x,
                                     ldx, (const T*)y, (const T*)dy, ldy, (T*)dx, lddx, HW, C, groups, gamma, save_mean_rstd, stat, relu, accumulate & 1,
                                     pl.rows_per_block);)
    } else {
        DT_SWITCH(dtype, T, hipLaunchKernelGGL(gn_bwd_kernel<T>, dim3(groups, N), dim3(GN_THREADS), 0, st, (const T*)x, ldx, (const T*)y, (const T*)dy, ldy,
                                               (T*)dx, lddx, HW, C, groups, gamma, save_mean_rstd, relu, accumulate & 1, part);)
    }
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, part, N, C, dgamma, dbeta, accumulate & 2);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// tf.exp on the distance outputs of the FCOS regression heads (FCOS.py:363) while they are laid out for the loss kernel, and its
// chain rule: y[m][c] = exp(x[m][c]) as f32 [M][C]; dx[m][c] = dy[m][c] * y[m][c] back in the conv rows (pad columns zeroed).
namespace odtk {
namespace {

template <typename T>
__global__ void __launch_bounds__(256) exp_rows_to_f32_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, long long M, int C) {
    const long long total = M * C, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / C;
        y[i] = expf(elem<T>::load(x[m * ldx + (int)(i - m * C)]));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) exp_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, T* __restrict__ dx, int lddx,
                                                           long long M, int C) {
    const long long total = M * lddx, step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long long m = i / lddx;
        const int c = (int)(i - m * lddx);
        dx[i] = elem<T>::store(c < C ? dy[m * C + c] * y[m * C + c] : 0.f);
    }
}

}  // namespace
}  // namespace odtk

extern "C" int odtk_exp_rows_to_f32(const void* x, int ldx, int dtype, float* y, long long M, int C, void* stream) {
    ODTK_REQUIRE(x && y && M > 0 && C > 0 && ldx >= C, "exp_rows_to_f32: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(exp_rows_to_f32_kernel<T>, dim3(grid_for(M * C, 256, 65536)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                           ldx, y, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_exp_rows_bwd(const float* dy, const float* y, void* dx, int lddx, int dtype, long long M, int C, void* stream) {
    ODTK_REQUIRE(dy && y && dx && M > 0 && C > 0 && lddx >= C, "exp_rows_bwd: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(exp_rows_bwd_kernel<T>, dim3(grid_for(M * lddx, 256, 65536)), dim3(256), 0, (hipStream_t)stream, dy, y, (T*)dx,
                                           lddx, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
