// The three element-wise pieces CenterNet's network needs beside the shared conv / batch-norm / pooling kernels
// (reference CenterNet.py): the normalising input transform (:51-65), tf.layers.average_pooling2d 2x2 / stride 2 (:423-431)
// and tf.train.AdamOptimizer (:154).  HBM-bound streaming kernels: 16 bytes per lane, every element touched once.
#include "common.h"

namespace odtk {
namespace {

#define DT_SWITCH(dtype, T, ...)                                         \
    if ((dtype) == ODTK_BF16) { typedef bf16_t T; __VA_ARGS__ }          \
    else if ((dtype) == ODTK_F32) { typedef float T; __VA_ARGS__ }       \
    else { set_error("bad dtype %d", (int)(dtype)); return ODTK_ERR_ARG; }

inline int grid_for(long long total, int threads, int cap = 65536) {
    long long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// (images / div - mean) / std in exactly this float32 order (CenterNet.py:63: (self.images / 255. - mean) / std)
template <typename T>
__global__ void preprocess_norm_kernel(const float* __restrict__ img, long long pixels, float div, float m0, float m1, float m2,
                                       float s0, float s1, float s2, int ldx, T* __restrict__ x) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < pixels; i += step) {
        const float r = (img[i * 3 + 0] / div - m0) / s0, g = (img[i * 3 + 1] / div - m1) / s1, b = (img[i * 3 + 2] / div - m2) / s2;
        T* o = x + i * ldx;
        o[0] = elem<T>::store(r); o[1] = elem<T>::store(g); o[2] = elem<T>::store(b);
        for (int c = 3; c < ldx; ++c) o[c] = elem<T>::store(0.f);
    }
}

// 2x2 / stride 2 average pooling on even maps (SAME = no padding): one thread per output element column chunk
template <typename T, bool BWD>
__global__ void avgpool2x2_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W, int ld) {
    // forward: src [N,H,W,ld] -> dst [N,H/2,W/2,ld];   backward: src = dy [N,H/2,W/2,ld] -> dst = dx [N,H,W,ld], each input gets dy / 4
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * ld;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const int c = (int)(i % ld);
        long long pix = i / ld;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        const size_t in00 = (((size_t)n * H + 2 * ho) * W + 2 * wo) * ld + c;
        const size_t rs = (size_t)W * ld;
        if (!BWD) {
            // TF's AvgPool: sum of the window in scan order, divided by the window size
            const float s = ((elem<T>::load(src[in00]) + elem<T>::load(src[in00 + ld])) + elem<T>::load(src[in00 + rs])) + elem<T>::load(src[in00 + rs + ld]);
            dst[i] = elem<T>::store(s / 4.f);
        } else {
            const T v = elem<T>::store(elem<T>::load(src[i]) / 4.f);
            dst[in00] = v; dst[in00 + ld] = v; dst[in00 + rs] = v; dst[in00 + rs + ld] = v;
        }
    }
}

// y = relu(a + b) on [M][ld] rows (RefineDet's transfer-connection block: tf.nn.relu(conv2 + dconv), RefineDet.py:371), and the gradient of a
// ReLU from its OUTPUT: dx (+)= dy where y > 0
template <typename T>
__global__ void add_relu_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, T* __restrict__ y, int ldy, long long M, int C) {
    const long long total = M * C;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const long long m = i / C; const int c = (int)(i - m * C);
        const float v = elem<T>::load(a[m * lda + c]) + elem<T>::load(b[m * ldb + c]);
        y[m * ldy + c] = elem<T>::store(fmaxf(v, 0.f));
    }
}

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, int ldy, T* __restrict__ dx, int lddx, long long M, int C, int accumulate) {
    const long long total = M * C;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const long long m = i / C; const int c = (int)(i - m * C);
        float g = elem<T>::load(y[m * ldy + c]) > 0.f ? elem<T>::load(dy[m * ldy + c]) : 0.f;
        if (accumulate) g += elem<T>::load(dx[m * lddx + c]);
        dx[m * lddx + c] = elem<T>::store(g);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PER_BLOCK = ADAM_THREADS * 4 * 8;     // the block size of odtk_sgd_blocks: the two optimizers share the l2_partial layout

// tf.train.AdamOptimizer (training/adam.py, kernels/training_ops.cc ApplyAdam), g = grad * gscale + wd * p (the L2 term is part of the loss):
//   m += (g - m) (1 - b1);  v += (g g - v) (1 - b2);  p -= (m lr_t) / (sqrt(v) + eps)   -- ApplyAdam's own float32 expression order;
//   lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) comes from the host
template <typename TC>
__global__ void __launch_bounds__(ADAM_THREADS) adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ g, long long n, float lr_t, float b1, float b2, float eps,
                                                            float wd, float gscale, float* __restrict__ l2_partial, TC* __restrict__ pc) {
    __shared__ float sm[ADAM_THREADS / 64];
    const long long base = (long long)blockIdx.x * ADAM_PER_BLOCK;
    float ss = 0.f;
    for (int it = 0; it < 8; ++it) {
        const long long i0 = base + ((long long)it * ADAM_THREADS + threadIdx.x) * 4;
        for (long long j = i0; j < n && j < i0 + 4; ++j) {
            float pv = p[j];
            ss += pv * pv;
            const float gv = g[j] * gscale + wd * pv;
            float mv = m[j], vv = v[j];
            mv += (gv - mv) * (1.f - b1);
            vv += (gv * gv - vv) * (1.f - b2);
            pv -= (mv * lr_t) / (sqrtf(vv) + eps);
            p[j] = pv; m[j] = mv; v[j] = vv;
            if (pc) pc[j] = elem<TC>::store(pv);
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0 && l2_partial) l2_partial[blockIdx.x] = 0.5f * ((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" int odtk_preprocess_norm(const float* images, long long pixels, float div, const float* mean3, const float* std3, int ldx, int dtype,
                                    void* x, void* stream) {
    ODTK_REQUIRE(images && x && mean3 && std3 && ldx >= 3 && div != 0.f, "preprocess_norm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(preprocess_norm_kernel<T>, dim3(grid_for(pixels, 256, 8192)), dim3(256), 0, st, images, pixels, div,
                                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], ldx, (T*)x);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_avgpool2x2_fwd(const void* x, void* y, int N, int H, int W, int ld, int dtype, void* stream) {
    ODTK_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && ld > 0, "avgpool2x2_fwd: bad argument");
    ODTK_REQUIRE(H % 2 == 0 && W % 2 == 0, "avgpool2x2: H=%d W=%d must be even (SAME padding of odd maps is not implemented)", H, W);
    const long long total = (long long)N * (H / 2) * (W / 2) * ld;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL((avgpool2x2_kernel<T, false>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)x, (T*)y, N, H, W, ld);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_avgpool2x2_bwd(const void* dy, void* dx, int N, int H, int W, int ld, int dtype, void* stream) {
    ODTK_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && ld > 0, "avgpool2x2_bwd: bad argument");
    ODTK_REQUIRE(H % 2 == 0 && W % 2 == 0, "avgpool2x2: H=%d W=%d must be even", H, W);
    const long long total = (long long)N * (H / 2) * (W / 2) * ld;
    DT_SWITCH(dtype, T, hipLaunchKernelGGL((avgpool2x2_kernel<T, true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)dy, (T*)dx, N, H, W, ld);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_add_relu_fwd(const void* a, int lda, const void* b, int ldb, void* y, int ldy, long long M, int C, int dtype, void* stream) {
    ODTK_REQUIRE(a && b && y && M > 0 && C > 0 && lda >= C && ldb >= C && ldy >= C, "add_relu_fwd: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(add_relu_kernel<T>, dim3(grid_for(M * C, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, (const T*)b, ldb,
                                           (T*)y, ldy, M, C);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_relu_bwd(const void* y, const void* dy, int ldy, void* dx, int lddx, long long M, int C, int dtype, int accumulate, void* stream) {
    ODTK_REQUIRE(y && dy && dx && M > 0 && C > 0 && ldy >= C && lddx >= C, "relu_bwd: bad argument");
    DT_SWITCH(dtype, T, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(grid_for(M * C, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)dy, ldy,
                                           (T*)dx, lddx, M, C, accumulate);)
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_adam(float* p, float* m, float* v, const float* grad, long long n, float lr_t, float beta1, float beta2, float eps,
                         float wd, float grad_scale, float* l2_partial, void* p_cast, int cast_dtype, void* stream) {
    ODTK_REQUIRE(p && m && v && grad && n > 0, "adam: bad argument");
    const int blocks = (int)((n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    if (p_cast && cast_dtype == ODTK_F32)
        hipLaunchKernelGGL(adam_kernel<float>, dim3(blocks), dim3(ADAM_THREADS), 0, st, p, m, v, grad, n, lr_t, beta1, beta2, eps, wd, grad_scale,
                           l2_partial, (float*)p_cast);
    else
        hipLaunchKernelGGL(adam_kernel<bf16_t>, dim3(blocks), dim3(ADAM_THREADS), 0, st, p, m, v, grad, n, lr_t, beta1, beta2, eps, wd, grad_scale,
                           l2_partial, (bf16_t*)p_cast);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
