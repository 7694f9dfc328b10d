// RefineDet box side (reference RefineDet.py:422-567 loss, :189-206 decode): the two-stage ARM -> ODM loss with both gradients, and the
// two-stage decode.  Matching is RetinaNet's (odtk_retina_match: best anchor per box, positive > 0.5, negative < 0.4), the ARM's hard negatives are
// mined by the SSD300 path (odtk_softmax_ce_const -> odtk_nms_batched); this file adds what is new: one pass over the positive rows and the mined
// negatives that evaluates six loss terms and scatters four gradients.  Latency-bound (a few thousand rows per image): LOSS_SPLIT workgroups per
// image, fixed-order partial sums (deterministic loss), float atomics only for the gradients (duplicate best anchors hit one row twice).
#include "common.h"

namespace odtk {
namespace {

constexpr int RD_THREADS = 256;
constexpr int RD_SPLIT = 8;
constexpr int RD_MAXC = 96;

__device__ __forceinline__ float rd_block_sum(float v, float* sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < RD_THREADS / 64; ++i) t += sm[i];
    return t;
}

struct RdArgs {
    const float* arm_loc; const float* arm_conf; const float* odm_loc; const float* odm_conf;
    int N, A, C;
    const float* yx; const float* hw; const float* gt; int P;
    const int* ngt; const int* best; const unsigned char* status; const int* rgindex; const int* counts;
    const float* negloss; const int* sel_idx; int sel_cap; const int* sel_cnt;
    float grad_scale;
    float* loss_parts;                       // [N][8]: neg_arm, pos_armconf, pos_coord_arm, neg_odm, pos_odmconf, pos_coord_odm, total, #odm negatives
    float* d_arm_loc; float* d_arm_conf; float* d_odm_loc; float* d_odm_conf;
    float* parts;                            // [N][RD_SPLIT][8]
};

__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float sl1g(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

// softmax cross entropy of row z [C] against `label`; adds (softmax - onehot) * g to dz
__device__ __forceinline__ float ce_row(const float* z, float* dz, int C, int label, float g) {
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float e[RD_MAXC];
    float s = 0.f;
    for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - m); s += e[c]; }
    for (int c = 0; c < C; ++c) atomicAdd(dz + c, (e[c] / s - (c == label ? 1.f : 0.f)) * g);
    return logf(s) - (z[label] - m);
}

// one row of the positive set: anchor `a`, ground-truth box `g` (RefineDet.py:505-563)
__device__ __forceinline__ void rd_positive(const RdArgs& k, int n, int a, int g, float gsc, float& armc, float& armx, float& odmc, float& odmx) {
    const size_t row = (size_t)n * k.A + a;
    const float* gb = k.gt + ((size_t)n * k.P + g) * 5;
    armc += ce_row(k.arm_conf + row * 2, k.d_arm_conf + row * 2, 2, 0, gsc);                       // ARM: class 0 = object
    odmc += ce_row(k.odm_conf + row * k.C, k.d_odm_conf + row * k.C, k.C, (int)gb[4], gsc);
    const float* al = k.arm_loc + row * 4;
    const float* ol = k.odm_loc + row * 4;
    float* dal = k.d_arm_loc + row * 4;
    float* dol = k.d_odm_loc + row * 4;
    float ax = 0.f, ox = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float ayx = k.yx[2 * a + j], ahw = k.hw[2 * a + j];
        const float gyx = gb[j], ghw = gb[2 + j];
        // ARM box loss against the anchor
        const float d_yx = al[j] - (gyx - ayx) / ahw;
        const float d_hw = al[2 + j] - logf(ghw / ahw);
        ax += sl1(d_yx) + sl1(d_hw);
        float g_ayx = sl1g(d_yx), g_ahw = sl1g(d_hw);
        // ODM box loss against the ARM-refined anchor (no stop_gradient in the reference: the targets depend on the ARM outputs)
        const float ryx = al[j] * ahw + ayx;
        const float rhw = expf(al[2 + j]) * ahw;
        const float oyx = (gyx - ryx) / rhw;
        const float e_yx = ol[j] - oyx;
        const float e_hw = ol[2 + j] - logf(ghw / rhw);
        ox += sl1(e_yx) + sl1(e_hw);
        const float s_yx = sl1g(e_yx), s_hw = sl1g(e_hw);
        atomicAdd(dol + j, s_yx * gsc);
        atomicAdd(dol + 2 + j, s_hw * gsc);
        g_ayx += s_yx * (ahw / rhw);                       // d(-oyx)/d(arm_yx) = ahw / rhw
        g_ahw += s_yx * oyx + s_hw;                        // d(-oyx)/d(arm_hw) = oyx;  d(-log(ghw / rhw))/d(arm_hw) = 1
        atomicAdd(dal + j, g_ayx * gsc);
        atomicAdd(dal + 2 + j, g_ahw * gsc);
    }
    armx += ax; odmx += ox;
}

__global__ void __launch_bounds__(RD_THREADS) refinedet_loss_kernel(const RdArgs k) {
    __shared__ float sm[RD_THREADS / 64];
    __shared__ float s_nodm;
    const int n = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int G = k.ngt[n], num_pos = k.counts[n * 4 + 0], nsel = k.sel_cnt[n], bg = k.C - 1;
    float neg_arm = 0.f, neg_odm = 0.f, nodm = 0.f;
    if (s == 0) {
        // the mined ARM negatives (a few hundred rows): ARM cross entropy against class 1, and -- for those whose ARM background LOGIT is below
        // 0.99 (sic, RefineDet.py:535) -- ODM cross entropy against the background class; the ODM mean needs the count first
        for (int i = tid; i < nsel; i += RD_THREADS) {
            const int a = k.sel_idx[(size_t)n * k.sel_cap + i];
            if (k.arm_conf[((size_t)n * k.A + a) * 2 + 1] < 0.99f) nodm += 1.f;
        }
        nodm = rd_block_sum(nodm, sm);
        if (tid == 0) s_nodm = nodm;
        __syncthreads();
        const float g_arm = k.grad_scale / (float)nsel, g_odm = k.grad_scale / s_nodm;
        for (int i = tid; i < nsel; i += RD_THREADS) {
            const int a = k.sel_idx[(size_t)n * k.sel_cap + i];
            const size_t row = (size_t)n * k.A + a;
            neg_arm += ce_row(k.arm_conf + row * 2, k.d_arm_conf + row * 2, 2, 1, g_arm);
            if (k.arm_conf[row * 2 + 1] < 0.99f) neg_odm += ce_row(k.odm_conf + row * k.C, k.d_odm_conf + row * k.C, k.C, bg, g_odm);
        }
    }
    float armc = 0.f, armx = 0.f, odmc = 0.f, odmx = 0.f;
    const float gsc = k.grad_scale / (float)num_pos;
    if (s == 0)
        for (int g = tid; g < G; g += RD_THREADS) rd_positive(k, n, k.best[(size_t)n * k.P + g], g, gsc, armc, armx, odmc, odmx);
    const int per = (k.A + RD_SPLIT - 1) / RD_SPLIT;
    const int a1 = min(k.A, (s + 1) * per);
    for (int a = s * per + tid; a < a1; a += RD_THREADS)
        if (k.status[(size_t)n * k.A + a] == 1) rd_positive(k, n, a, k.rgindex[(size_t)n * k.A + a], gsc, armc, armx, odmc, odmx);
    float v[7] = {neg_arm, armc, armx, neg_odm, odmc, odmx, 0.f};
    for (int i = 0; i < 6; ++i) v[i] = rd_block_sum(v[i], sm);
    if (tid == 0) {
        float* o = k.parts + ((size_t)n * RD_SPLIT + s) * 8;
        for (int i = 0; i < 6; ++i) o[i] = v[i];
        o[6] = s == 0 ? s_nodm : 0.f;
    }
}

__global__ void __launch_bounds__(64) refinedet_loss_final_kernel(const RdArgs k) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= k.N) return;
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < RD_SPLIT; ++s)
        for (int i = 0; i < 7; ++i) t[i] += k.parts[((size_t)n * RD_SPLIT + s) * 8 + i];
    const float np_ = (float)k.counts[n * 4 + 0], ns = (float)k.sel_cnt[n];
    float* o = k.loss_parts + (size_t)n * 8;
    o[0] = t[0] / ns; o[1] = t[1] / np_; o[2] = t[2] / np_;            // means of empty sets -> NaN, as TensorFlow
    o[3] = t[3] / t[6]; o[4] = t[4] / np_; o[5] = t[5] / np_;
    o[6] = (o[0] + o[1] + o[2]) + (o[3] + o[4] + o[5]);
    o[7] = t[6];
}

// RefineDet.py:189-206: one thread per anchor
__global__ void refinedet_decode_kernel(const float* __restrict__ arm_loc, const float* __restrict__ arm_conf, const float* __restrict__ odm_loc,
                                        const float* __restrict__ odm_conf, int A, int C, const float* __restrict__ yx, const float* __restrict__ hw,
                                        float thr, float* __restrict__ conf, float* __restrict__ boxes, unsigned char* __restrict__ keep,
                                        unsigned char* __restrict__ cand) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const float z0 = arm_conf[2 * a], z1 = arm_conf[2 * a + 1];
    const float am = fmaxf(z0, z1);
    const float e0 = expf(z0 - am), e1 = expf(z1 - am);
    const bool arm_ok = e1 / (e0 + e1) < 0.99f;
    const float* z = odm_conf + (size_t)a * C;
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float e[RD_MAXC];
    float s = 0.f;
    for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - m); s += e[c]; }
    int arg = 0; float best = -1.f;
    for (int c = 0; c < C; ++c) {
        const float p = e[c] / s;
        e[c] = p;
        if (p > best) { best = p; arg = c; }
    }
    const bool kp = arm_ok && arg < C - 1;
    keep[a] = kp ? 1 : 0;
    for (int c = 0; c < C - 1; ++c) {
        conf[(size_t)a * (C - 1) + c] = e[c];
        cand[(size_t)a * (C - 1) + c] = (kp && e[c] >= thr) ? 1 : 0;
    }
    const float* al = arm_loc + (size_t)a * 4;
    const float* ol = odm_loc + (size_t)a * 4;
    float lo[2], hi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float ayx = yx[2 * a + j], ahw = hw[2 * a + j];
        const float ryx = al[j] * ahw + ayx;
        const float rhw = expf(al[2 + j]) * ahw;
        const float oyx = ol[j] * rhw + ryx;
        const float ohw = expf(ol[2 + j]) * rhw;
        lo[j] = oyx - ohw / 2.f; hi[j] = oyx + ohw / 2.f;
    }
    boxes[4 * a + 0] = lo[0]; boxes[4 * a + 1] = lo[1]; boxes[4 * a + 2] = hi[0]; boxes[4 * a + 3] = hi[1];
}

static float* g_rd_scratch[16];
constexpr int RD_SCRATCH_IMAGES = 4096;

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" int odtk_refinedet_loss(const float* arm_loc, const float* arm_conf, const float* odm_loc, const float* odm_conf, int N, int A, int C,
                                   const float* yx, const float* hw, const float* gt, int P, const int* ngt, const int* best,
                                   const unsigned char* status, const int* rgindex, const int* counts, const float* negloss, const int* sel_idx,
                                   int sel_cap, const int* sel_cnt, float grad_scale, float* loss_parts, float* d_arm_loc, float* d_arm_conf,
                                   float* d_odm_loc, float* d_odm_conf, void* stream) {
    ODTK_REQUIRE(arm_loc && arm_conf && odm_loc && odm_conf && yx && hw && gt && ngt && best && status && rgindex && counts && negloss && sel_idx &&
                 sel_cnt && loss_parts && d_arm_loc && d_arm_conf && d_odm_loc && d_odm_conf, "refinedet_loss: null pointer");
    ODTK_REQUIRE(N > 0 && N <= RD_SCRATCH_IMAGES && A > 0 && C > 1 && C <= RD_MAXC && P > 0, "refinedet_loss: N=%d A=%d C=%d P=%d out of range", N, A, C, P);
    int dev = 0;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    ODTK_REQUIRE(dev >= 0 && dev < 16, "refinedet_loss: device %d unsupported", dev);
    if (!g_rd_scratch[dev]) ODTK_CHECK_HIP(hipMalloc((void**)&g_rd_scratch[dev], (size_t)RD_SCRATCH_IMAGES * RD_SPLIT * 8 * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    if (int e = zero_async(d_arm_loc, (size_t)N * A * 4 * sizeof(float), st)) return e;
    if (int e = zero_async(d_arm_conf, (size_t)N * A * 2 * sizeof(float), st)) return e;
    if (int e = zero_async(d_odm_loc, (size_t)N * A * 4 * sizeof(float), st)) return e;
    if (int e = zero_async(d_odm_conf, (size_t)N * A * C * sizeof(float), st)) return e;
    RdArgs k;
    k.arm_loc = arm_loc; k.arm_conf = arm_conf; k.odm_loc = odm_loc; k.odm_conf = odm_conf; k.N = N; k.A = A; k.C = C;
    k.yx = yx; k.hw = hw; k.gt = gt; k.P = P; k.ngt = ngt; k.best = best; k.status = status; k.rgindex = rgindex; k.counts = counts;
    k.negloss = negloss; k.sel_idx = sel_idx; k.sel_cap = sel_cap; k.sel_cnt = sel_cnt; k.grad_scale = grad_scale;
    k.loss_parts = loss_parts; k.d_arm_loc = d_arm_loc; k.d_arm_conf = d_arm_conf; k.d_odm_loc = d_odm_loc; k.d_odm_conf = d_odm_conf;
    k.parts = g_rd_scratch[dev];
    hipLaunchKernelGGL(refinedet_loss_kernel, dim3(N, RD_SPLIT), dim3(RD_THREADS), 0, st, k);
    hipLaunchKernelGGL(refinedet_loss_final_kernel, dim3((N + 63) / 64), dim3(64), 0, st, k);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_refinedet_decode(const float* arm_loc, const float* arm_conf, const float* odm_loc, const float* odm_conf, int A, int C,
                                     const float* yx, const float* hw, float score_threshold, float* conf, float* boxes, unsigned char* keep,
                                     unsigned char* cand, void* stream) {
    ODTK_REQUIRE(arm_loc && arm_conf && odm_loc && odm_conf && yx && hw && conf && boxes && keep && cand, "refinedet_decode: null pointer");
    ODTK_REQUIRE(A > 0 && C > 1 && C <= RD_MAXC, "refinedet_decode: A=%d C=%d out of range", A, C);
    hipLaunchKernelGGL(refinedet_decode_kernel, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream, arm_loc, arm_conf, odm_loc, odm_conf, A, C, yx,
                       hw, score_threshold, conf, boxes, keep, cand);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
