// Dense detection heads (gfx950): CenterNet, FCOS and YOLOv3 loss (forward + gradients) and inference decode.
// SURVEY.md 8(f).1 / kernels K17-K20.  HBM-bound elementwise / small-reduction work: one thread per heat-map
// element (coalesced over the channel-minor [N][H][W][C] layout), ground truth staged in LDS, per-workgroup
// partial sums reduced in a fixed order (deterministic losses), no atomics on the loss path.
//
// Reference: CenterNet.py:187-270 (loss, gaussian radius), :159-185 (decode);
//            FCOS.py:153-189 (level assignment), :266-348 (per-level loss), :197-246 (decode candidates);
//            YOLOv3.py:117-310 (loss loop), :320-350 (decode candidates).
// The CPU restatements they are tested against: oracle/centernet_ref.py, oracle/fcos_ref.py, oracle/yolov3_ref.py.
#include "common.h"
#include <math.h>

namespace odtk {
namespace {

constexpr int DH_THREADS = 256, DH_MAX_GT = 128, DH_MAXC = 64;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float log_sigmoidf_(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// first index of the minimum of gt[:, 0] (tf.argmin(ground_truth, axis=0)[0]: the first pad row, yc = -1)
__device__ __forceinline__ int first_argmin_col0(const float* gt, int P) {
    int best = 0;
    float bv = gt[0];
    for (int p = 1; p < P; ++p) {
        const float v = gt[p * 5];
        if (v < bv) { bv = v; best = p; }
    }
    return best;
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// =======================================================================================================
// CenterNet
// =======================================================================================================
struct CnGt { float gy, gx; int cy, cx, cls; };

struct CnArgs {
    const float *keypoints, *offset, *size, *gt;
    int N, H, W, C, P;
    float stride, grad_scale;
    float* info;          // [N][2 + 4]  G, sigma, (unused)
    float* parts;         // [N][nblk][2]  pos / neg partial sums
    int nblk;
    float *loss_parts, *d_keypoints, *d_offset, *d_size;
};

// per image: G, sigma (CenterNet.py:254-270: ONE scalar = min over boxes and roots), the offset / size L1 losses
// and their sparse gradients (d_offset / d_size were zeroed by the caller of this kernel)
__global__ void __launch_bounds__(64) centernet_prep_kernel(const CnArgs a) {
    const int n = blockIdx.x, t = threadIdx.x;
    const float* gt = a.gt + (size_t)n * a.P * 5;
    __shared__ int sG;
    if (t == 0) sG = first_argmin_col0(gt, a.P);
    __syncthreads();
    const int G = sG;
    float rmin = INFINITY, l_off = 0.f, l_size = 0.f;
    for (int g = t; g < G; g += 64) {
        const float h = gt[g * 5 + 2] / a.stride, w = gt[g * 5 + 3] / a.stride;
        const float mo = 0.7f;
        const float b1 = h + w, c1 = w * h * (1.f - mo) / (1.f + mo);
        const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * 1.f * c1)) / 2.f;
        const float b2 = 2.f * (h + w), c2 = (1.f - mo) * w * h;
        const float r2 = (b2 + sqrtf(b2 * b2 - 4.f * 4.f * c2)) / 2.f;
        const float a3 = 4.f * mo, b3 = -2.f * mo * (h + w), c3 = (mo - 1.f) * w * h;
        const float r3 = (b3 + sqrtf(b3 * b3 - 4.f * a3 * c3)) / 2.f;
        rmin = fminf(rmin, fminf(r1, fminf(r2, r3)));
        // offset / size regression at the centre cell (CenterNet.py:195-207): mean |.| over [G, 2]
        const float y = gt[g * 5 + 0] / a.stride, x = gt[g * 5 + 1] / a.stride;
        const float fy = floorf(y), fx = floorf(x);
        const int cy = (int)fy, cx = (int)fx;
        if (cy >= 0 && cy < a.H && cx >= 0 && cx < a.W) {
            const size_t cell = ((size_t)n * a.H + cy) * a.W + cx;
            const float gsc = a.grad_scale / (2.f * (float)G);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float og = (k == 0 ? y - fy : x - fx), op = a.offset[cell * 2 + k];
                const float zg = gt[g * 5 + 2 + k] / a.stride, zp = a.size[cell * 2 + k];
                l_off += fabsf(og - op);
                l_size += fabsf(zg - zp);
                const float so = op > og ? 1.f : (op < og ? -1.f : 0.f), sz = zp > zg ? 1.f : (zp < zg ? -1.f : 0.f);
                atomicAdd(a.d_offset + cell * 2 + k, so * gsc);
                atomicAdd(a.d_size + cell * 2 + k, 0.1f * sz * gsc);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmin = fminf(rmin, __shfl_xor(rmin, o));
        l_off += __shfl_xor(l_off, o);
        l_size += __shfl_xor(l_size, o);
    }
    if (t == 0) {
        a.info[n * 4 + 0] = (float)G;
        a.info[n * 4 + 1] = rmin;
        a.info[n * 4 + 2] = l_off / (2.f * (float)G);
        a.info[n * 4 + 3] = l_size / (2.f * (float)G);
    }
}

// penalty-reduced focal loss over the heat map (CenterNet.py:211-251), one thread per (pixel, class) element
__global__ void __launch_bounds__(DH_THREADS) centernet_heat_kernel(const CnArgs a) {
    __shared__ CnGt sg[DH_MAX_GT];
    __shared__ float red[DH_THREADS / 64];
    const int n = blockIdx.y;
    const int G = (int)a.info[n * 4 + 0];
    const float sigma = a.info[n * 4 + 1];
    const float* gt = a.gt + (size_t)n * a.P * 5;
    for (int g = threadIdx.x; g < G; g += DH_THREADS) {
        CnGt v;
        v.gy = gt[g * 5 + 0] / a.stride; v.gx = gt[g * 5 + 1] / a.stride;
        v.cy = (int)floorf(v.gy); v.cx = (int)floorf(v.gx);
        v.cls = (int)gt[g * 5 + 4];
        sg[g] = v;
    }
    __syncthreads();
    const int per = a.H * a.W * a.C;
    const int e = blockIdx.x * DH_THREADS + threadIdx.x;
    float pos = 0.f, neg = 0.f;
    if (e < per) {
        const int c = e % a.C, px = e / a.C;
        const int y = px / a.W, x = px - y * a.W;
        const float fy = (float)y, fx = (float)x;
        const float two_s2 = 2.f * (sigma * sigma);
        float redu = 0.f;
        bool hit = false;
        for (int g = 0; g < G; ++g) {
            if (sg[g].cls != c) continue;
            const float dy = sg[g].gy - fy, dx = sg[g].gx - fx;
            redu = fmaxf(redu, expf(-(dy * dy + dx * dx) / two_s2));
            hit = hit || (sg[g].cy == y && sg[g].cx == x);
        }
        const size_t at = (size_t)n * per + e;
        const float k = a.keypoints[at];
        const float s = sigmoidf_(k), ls = log_sigmoidf_(k), om = 1.f - s;
        float grad;
        if (hit) {
            pos = -(om * om) * ls;
            grad = 2.f * s * om * om * ls - om * om * om;
        } else {
            const float w1 = 1.f - redu, w = (w1 * w1) * (w1 * w1);
            const float l1m = -k + ls;                                     // log(1 - sigmoid(k))
            neg = -w * (s * s) * l1m;
            grad = -w * (2.f * s * s * om * l1m - s * s * s);
        }
        a.d_keypoints[at] = grad * a.grad_scale / (float)G;
    }
    const float tp = block_sum<DH_THREADS>(pos, red);
    const float tn = block_sum<DH_THREADS>(neg, red);
    if (threadIdx.x == 0) {
        a.parts[((size_t)n * a.nblk + blockIdx.x) * 2 + 0] = tp;
        a.parts[((size_t)n * a.nblk + blockIdx.x) * 2 + 1] = tn;
    }
}

// loss_parts[n] = { keypoint loss, offset loss, size loss, total }  (CenterNet.py:208, :250)
__global__ void __launch_bounds__(DH_THREADS) centernet_final_kernel(const CnArgs a) {
    __shared__ float red[DH_THREADS / 64];
    const int n = blockIdx.x;
    float p = 0.f, q = 0.f;
    for (int b = threadIdx.x; b < a.nblk; b += DH_THREADS) {
        p += a.parts[((size_t)n * a.nblk + b) * 2 + 0];
        q += a.parts[((size_t)n * a.nblk + b) * 2 + 1];
    }
    const float tp = block_sum<DH_THREADS>(p, red);
    const float tn = block_sum<DH_THREADS>(q, red);
    if (threadIdx.x == 0) {
        const float G = a.info[n * 4 + 0];
        const float kl = tp / G + tn / G, ol = a.info[n * 4 + 2], sl = a.info[n * 4 + 3];
        a.loss_parts[n * 4 + 0] = kl; a.loss_parts[n * 4 + 1] = ol; a.loss_parts[n * 4 + 2] = sl;
        a.loss_parts[n * 4 + 3] = kl + 0.1f * sl + ol;
    }
}

// ---- decode (CenterNet.py:159-185), one image
__global__ void __launch_bounds__(DH_THREADS) centernet_score_kernel(const float* __restrict__ keypoints, int HW, int C,
                                                                     float* __restrict__ score, int* __restrict__ cls) {
    const int p = blockIdx.x * DH_THREADS + threadIdx.x;
    if (p >= HW) return;
    float best = sigmoidf_(keypoints[(size_t)p * C]);
    int bc = 0;
    for (int c = 1; c < C; ++c) {
        const float v = sigmoidf_(keypoints[(size_t)p * C + c]);
        if (v > best) { best = v; bc = c; }                 // tf.argmax: first maximum
    }
    score[p] = best; cls[p] = bc;
}

// 3x3 peak test, threshold, top-k (descending score, lower index first): one workgroup, bitonic sort in LDS
__global__ void __launch_bounds__(1024) centernet_topk_kernel(const float* __restrict__ score, const int* __restrict__ cls,
                                                              const float* __restrict__ offset, const float* __restrict__ size,
                                                              int H, int W, float stride, float thr, int top_k,
                                                              float* __restrict__ out_scores, float* __restrict__ out_bbox,
                                                              int* __restrict__ out_cls, int* __restrict__ out_count) {
    extern __shared__ unsigned long long keys[];
    __shared__ int cnt;
    const int HW = H * W;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 1024) {
        const int y = p / W, x = p - y * W;
        const float s = score[p];
        float peak = s;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) peak = fmaxf(peak, score[yy * W + xx]);
            }
        if (s == peak && s > thr) {
            const int at = atomicAdd(&cnt, 1);
            keys[at] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)p);    // s > 0
        }
    }
    __syncthreads();
    const int n = cnt;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + threadIdx.x; i < n2; i += 1024) keys[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const int kk = n < top_k ? n : top_k;
    if (threadIdx.x == 0) *out_count = kk;
    for (int i = threadIdx.x; i < kk; i += 1024) {
        const unsigned long long key = keys[i];
        const int p = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int y = p / W, x = p - y * W;
        const float cy = (float)y + offset[p * 2], cx = (float)x + offset[p * 2 + 1];
        const float h = size[p * 2], w = size[p * 2 + 1];
        out_scores[i] = __uint_as_float((unsigned)(key >> 32));
        out_bbox[i * 4 + 0] = (cy - h / 2.f) * stride; out_bbox[i * 4 + 1] = (cx - w / 2.f) * stride;
        out_bbox[i * 4 + 2] = (cy + h / 2.f) * stride; out_bbox[i * 4 + 3] = (cx + w / 2.f) * stride;
        out_cls[i] = cls[p];
    }
}

// =======================================================================================================
// FCOS
// =======================================================================================================
constexpr int FC_LEVELS = 5;
struct FcGt { float y1, x1, y2, x2; int cls; };

struct FcArgs {
    const float* conf[FC_LEVELS]; const float* reg[FC_LEVELS]; const float* center[FC_LEVELS];
    float* d_conf[FC_LEVELS]; float* d_reg[FC_LEVELS]; float* d_center[FC_LEVELS];
    int H[FC_LEVELS], W[FC_LEVELS], nblk[FC_LEVELS], boff[FC_LEVELS + 1];
    float stride[FC_LEVELS];
    const float* gt;
    int N, C, P;
    float grad_scale;
    float* parts;         // [N][total blocks][4]  iou, heat, centre, sum(heatmap_gt)
    float* lsum;          // [N][5][4]  reduced, + [N][5] active flag in slot 3's sign? (see fcos_reduce_kernel)
    float* loss;          // [N]
};

// boxes of image n that train level l (FCOS.py:158-163), scaled by the level stride (FCOS.py:268-276)
__device__ __forceinline__ int fcos_stage_gt(const FcArgs& a, int n, int l, FcGt* sg) {
    __shared__ int sM;
    const float* gt = a.gt + (size_t)n * a.P * 5;
    if (threadIdx.x == 0) {
        const int G = first_argmin_col0(gt, a.P);
        const float lo = l == 0 ? -1.f : 64.f * (float)(1 << (l - 1)), hi = l == 4 ? INFINITY : 64.f * (float)(1 << l);
        int m = 0;
        for (int g = 0; g < G; ++g) {
            const float sz = sqrtf(gt[g * 5 + 2] * gt[g * 5 + 3]);
            const bool in = (l == 0 ? sz <= hi : (l == 4 ? sz >= lo : (sz >= lo && sz <= hi)));
            if (!in) continue;
            const float s = a.stride[l];
            const float y = gt[g * 5] / s, x = gt[g * 5 + 1] / s, h = gt[g * 5 + 2] / s, w = gt[g * 5 + 3] / s;
            FcGt v;
            v.y1 = y - h / 2.f; v.y2 = y + h / 2.f; v.x1 = x - w / 2.f; v.x2 = x + w / 2.f; v.cls = (int)gt[g * 5 + 4];
            sg[m++] = v;
        }
        sM = m;
    }
    __syncthreads();
    return sM;
}

struct FcTarget { float dl, dr, dt, db, loc; unsigned long long hm; };

// FCOS.py:277-305 and :329-342 for one location
__device__ __forceinline__ FcTarget fcos_target(const FcGt* sg, int M, float gy, float gx) {
    FcTarget t;
    t.hm = 0ull;
    float amin = INFINITY;
    bool any = false;
    for (int g = 0; g < M; ++g) {
        const float dl = gx - sg[g].x1, dr = sg[g].x2 - gx, dt = gy - sg[g].y1, db = sg[g].y2 - gy;
        if (dt > 0.f && db > 0.f && dl > 0.f && dr > 0.f) {
            any = true;
            t.hm |= 1ull << sg[g].cls;
            amin = fminf(amin, (dl + dr) * (dt + db));
        }
    }
    t.loc = any ? 1.f : 0.f;
    t.dl = t.dr = t.dt = t.db = 0.f;
    if (any)
        for (int g = 0; g < M; ++g) {
            const float dl = gx - sg[g].x1, dr = sg[g].x2 - gx, dt = gy - sg[g].y1, db = sg[g].y2 - gy;
            if (dt > 0.f && db > 0.f && dl > 0.f && dr > 0.f && (dl + dr) * (dt + db) == amin) {
                t.dl = fmaxf(t.dl, dl); t.dr = fmaxf(t.dr, dr); t.dt = fmaxf(t.dt, dt); t.db = fmaxf(t.db, db);
            }
        }
    return t;
}

// PASS 0: loss sums per workgroup;  PASS 1: gradients (needs the per-level 1 / sum(heatmap_gt) of pass 0)
template <int PASS>
__global__ void __launch_bounds__(DH_THREADS) fcos_level_kernel(const FcArgs a) {
    __shared__ FcGt sg[DH_MAX_GT];
    __shared__ float red[DH_THREADS / 64];
    const int n = blockIdx.y;
    int l = 0;
    while (l + 1 < FC_LEVELS && (int)blockIdx.x >= a.boff[l + 1]) ++l;
    const int b = blockIdx.x - a.boff[l];
    const int M = fcos_stage_gt(a, n, l, sg);
    const int H = a.H[l], W = a.W[l], C = a.C, HW = H * W;
    const int p = b * DH_THREADS + threadIdx.x;
    const bool live = p < HW;
    float s_iou = 0.f, s_heat = 0.f, s_cen = 0.f, s_hm = 0.f;
    float inv = 0.f;
    if (PASS == 1) inv = M > 0 ? a.grad_scale / a.lsum[((size_t)n * FC_LEVELS + l) * 4 + 3] : 0.f;
    if (live && (M > 0 || PASS == 1)) {
        const size_t px = (size_t)n * HW + p;
        if (M == 0) {                                   // level without boxes: contributes 0 (tf.cond, FCOS.py:165-188)
            for (int c = 0; c < C; ++c) a.d_conf[l][px * C + c] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) a.d_reg[l][px * 4 + k] = 0.f;
            a.d_center[l][px] = 0.f;
        } else {
            const int y = p / W, x = p - y * W;
            const FcTarget t = fcos_target(sg, M, (float)y, (float)x);
            const float4 pr = *reinterpret_cast<const float4*>(a.reg[l] + px * 4);      // l, r, t, b
            const float mnl = fminf(t.dl, pr.x), mnr = fminf(t.dr, pr.y), mnt = fminf(t.dt, pr.z), mnb = fminf(t.db, pr.w);
            const float iw = mnl + mnr, ih = mnt + mnb;
            const float inter = iw * ih;
            const float ag = (t.dl + t.dr) * (t.dt + t.db), ap = (pr.x + pr.y) * (pr.z + pr.w);
            const float uni = ag + ap - inter;
            const float iou = inter / (uni + 1e-12f);
            const float lrmin = fminf(t.dl, t.dr), tbmin = fminf(t.dt, t.db), lrmax = fmaxf(t.dl, t.dr), tbmax = fmaxf(t.dt, t.db);
            const float cgt = sqrtf(lrmin * tbmin / (lrmax * tbmax + 1e-12f));
            const float cx = a.center[l][px];
            if (PASS == 0) {
                s_iou = -logf(iou + 1e-12f) * t.loc;
                s_cen = fmaxf(cx, 0.f) - cx * cgt + log1pf(expf(-fabsf(cx)));
                for (int c = 0; c < C; ++c) {
                    const float k = a.conf[l][px * C + c];
                    const float s = sigmoidf_(k), ls = log_sigmoidf_(k), om = 1.f - s;
                    if ((t.hm >> c) & 1ull) { s_heat += -.25f * (om * om) * ls; s_hm += 1.f; }
                    else s_heat += -.25f * (s * s) * (-k + ls);
                }
            } else {
                // d(-log(iou + eps)) / d pred side, iou = I / (U + eps), U = Ag + Ap - I
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t.loc > 0.f) {
                    const float den = uni + 1e-12f;
                    const float k0 = -1.f / (iou + 1e-12f) / (den * den);
                    const float dI[4] = {pr.x < t.dl ? ih : 0.f, pr.y < t.dr ? ih : 0.f, pr.z < t.dt ? iw : 0.f, pr.w < t.db ? iw : 0.f};
                    const float dA[4] = {pr.z + pr.w, pr.z + pr.w, pr.x + pr.y, pr.x + pr.y};
                    float* gp = reinterpret_cast<float*>(&g);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gp[k] = k0 * (dI[k] * den - inter * (dA[k] - dI[k])) * inv;
                }
                *reinterpret_cast<float4*>(a.d_reg[l] + px * 4) = g;
                a.d_center[l][px] = (sigmoidf_(cx) - cgt) * inv;
                for (int c = 0; c < C; ++c) {
                    const float k = a.conf[l][px * C + c];
                    const float s = sigmoidf_(k), ls = log_sigmoidf_(k), om = 1.f - s;
                    float gr;
                    if ((t.hm >> c) & 1ull) gr = .25f * (2.f * s * om * om * ls - om * om * om);
                    else gr = -.25f * (2.f * s * s * om * (-k + ls) - s * s * s);
                    a.d_conf[l][px * C + c] = gr * inv;
                }
            }
        }
    }
    if (PASS == 0) {
        const float t0 = block_sum<DH_THREADS>(s_iou, red), t1 = block_sum<DH_THREADS>(s_heat, red);
        const float t2 = block_sum<DH_THREADS>(s_cen, red), t3 = block_sum<DH_THREADS>(s_hm, red);
        if (threadIdx.x == 0) {
            float* o = a.parts + ((size_t)n * a.boff[FC_LEVELS] + blockIdx.x) * 4;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
        }
    }
}

// per (image, level) sums in workgroup order; loss[n] = sum over active levels of (iou + heat + centre) / sum(hm)
__global__ void __launch_bounds__(64) fcos_reduce_kernel(const FcArgs a) {
    __shared__ FcGt sg[DH_MAX_GT];
    const int n = blockIdx.x;
    float total = 0.f;
    for (int l = 0; l < FC_LEVELS; ++l) {
        const int M = fcos_stage_gt(a, n, l, sg);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = threadIdx.x; b < a.nblk[l]; b += 64) {
            const float* o = a.parts + ((size_t)n * a.boff[FC_LEVELS] + a.boff[l] + b) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += o[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
        if (threadIdx.x == 0) {
            float* d = a.lsum + ((size_t)n * FC_LEVELS + l) * 4;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
            if (M > 0) total += (s[0] + s[1] + s[2]) / s[3];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.loss[n] = total;
}

// decode candidates of ONE image (FCOS.py:197-246): pconf [L][C], pbbox [L][4] = y1,x1,y2,x2 px
struct FcDecArgs {
    const float* conf[FC_LEVELS]; const float* reg[FC_LEVELS]; const float* center[FC_LEVELS];
    int H[FC_LEVELS], W[FC_LEVELS], off[FC_LEVELS + 1];
    float stride[FC_LEVELS];
    int C;
    float *pconf, *pbbox;
};
__global__ void __launch_bounds__(DH_THREADS) fcos_candidates_kernel(const FcDecArgs a) {
    const int i = blockIdx.x * DH_THREADS + threadIdx.x;
    if (i >= a.off[FC_LEVELS]) return;
    int l = 0;
    while (l + 1 < FC_LEVELS && i >= a.off[l + 1]) ++l;
    const int p = i - a.off[l];
    const int y = p / a.W[l], x = p - y * a.W[l];
    const float sc = sigmoidf_(a.center[l][p]);
    for (int c = 0; c < a.C; ++c) a.pconf[(size_t)i * a.C + c] = sigmoidf_(a.conf[l][(size_t)p * a.C + c]) * sc;
    const float4 r = *reinterpret_cast<const float4*>(a.reg[l] + (size_t)p * 4);
    const float s = a.stride[l], gy = (float)y, gx = (float)x;
    *reinterpret_cast<float4*>(a.pbbox + (size_t)i * 4) = make_float4((gy - r.z) * s, (gx - r.x) * s, (gy + r.w) * s, (gx + r.y) * s);
}


// =======================================================================================================
// YOLOv3 (YOLOv3.py:117-310 loss loop, :320-350 decode candidates).  Quirks of the reference are reproduced,
// see oracle/yolov3_ref.py: unclamped intersections, no-object boxes built as y1x1 -/+ y2x2/2, head / prior /
// stride pairing, strict ">" head assignment, prior + exp(t) sizes.
// pred_l [N][H_l][W_l][P][C+5] = class(C), yx(2), hw(2), obj(1)
// =======================================================================================================
constexpr int YL_HEADS = 3, YL_MAXP = 8;
struct YlArgs {
    const float* pred[YL_HEADS]; float* d_pred[YL_HEADS];
    int H[YL_HEADS], W[YL_HEADS], nblk[YL_HEADS], boff[YL_HEADS + 1];
    float prior[YL_HEADS][YL_MAXP][2];
    float gstride[YL_HEADS], dscale[YL_HEADS];
    const float* gt;
    int N, P, C, pad;
    float coord_scale, noobj_scale, obj_scale, class_scale, grad_scale;
    float* parts;          // [N][total blocks] no-object partial sums
    float* loss_parts;     // [N][5] coord, class, obj, noobj, total
};
struct YlGt { float y1, x1, y2, x2, area; int cy, cx; };

__device__ __forceinline__ float bce_logits(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }

// no-object term: every prior of every cell that holds no ground-truth centre (YOLOv3.py:126-131, :247-306)
__global__ void __launch_bounds__(DH_THREADS) yolov3_noobj_kernel(const YlArgs a) {
    __shared__ YlGt sg[DH_MAX_GT];
    __shared__ float red[DH_THREADS / 64];
    __shared__ int sG;
    const int n = blockIdx.y;
    int l = 0;
    while (l + 1 < YL_HEADS && (int)blockIdx.x >= a.boff[l + 1]) ++l;
    const float* gt = a.gt + (size_t)n * a.pad * 5;
    if (threadIdx.x == 0) sG = first_argmin_col0(gt, a.pad);
    __syncthreads();
    const int G = sG;
    for (int g = threadIdx.x; g < G; g += DH_THREADS) {
        const float s = a.gstride[l];
        const float y = gt[g * 5] / s, x = gt[g * 5 + 1] / s, h = gt[g * 5 + 2] / s, w = gt[g * 5 + 3] / s;
        YlGt v;
        v.y1 = y - h / 2.f; v.x1 = x - w / 2.f; v.y2 = y + h / 2.f; v.x2 = x + w / 2.f;
        v.area = (v.y2 - v.y1) * (v.x2 - v.x1);
        v.cy = (int)floorf(y); v.cx = (int)floorf(x);
        sg[g] = v;
    }
    __syncthreads();
    const int H = a.H[l], W = a.W[l], E = a.C + 5;
    const int i = (blockIdx.x - a.boff[l]) * DH_THREADS + threadIdx.x;
    float loss = 0.f;
    if (i < H * W * a.P) {
        const int k = i % a.P, cell = i / a.P;
        const int cy = cell / W, cx = cell - cy * W;
        bool occupied = false;
        for (int g = 0; g < G; ++g) occupied = occupied || (sg[g].cy == cy && sg[g].cx == cx);
        if (!occupied) {
            const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f, ah = a.prior[l][k][0], aw = a.prior[l][k][1];
            const float q1y = ay - ah / 2.f, q1x = ax - aw / 2.f, q2y = ay + ah / 2.f, q2x = ax + aw / 2.f;
            const float b1y = q1y - q2y / 2.f, b1x = q1x - q2x / 2.f, b2y = q1y + q2y / 2.f, b2x = q1x + q2x / 2.f;
            const float aarea = (b2y - b1y) * (b2x - b1x);
            float mx = -INFINITY;
            for (int g = 0; g < G; ++g) {
                const float inter = (fminf(sg[g].y2, b2y) - fmaxf(sg[g].y1, b1y)) * (fminf(sg[g].x2, b2x) - fmaxf(sg[g].x1, b1x));
                mx = fmaxf(mx, inter / (aarea + sg[g].area - inter));
            }
            if (mx <= 0.5f) {
                const size_t at = (((size_t)n * H * W + cell) * a.P + k) * E + a.C + 4;
                const float x = a.pred[l][at];
                loss = bce_logits(x, 0.f);
                a.d_pred[l][at] = sigmoidf_(x) * a.noobj_scale * a.grad_scale / (float)G;
            }
        }
    }
    const float t = block_sum<DH_THREADS>(loss, red);
    if (threadIdx.x == 0) a.parts[(size_t)n * a.boff[YL_HEADS] + blockIdx.x] = t;
}

// responsible predictions: one thread per ground-truth box (YOLOv3.py:132-246), + the image's totals
__global__ void __launch_bounds__(64) yolov3_pos_kernel(const YlArgs a) {
    const int n = blockIdx.x, t = threadIdx.x;
    const float* gt = a.gt + (size_t)n * a.pad * 5;
    __shared__ int sG;
    if (t == 0) sG = first_argmin_col0(gt, a.pad);
    __syncthreads();
    const int G = sG, E = a.C + 5;
    float coord = 0.f, cls = 0.f, obj = 0.f;
    for (int g = t; g < G; g += 64) {
        float best_iou[YL_HEADS], ty[YL_HEADS], tx[YL_HEADS], th[YL_HEADS], tw[YL_HEADS];
        int best_k[YL_HEADS], cyv[YL_HEADS], cxv[YL_HEADS];
#pragma unroll
        for (int l = 0; l < YL_HEADS; ++l) {
            const float s = a.gstride[l];
            const float y = gt[g * 5] / s, x = gt[g * 5 + 1] / s, h = gt[g * 5 + 2] / s, w = gt[g * 5 + 3] / s;
            const float fy = floorf(y), fx = floorf(x);
            cyv[l] = (int)fy; cxv[l] = (int)fx;
            const float gy1 = y - h / 2.f, gx1 = x - w / 2.f, gy2 = y + h / 2.f, gx2 = x + w / 2.f;
            const float garea = (gy2 - gy1) * (gx2 - gx1);
            const float ay = fy + 0.5f, ax = fx + 0.5f;
            float bi = -INFINITY; int bk = 0;
            for (int k = 0; k < a.P; ++k) {
                const float ah = a.prior[l][k][0], aw = a.prior[l][k][1];
                const float ay1 = ay - ah / 2.f, ax1 = ax - aw / 2.f, ay2 = ay + ah / 2.f, ax2 = ax + aw / 2.f;
                const float inter = (fminf(gy2, ay2) - fmaxf(gy1, ay1)) * (fminf(gx2, ax2) - fmaxf(gx1, ax1));
                const float iou = inter / (ah * aw + garea - inter);
                if (iou > bi) { bi = iou; bk = k; }                     // tf.argmax: first maximum
            }
            best_iou[l] = bi; best_k[l] = bk;
            ty[l] = y - fy; tx[l] = x - fx;
            th[l] = logf(h / a.prior[l][bk][0]); tw[l] = logf(w / a.prior[l][bk][1]);
        }
        const int l = (best_iou[0] > best_iou[1] && best_iou[0] > best_iou[2]) ? 0
                      : (best_iou[1] > best_iou[0] && best_iou[1] > best_iou[2]) ? 1 : 2;     // :186-190
        const size_t at = (((size_t)n * a.H[l] * a.W[l] + (size_t)cyv[l] * a.W[l] + cxv[l]) * a.P + best_k[l]) * E;
        const float* p = a.pred[l] + at;
        float* d = a.d_pred[l] + at;
        const float gs = a.grad_scale / (float)G;
        const int label = (int)gt[g * 5 + 4];
        for (int c = 0; c < a.C; ++c) {
            const float z = c == label ? 1.f : 0.f;
            cls += bce_logits(p[c], z);
            atomicAdd(d + c, (sigmoidf_(p[c]) - z) * a.class_scale * gs);
        }
        const float tgt[4] = {ty[l], tx[l], th[l], tw[l]};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            coord += bce_logits(p[a.C + k], tgt[k]);
            atomicAdd(d + a.C + k, (sigmoidf_(p[a.C + k]) - tgt[k]) * a.coord_scale * gs);
            const float e = p[a.C + 2 + k] - tgt[2 + k];
            coord += 0.5f * (e * e);
            atomicAdd(d + a.C + 2 + k, e * a.coord_scale * gs);
        }
        obj += bce_logits(p[a.C + 4], 1.f);
        atomicAdd(d + a.C + 4, (sigmoidf_(p[a.C + 4]) - 1.f) * a.obj_scale * gs);
    }
    float noobj = 0.f;
    for (int b = t; b < a.boff[YL_HEADS]; b += 64) noobj += a.parts[(size_t)n * a.boff[YL_HEADS] + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        coord += __shfl_xor(coord, o); cls += __shfl_xor(cls, o); obj += __shfl_xor(obj, o); noobj += __shfl_xor(noobj, o);
    }
    if (t == 0) {
        float* o = a.loss_parts + n * 5;
        const float Gf = (float)G;
        o[0] = coord; o[1] = cls; o[2] = obj; o[3] = noobj;
        o[4] = (a.coord_scale * coord + a.class_scale * cls + a.obj_scale * obj) / Gf + a.noobj_scale * noobj / Gf;      // :307-309
    }
}

// decode candidates of ONE image: confidence [L][C], bbox [L][4]   (YOLOv3.py:320-350)
__global__ void __launch_bounds__(DH_THREADS) yolov3_candidates_kernel(const YlArgs a, float* __restrict__ conf, float* __restrict__ bbox) {
    const int i = blockIdx.x * DH_THREADS + threadIdx.x;
    int off[YL_HEADS + 1];
    off[0] = 0;
#pragma unroll
    for (int l = 0; l < YL_HEADS; ++l) off[l + 1] = off[l] + a.H[l] * a.W[l] * a.P;
    if (i >= off[YL_HEADS]) return;
    int l = 0;
    while (l + 1 < YL_HEADS && i >= off[l + 1]) ++l;
    const int r = i - off[l], E = a.C + 5;
    const int k = r % a.P, cell = r / a.P;
    const int cy = cell / a.W[l], cx = cell - cy * a.W[l];
    const float* p = a.pred[l] + (size_t)r * E;
    const float so = sigmoidf_(p[a.C + 4]);
    for (int c = 0; c < a.C; ++c) conf[(size_t)i * a.C + c] = sigmoidf_(p[c]) * so;
    const float y = ((float)cy + 0.5f) + sigmoidf_(p[a.C]), x = ((float)cx + 0.5f) + sigmoidf_(p[a.C + 1]);
    const float h = a.prior[l][k][0] + expf(p[a.C + 2]), w = a.prior[l][k][1] + expf(p[a.C + 3]);
    const float s = a.dscale[l];
    *reinterpret_cast<float4*>(bbox + (size_t)i * 4) = make_float4((y - h / 2.f) * s, (x - w / 2.f) * s, (y + h / 2.f) * s, (x + w / 2.f) * s);
}

}  // namespace
}  // namespace odtk

using namespace odtk;

// ---------------------------------------------------------------------------------------------------- CenterNet
extern "C" long long odtk_centernet_workspace_bytes(int N, int H, int W, int C) {
    const long long nblk = ((long long)H * W * C + DH_THREADS - 1) / DH_THREADS;
    return (long long)N * 4 * 4 + (long long)N * nblk * 2 * 4 + (long long)H * W * 8;
}

extern "C" int odtk_centernet_loss(const float* keypoints, const float* offset, const float* size, const float* gt, int N, int H,
                                   int W, int C, int P, float stride, float grad_scale, float* loss_parts, float* d_keypoints,
                                   float* d_offset, float* d_size, void* workspace, void* stream) {
    ODTK_REQUIRE(keypoints && offset && size && gt && loss_parts && d_keypoints && d_offset && d_size && workspace,
                 "centernet_loss: null pointer");
    ODTK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && P > 0 && P <= DH_MAX_GT, "centernet_loss: N=%d H=%d W=%d C=%d P=%d out of range", N, H, W, C, P);
    hipStream_t st = (hipStream_t)stream;
    CnArgs a;
    a.keypoints = keypoints; a.offset = offset; a.size = size; a.gt = gt; a.N = N; a.H = H; a.W = W; a.C = C; a.P = P;
    a.stride = stride; a.grad_scale = grad_scale;
    a.nblk = ceil_div(H * W * C, DH_THREADS);
    a.info = (float*)workspace;
    a.parts = a.info + (size_t)N * 4;
    a.loss_parts = loss_parts; a.d_keypoints = d_keypoints; a.d_offset = d_offset; a.d_size = d_size;
    if (int e = zero_async(d_offset, (size_t)N * H * W * 2 * sizeof(float), st)) return e;
    if (int e = zero_async(d_size, (size_t)N * H * W * 2 * sizeof(float), st)) return e;
    hipLaunchKernelGGL(centernet_prep_kernel, dim3(N), dim3(64), 0, st, a);
    hipLaunchKernelGGL(centernet_heat_kernel, dim3(a.nblk, N), dim3(DH_THREADS), 0, st, a);
    hipLaunchKernelGGL(centernet_final_kernel, dim3(N), dim3(DH_THREADS), 0, st, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_centernet_decode(const float* keypoints, const float* offset, const float* size, int H, int W, int C,
                                     float stride, float score_threshold, int top_k, float* scores, float* bbox, int* class_id,
                                     int* count, void* workspace, void* stream) {
    ODTK_REQUIRE(keypoints && offset && size && scores && bbox && class_id && count && workspace, "centernet_decode: null pointer");
    ODTK_REQUIRE(H > 0 && W > 0 && C > 0 && top_k > 0 && H * W <= 16384, "centernet_decode: H=%d W=%d (H*W <= 16384) C=%d top_k=%d", H, W, C, top_k);
    ODTK_REQUIRE(score_threshold >= 0.f, "centernet_decode: score_threshold must be >= 0");
    hipStream_t st = (hipStream_t)stream;
    float* score = (float*)workspace;
    int* cls = (int*)(score + (size_t)H * W);
    hipLaunchKernelGGL(centernet_score_kernel, dim3(ceil_div(H * W, DH_THREADS)), dim3(DH_THREADS), 0, st, keypoints, H * W, C, score, cls);
    int n2 = 1;
    while (n2 < H * W) n2 <<= 1;
    ODTK_CHECK_HIP(hipFuncSetAttribute((const void*)centernet_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    hipLaunchKernelGGL(centernet_topk_kernel, dim3(1), dim3(1024), (size_t)n2 * 8, st, score, cls, offset, size, H, W, stride,
                       score_threshold, top_k, scores, bbox, class_id, count);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------- FCOS
static int fcos_fill(FcArgs& a, const int* shapes, int N, int C, int P) {
    int off = 0;
    for (int l = 0; l < FC_LEVELS; ++l) {
        a.H[l] = shapes[2 * l]; a.W[l] = shapes[2 * l + 1];
        ODTK_REQUIRE(a.H[l] > 0 && a.W[l] > 0, "fcos: bad level %d shape", l);
        a.stride[l] = (float)(8 << l);
        a.nblk[l] = ceil_div(a.H[l] * a.W[l], DH_THREADS);
        a.boff[l] = off;
        off += a.nblk[l];
    }
    a.boff[FC_LEVELS] = off;
    a.N = N; a.C = C; a.P = P;
    return ODTK_OK;
}

extern "C" long long odtk_fcos_workspace_bytes(const int* shapes, int N) {
    long long blocks = 0;
    for (int l = 0; l < FC_LEVELS; ++l) blocks += ((long long)shapes[2 * l] * shapes[2 * l + 1] + DH_THREADS - 1) / DH_THREADS;
    return (long long)N * blocks * 4 * 4 + (long long)N * FC_LEVELS * 4 * 4;
}

extern "C" int odtk_fcos_loss(const float* const* conf, const float* const* reg, const float* const* center, const int* shapes,
                              const float* gt, int N, int C, int P, float grad_scale, float* loss, float* const* d_conf,
                              float* const* d_reg, float* const* d_center, void* workspace, void* stream) {
    ODTK_REQUIRE(conf && reg && center && shapes && gt && loss && d_conf && d_reg && d_center && workspace, "fcos_loss: null pointer");
    ODTK_REQUIRE(N > 0 && C > 0 && C <= DH_MAXC && P > 0 && P <= DH_MAX_GT, "fcos_loss: N=%d C=%d P=%d out of range", N, C, P);
    FcArgs a;
    memset(&a, 0, sizeof(a));
    if (int e = fcos_fill(a, shapes, N, C, P)) return e;
    for (int l = 0; l < FC_LEVELS; ++l) {
        ODTK_REQUIRE(conf[l] && reg[l] && center[l] && d_conf[l] && d_reg[l] && d_center[l], "fcos_loss: null level pointer %d", l);
        a.conf[l] = conf[l]; a.reg[l] = reg[l]; a.center[l] = center[l];
        a.d_conf[l] = d_conf[l]; a.d_reg[l] = d_reg[l]; a.d_center[l] = d_center[l];
    }
    a.gt = gt; a.grad_scale = grad_scale; a.loss = loss;
    a.parts = (float*)workspace;
    a.lsum = a.parts + (size_t)N * a.boff[FC_LEVELS] * 4;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fcos_level_kernel<0>, dim3(a.boff[FC_LEVELS], N), dim3(DH_THREADS), 0, st, a);
    hipLaunchKernelGGL(fcos_reduce_kernel, dim3(N), dim3(64), 0, st, a);
    hipLaunchKernelGGL(fcos_level_kernel<1>, dim3(a.boff[FC_LEVELS], N), dim3(DH_THREADS), 0, st, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_fcos_decode_candidates(const float* const* conf, const float* const* reg, const float* const* center,
                                           const int* shapes, int C, float* pconf, float* pbbox, void* stream) {
    ODTK_REQUIRE(conf && reg && center && shapes && pconf && pbbox, "fcos_decode_candidates: null pointer");
    ODTK_REQUIRE(C > 0, "fcos_decode_candidates: C=%d", C);
    FcDecArgs a;
    int off = 0;
    for (int l = 0; l < FC_LEVELS; ++l) {
        ODTK_REQUIRE(conf[l] && reg[l] && center[l] && shapes[2 * l] > 0 && shapes[2 * l + 1] > 0, "fcos_decode_candidates: bad level %d", l);
        a.conf[l] = conf[l]; a.reg[l] = reg[l]; a.center[l] = center[l];
        a.H[l] = shapes[2 * l]; a.W[l] = shapes[2 * l + 1]; a.stride[l] = (float)(8 << l);
        a.off[l] = off;
        off += a.H[l] * a.W[l];
    }
    a.off[FC_LEVELS] = off; a.C = C; a.pconf = pconf; a.pbbox = pbbox;
    hipLaunchKernelGGL(fcos_candidates_kernel, dim3(ceil_div(off, DH_THREADS)), dim3(DH_THREADS), 0, (hipStream_t)stream, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

// ---------------------------------------------------------------------------------------------------- YOLOv3
static int yolo_fill(YlArgs& a, const float* const* pred, const int* shapes, const float* priors, const float* head_stride,
                     const float* decode_scale, int N, int P, int C) {
    ODTK_REQUIRE(pred && shapes && priors && head_stride, "yolov3: null pointer");
    ODTK_REQUIRE(P > 0 && P <= YL_MAXP && C > 0 && N > 0, "yolov3: P=%d C=%d N=%d out of range", P, C, N);
    int off = 0;
    for (int l = 0; l < YL_HEADS; ++l) {
        ODTK_REQUIRE(pred[l] && shapes[2 * l] > 0 && shapes[2 * l + 1] > 0, "yolov3: bad head %d", l);
        a.pred[l] = pred[l];
        a.H[l] = shapes[2 * l]; a.W[l] = shapes[2 * l + 1];
        a.nblk[l] = ceil_div(a.H[l] * a.W[l] * P, DH_THREADS);
        a.boff[l] = off;
        off += a.nblk[l];
        a.gstride[l] = head_stride[l];
        a.dscale[l] = decode_scale ? decode_scale[l] : 0.f;
        for (int k = 0; k < P; ++k) { a.prior[l][k][0] = priors[(l * P + k) * 2]; a.prior[l][k][1] = priors[(l * P + k) * 2 + 1]; }
    }
    a.boff[YL_HEADS] = off;
    a.N = N; a.P = P; a.C = C;
    return ODTK_OK;
}

extern "C" long long odtk_yolov3_workspace_bytes(const int* shapes, int num_priors, int N) {
    long long blocks = 0;
    for (int l = 0; l < YL_HEADS; ++l) blocks += ((long long)shapes[2 * l] * shapes[2 * l + 1] * num_priors + DH_THREADS - 1) / DH_THREADS;
    return (long long)N * blocks * 4;
}

extern "C" int odtk_yolov3_loss(const float* const* pred, const int* shapes, const float* priors, const float* head_stride,
                                const float* gt, int N, int num_priors, int C, int pad, float coord_scale, float noobj_scale,
                                float obj_scale, float class_scale, float grad_scale, float* loss_parts, float* const* d_pred,
                                void* workspace, void* stream) {
    YlArgs a;
    memset(&a, 0, sizeof(a));
    if (int e = yolo_fill(a, pred, shapes, priors, head_stride, nullptr, N, num_priors, C)) return e;
    ODTK_REQUIRE(gt && loss_parts && d_pred && workspace && pad > 0 && pad <= DH_MAX_GT, "yolov3_loss: bad argument (pad=%d)", pad);
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < YL_HEADS; ++l) {
        ODTK_REQUIRE(d_pred[l], "yolov3_loss: null gradient pointer %d", l);
        a.d_pred[l] = d_pred[l];
        if (int e = zero_async(d_pred[l], (size_t)N * a.H[l] * a.W[l] * num_priors * (C + 5) * sizeof(float), st)) return e;
    }
    a.gt = gt; a.pad = pad; a.coord_scale = coord_scale; a.noobj_scale = noobj_scale; a.obj_scale = obj_scale;
    a.class_scale = class_scale; a.grad_scale = grad_scale; a.parts = (float*)workspace; a.loss_parts = loss_parts;
    hipLaunchKernelGGL(yolov3_noobj_kernel, dim3(a.boff[YL_HEADS], N), dim3(DH_THREADS), 0, st, a);
    hipLaunchKernelGGL(yolov3_pos_kernel, dim3(N), dim3(64), 0, st, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_yolov3_decode_candidates(const float* const* pred, const int* shapes, const float* priors,
                                             const float* decode_scale, int num_priors, int C, float* confidence, float* bbox,
                                             void* stream) {
    YlArgs a;
    memset(&a, 0, sizeof(a));
    ODTK_REQUIRE(decode_scale && confidence && bbox, "yolov3_decode_candidates: null pointer");
    const float ones[YL_HEADS] = {1.f, 1.f, 1.f};
    if (int e = yolo_fill(a, pred, shapes, priors, ones, decode_scale, 1, num_priors, C)) return e;
    int total = 0;
    for (int l = 0; l < YL_HEADS; ++l) total += a.H[l] * a.W[l] * num_priors;
    hipLaunchKernelGGL(yolov3_candidates_kernel, dim3(ceil_div(total, DH_THREADS)), dim3(DH_THREADS), 0, (hipStream_t)stream, a, confidence, bbox);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}


// =======================================================================================================
// YOLOv2: YOLOv2.py:102-166 (per-image loss), :177-186 (decode).  One head [N][H][W][P][C + 5] = class (C), yx (2), hw (2),
// objectness (1) logits; ground truth and priors in CELL units (pixels / 32).  The reference's arithmetic is followed literally, quirks included:
// intersections are NOT clamped at 0 (prod of two negative extents is positive); the no-object IoU is taken on a mangled prior box
// ("yx" := y1x1, "hw" := y2x2 of the prior, then y1x1 := "yx" - "hw"/2, y2x2 := "yx" + "hw"/2); every prior of a cell that holds a box centre is
// exempt from the no-object term; two boxes in one cell both write (gradients add).  One workgroup per image; the box loop is serial with one thread
// per output channel, so duplicate (cell, prior) targets accumulate in a fixed order (deterministic).
// =======================================================================================================
namespace odtk {
namespace {

constexpr int Y2_MAX_CELLS = 64 * 64, Y2_MAX_PRIORS = 16;

struct Y2Args {
    const float *pred, *gt;
    float pri[Y2_MAX_PRIORS * 2];
    int N, H, W, P, C, pad;
    float coord, noobj, obj, cls, grad_scale, stride;
    float* loss_parts;   // [N][5] coord, class, obj, no-object sums, per-image total
    float* d_pred;
};

__global__ void __launch_bounds__(DH_THREADS) yolov2_loss_kernel(const Y2Args a) {
    __shared__ unsigned char s_has[Y2_MAX_CELLS];
    __shared__ float s_g[DH_MAX_GT][9];               // gy gx gh gw (cells), y1 x1 y2 x2, label
    __shared__ float red[DH_THREADS / 64];
    __shared__ int s_G;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int E = a.C + 5, cells = a.H * a.W;
    const long long img = (long long)cells * a.P * E;
    const float* pr = a.pred + (long long)n * img;
    float* dp = a.d_pred + (long long)n * img;
    const float* gt = a.gt + (long long)n * a.pad * 5;
    if (tid == 0) s_G = min(first_argmin_col0(gt, a.pad), DH_MAX_GT);
    for (long long i = tid; i < img; i += DH_THREADS) dp[i] = 0.f;
    for (int i = tid; i < cells; i += DH_THREADS) s_has[i] = 0;
    __syncthreads();
    const int G = s_G;
    for (int g = tid; g < G; g += DH_THREADS) {
        const float gy = gt[g * 5] / a.stride, gx = gt[g * 5 + 1] / a.stride, gh = gt[g * 5 + 2] / a.stride, gw = gt[g * 5 + 3] / a.stride;
        s_g[g][0] = gy; s_g[g][1] = gx; s_g[g][2] = gh; s_g[g][3] = gw;
        s_g[g][4] = gy - gh / 2.f; s_g[g][5] = gx - gw / 2.f; s_g[g][6] = gy + gh / 2.f; s_g[g][7] = gx + gw / 2.f;
        s_g[g][8] = gt[g * 5 + 4];
        const int cy = (int)floorf(gy), cx = (int)floorf(gx);
        if (cy >= 0 && cy < a.H && cx >= 0 && cx < a.W) s_has[cy * a.W + cx] = 1;
    }
    __syncthreads();
    // ---- no-object term: every prior of the cells without a box centre
    float l_noobj = 0.f;
    for (int idx = tid; idx < cells * a.P; idx += DH_THREADS) {
        const int cell = idx / a.P, k = idx - cell * a.P;
        if (s_has[cell]) continue;
        const int cy = cell / a.W, cx = cell - cy * a.W;
        const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f, ph = a.pri[2 * k], pw = a.pri[2 * k + 1];
        const float my = ay - ph / 2.f, mx = ax - pw / 2.f, mh = ay + ph / 2.f, mw = ax + pw / 2.f;     // "yx" := y1x1, "hw" := y2x2
        const float y1 = my - mh / 2.f, x1 = mx - mw / 2.f, y2 = my + mh / 2.f, x2 = mx + mw / 2.f;
        const float aarea = (y2 - y1) * (x2 - x1);
        float best = -INFINITY;
        for (int g = 0; g < G; ++g) {
            const float inter = (fminf(s_g[g][6], y2) - fmaxf(s_g[g][4], y1)) * (fminf(s_g[g][7], x2) - fmaxf(s_g[g][5], x1));
            const float garea = (s_g[g][6] - s_g[g][4]) * (s_g[g][7] - s_g[g][5]);
            const float iou = inter / (aarea + garea - inter);
            best = fmaxf(best, iou);
        }
        if (best <= 0.6f) {
            const float x = pr[(long long)idx * E + a.C + 4];
            l_noobj += bce_logits(x, 0.f);
            dp[(long long)idx * E + a.C + 4] = a.noobj * sigmoidf_(x) * a.grad_scale;
        }
    }
    __syncthreads();
    // ---- the boxes, one after the other; thread t owns output channel t of the chosen (cell, prior)
    float l_coord = 0.f, l_cls = 0.f, l_obj = 0.f;
    for (int g = 0; g < G; ++g) {
        const float gy = s_g[g][0], gx = s_g[g][1], gh = s_g[g][2], gw = s_g[g][3];
        const int cy = (int)floorf(gy), cx = (int)floorf(gx);
        if (cy < 0 || cy >= a.H || cx < 0 || cx >= a.W) continue;
        const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f;
        const float garea = (s_g[g][6] - s_g[g][4]) * (s_g[g][7] - s_g[g][5]);
        int bk = 0;
        float bv = -INFINITY;
        for (int k = 0; k < a.P; ++k) {
            const float ph = a.pri[2 * k], pw = a.pri[2 * k + 1];
            const float y1 = ay - ph / 2.f, x1 = ax - pw / 2.f, y2 = ay + ph / 2.f, x2 = ax + pw / 2.f;
            const float inter = (fminf(s_g[g][6], y2) - fmaxf(s_g[g][4], y1)) * (fminf(s_g[g][7], x2) - fmaxf(s_g[g][5], x1));
            const float iou = inter / (ph * pw + garea - inter);
            if (iou > bv) { bv = iou; bk = k; }        // first maximum
        }
        if (tid < E) {
            const long long o = ((long long)(cy * a.W + cx) * a.P + bk) * E + tid;
            const float x = pr[o];
            float grad;
            if (tid < a.C) {
                const float z = (tid == (int)s_g[g][8]) ? 1.f : 0.f;
                l_cls += bce_logits(x, z);
                grad = a.cls * (sigmoidf_(x) - z);
            } else if (tid < a.C + 2) {
                const float v = tid == a.C ? gy : gx;
                const float z = v - floorf(v);
                l_coord += bce_logits(x, z);
                grad = a.coord * (sigmoidf_(x) - z);
            } else if (tid < a.C + 4) {
                const float z = tid == a.C + 2 ? logf(gh / a.pri[2 * bk]) : logf(gw / a.pri[2 * bk + 1]);
                const float d = x - z;
                l_coord += 0.5f * (d * d);
                grad = a.coord * d;
            } else {
                l_obj += bce_logits(x, 1.f);
                grad = a.obj * (sigmoidf_(x) - 1.f);
            }
            dp[o] += grad * a.grad_scale;
        }
    }
    const float t_coord = block_sum<DH_THREADS>(l_coord, red);
    const float t_cls = block_sum<DH_THREADS>(l_cls, red);
    const float t_obj = block_sum<DH_THREADS>(l_obj, red);
    const float t_noobj = block_sum<DH_THREADS>(l_noobj, red);
    if (tid == 0) {
        float* lp = a.loss_parts + n * 5;
        lp[0] = t_coord; lp[1] = t_cls; lp[2] = t_obj; lp[3] = t_noobj;
        lp[4] = a.coord * t_coord + a.cls * t_cls + a.obj * t_obj + a.noobj * t_noobj;
    }
}

__global__ void __launch_bounds__(DH_THREADS) yolov2_decode_kernel(const Y2Args a, float* __restrict__ conf, float* __restrict__ bbox) {
    const int idx = blockIdx.x * DH_THREADS + threadIdx.x;
    if (idx >= a.H * a.W * a.P) return;
    const int E = a.C + 5;
    const int cell = idx / a.P, k = idx - cell * a.P;
    const int cy = cell / a.W, cx = cell - cy * a.W;
    const float* pr = a.pred + (long long)idx * E;
    const float so = sigmoidf_(pr[a.C + 4]);
    for (int c = 0; c < a.C; ++c) conf[(long long)idx * a.C + c] = sigmoidf_(pr[c]) * so;
    const float y = ((float)cy + 0.5f) + sigmoidf_(pr[a.C]), x = ((float)cx + 0.5f) + sigmoidf_(pr[a.C + 1]);
    const float h = a.pri[2 * k] + expf(pr[a.C + 2]), w = a.pri[2 * k + 1] + expf(pr[a.C + 3]);       // sums, as written (YOLOv2.py:183-184)
    float* b = bbox + (long long)idx * 4;
    b[0] = (y - h / 2.f) * a.stride; b[1] = (x - w / 2.f) * a.stride; b[2] = (y + h / 2.f) * a.stride; b[3] = (x + w / 2.f) * a.stride;
}

int y2_fill(Y2Args& a, const float* pred, int H, int W, int P, int C, const float* priors) {
    ODTK_REQUIRE(pred && priors, "yolov2: null pointer");
    ODTK_REQUIRE(H > 0 && W > 0 && H * W <= Y2_MAX_CELLS && P > 0 && P <= Y2_MAX_PRIORS && C > 0 && C + 5 <= DH_THREADS,
                 "yolov2: unsupported geometry H=%d W=%d priors=%d classes=%d", H, W, P, C);
    a.pred = pred; a.H = H; a.W = W; a.P = P; a.C = C;
    for (int i = 0; i < 2 * P; ++i) a.pri[i] = priors[i];
    return ODTK_OK;
}

}  // namespace
}  // namespace odtk

extern "C" int odtk_yolov2_loss(const float* pred, int N, int H, int W, int num_priors, int C, const float* priors, float stride, const float* gt,
                                int pad, float coord_scale, float noobj_scale, float obj_scale, float class_scale, float grad_scale,
                                float* loss_parts, float* d_pred, void* stream) {
    Y2Args a;
    memset(&a, 0, sizeof(a));
    if (int e = y2_fill(a, pred, H, W, num_priors, C, priors)) return e;
    ODTK_REQUIRE(gt && loss_parts && d_pred && N > 0 && pad > 0, "yolov2_loss: bad argument");
    a.gt = gt; a.N = N; a.pad = pad; a.stride = stride;
    a.coord = coord_scale; a.noobj = noobj_scale; a.obj = obj_scale; a.cls = class_scale; a.grad_scale = grad_scale;
    a.loss_parts = loss_parts; a.d_pred = d_pred;
    hipLaunchKernelGGL(yolov2_loss_kernel, dim3(N), dim3(DH_THREADS), 0, (hipStream_t)stream, a);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_yolov2_decode_candidates(const float* pred, int H, int W, int num_priors, int C, const float* priors, float stride,
                                             float* confidence, float* bbox, void* stream) {
    Y2Args a;
    memset(&a, 0, sizeof(a));
    if (int e = y2_fill(a, pred, H, W, num_priors, C, priors)) return e;
    ODTK_REQUIRE(confidence && bbox, "yolov2_decode_candidates: null pointer");
    a.stride = stride;
    hipLaunchKernelGGL(yolov2_decode_kernel, dim3(ceil_div(H * W * num_priors, DH_THREADS)), dim3(DH_THREADS), 0, (hipStream_t)stream, a, confidence, bbox);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
