// RetinaNet box side (gfx950): anchors, IoU matching with the 0.4 / 0.5 ignore band, softmax focal loss +
// smooth-L1 with their gradients.  SURVEY.md 8(f).1 / kernel K16.
//
// Reference: RetinaNet.py:328-355 (_get_abbox), :357-452 (_compute_one_image_loss), :457-474 (_focal_loss),
// :194-213 (batch loop).  Unlike SSD300 (8 828 priors, one workgroup per image) RetinaNet has 47 961 anchors at
// 500x500 and 120 087 at 800x800, so matching is tiled over anchors: every (anchor tile, image) workgroup
// emits per-anchor max / arg-max over the ground truth and a per-GT partial arg-max over its tile; a second
// tiny kernel reduces the partials in tile order (= first arg-max) and fixes up the best-anchor set.
// Everything that produces an INDEX follows the float32 operation order of the graph (this file is compiled
// with -ffp-contract=off and uses IEEE division) so indices are bit-identical to the CPU oracle.
#include "common.h"
#include <math.h>
namespace odtk { namespace cv { int misc_scratch(size_t bytes, hipStream_t st, char** out); } }      // conv_v3.hip: per-(device, slot) arena

namespace odtk {
namespace {

constexpr int RL_MAX_LEVELS = 8, RL_MAX_NA = 16, RL_MAX_GT = 128, RL_THREADS = 256, RL_MAXC = 32;
constexpr int RL_LOSS_TILES = 8;                   // anchor tiles per workgroup of the loss pass (2 atomics per workgroup)
constexpr int RL_TILES_PER_BLOCK = 4;              // anchor tiles per workgroup of the IoU pass (amortises the GT set-up)

struct AnchorArgs {
    int nlevels, total;
    float input_dim;
    int fh[RL_MAX_LEVELS], fw[RL_MAX_LEVELS], na[RL_MAX_LEVELS], off[RL_MAX_LEVELS + 1];
    float hw[RL_MAX_LEVELS][RL_MAX_NA][2];
};

// RetinaNet.py:328-355.  rate = input_dim / fh (float division first), centre = (i + 0.5) * rate for BOTH axes.
__global__ void retina_anchors_kernel(const AnchorArgs p, float* __restrict__ y1x1, float* __restrict__ y2x2,
                                      float* __restrict__ yx, float* __restrict__ hw) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.total) return;
    int l = 0;
    while (l + 1 < p.nlevels && a >= p.off[l + 1]) ++l;
    const int local = a - p.off[l];
    const int na = p.na[l], fw = p.fw[l];
    const int cell = local / na, an = local - cell * na;
    const int iy = cell / fw, ix = cell - iy * fw;
    const float rate = p.input_dim / (float)p.fh[l];
    const float cy = ((float)iy + 0.5f) * rate;
    const float cx = ((float)ix + 0.5f) * rate;
    const float ph = p.hw[l][an][0], pw = p.hw[l][an][1];
    const float y1 = cy - ph / 2.f, x1 = cx - pw / 2.f;
    const float y2 = cy + ph / 2.f, x2 = cx + pw / 2.f;
    y1x1[2 * a] = y1; y1x1[2 * a + 1] = x1;
    y2x2[2 * a] = y2; y2x2[2 * a + 1] = x2;
    yx[2 * a] = y1 / 2.f + y2 / 2.f; yx[2 * a + 1] = x1 / 2.f + x2 / 2.f;      // :353
    hw[2 * a] = y2 - y1; hw[2 * a + 1] = x2 - x1;                              // :354
}

struct GtBox { float y1, x1, y2, x2, area; };

__device__ __forceinline__ float iou_ga(const GtBox& g, float ay1, float ax1, float ay2, float ax2, float aarea) {
    const float iy1 = fmaxf(ay1, g.y1), ix1 = fmaxf(ax1, g.x1);
    const float iy2 = fminf(ay2, g.y2), ix2 = fminf(ax2, g.x2);
    const float ih = fmaxf(iy2 - iy1, 0.f), iw = fmaxf(ix2 - ix1, 0.f);
    const float inter = ih * iw;
    return inter / (aarea + g.area - inter);                                   // :382
}

// number of valid GT rows = first index of the minimum of column 0 (tf.argmin, :358)
__device__ int gt_count(const float* gt, int P, float* s_val, int* s_idx) {
    const int tid = threadIdx.x;
    float v = INFINITY; int idx = 0x7fffffff;
    for (int i = tid; i < P; i += RL_THREADS) {
        const float x = gt[i * 5];
        if (x < v) { v = x; idx = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(v, o); const int i2 = __shfl_xor(idx, o);
        if (v2 < v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = idx; }
    __syncthreads();
    float bv = s_val[0]; int bi = s_idx[0];
    for (int w = 1; w < RL_THREADS / 64; ++w)
        if (s_val[w] < bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
    __syncthreads();
    return bi;
}

// Pass 1: per anchor max / first arg-max over the GT; per (tile, GT) first arg-max over the tile's anchors.
__global__ void __launch_bounds__(RL_THREADS) retina_iou_kernel(
    const float* __restrict__ y1x1, const float* __restrict__ y2x2, const float* __restrict__ hw, int A,
    const float* __restrict__ gt, int P, float* __restrict__ maxiou, int* __restrict__ rgindex,
    float* __restrict__ part_iou, int* __restrict__ part_idx, unsigned char* __restrict__ status,
    int* __restrict__ counts) {
    __shared__ GtBox s_g[RL_MAX_GT];
    __shared__ float s_val[RL_THREADS / 64];
    __shared__ int s_idx[RL_THREADS / 64];
    __shared__ float s_wv[RL_THREADS / 64][RL_MAX_GT];
    __shared__ int s_wi[RL_THREADS / 64][RL_MAX_GT];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = (A + RL_THREADS - 1) / RL_THREADS;
    const float* g = gt + (size_t)n * P * 5;
    const int G = gt_count(g, P, s_val, s_idx);
    for (int i = tid; i < G; i += RL_THREADS) {
        const float yc = g[i * 5], xc = g[i * 5 + 1], h = g[i * 5 + 2], w = g[i * 5 + 3];
        GtBox b;
        b.y1 = yc - h / 2.f; b.x1 = xc - w / 2.f; b.y2 = yc + h / 2.f; b.x2 = xc + w / 2.f;   // :361-362
        b.area = h * w;
        s_g[i] = b;
    }
    __syncthreads();
    int blk_pos = 0, blk_neg = 0;
    for (int tile = blockIdx.x * RL_TILES_PER_BLOCK; tile < ntile && tile < (int)(blockIdx.x + 1) * RL_TILES_PER_BLOCK; ++tile) {
    const int a = tile * RL_THREADS + tid;
    const bool in = a < A;
    float ay1 = 0.f, ax1 = 0.f, ay2 = 0.f, ax2 = 0.f, aarea = 1.f;
    if (in) {
        ay1 = y1x1[2 * a]; ax1 = y1x1[2 * a + 1]; ay2 = y2x2[2 * a]; ax2 = y2x2[2 * a + 1];
        aarea = hw[2 * a] * hw[2 * a + 1];
    }
    float m = -1.f; int r = 0;
    for (int gi = 0; gi < G; ++gi) {
        const float v = in ? iou_ga(s_g[gi], ay1, ax1, ay2, ax2, aarea) : -1.f;
        if (v > m) { m = v; r = gi; }                       // first max over the GT (:410, :414)
        // first arg-max over this tile's anchors for GT gi (lowest anchor index among equal maxima)
        float bv = v; int bi = in ? a : 0x7fffffff;
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o); const int i2 = __shfl_xor(bi, o);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { s_wv[wave][gi] = bv; s_wi[wave][gi] = bi; }
    }
    // provisional status (best anchors are fixed up per image afterwards): 0 ignore, 1 positive, 2 negative (:416-417)
    const unsigned char st = !in ? 0 : (m > 0.5f ? 1 : (m < 0.4f ? 2 : 0));
    if (in) { maxiou[(size_t)n * A + a] = m; rgindex[(size_t)n * A + a] = r; status[(size_t)n * A + a] = st; }
    {
        const unsigned long long bp = __ballot(st == 1), bn = __ballot(st == 2);
        if (lane == 0) { blk_pos += __popcll(bp); blk_neg += __popcll(bn); }      // per-wave running counts
    }
    __syncthreads();
    for (int gi = tid; gi < G; gi += RL_THREADS) {
        float bv = s_wv[0][gi]; int bi = s_wi[0][gi];
        for (int w = 1; w < RL_THREADS / 64; ++w)
            if (s_wv[w][gi] > bv || (s_wv[w][gi] == bv && s_wi[w][gi] < bi)) { bv = s_wv[w][gi]; bi = s_wi[w][gi]; }
        part_iou[((size_t)n * ntile + tile) * P + gi] = bv;
        part_idx[((size_t)n * ntile + tile) * P + gi] = bi;
    }
    __syncthreads();
    }
    // one atomic per wave at the very end (same-address atomics serialise in L2: keep them few)
    if (lane == 0) {
        if (blk_pos) atomicAdd(counts + n * 4 + 0, blk_pos);
        if (blk_neg) atomicAdd(counts + n * 4 + 1, blk_neg);
    }
}

// Pass 2 (one workgroup per image): best anchor per GT = first arg-max over all tiles (tiles in ascending order,
// strict >), then the best anchors leave the "other" set (:397-407): status 3, counts corrected once per DISTINCT
// best anchor.  counts[n] = {rows of the positive set = G + #(IoU > 0.5 among the others), #negatives, min(3 * positives, negatives), 0}.
__global__ void __launch_bounds__(RL_THREADS) retina_status_kernel(
    const float* __restrict__ gt, int P, int A, int ntile, const float* __restrict__ part_iou,
    const int* __restrict__ part_idx, const float* __restrict__ maxiou, int* __restrict__ ngt, int* __restrict__ best,
    unsigned char* __restrict__ status, int* __restrict__ counts) {
    __shared__ float s_val[RL_THREADS / 64];
    __shared__ int s_idx[RL_THREADS / 64];
    __shared__ float s_pv[RL_THREADS / 64];
    __shared__ int s_pi[RL_THREADS / 64];
    __shared__ int s_best[RL_MAX_GT];
    __shared__ int s_fix[2];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gt_count(gt + (size_t)n * P * 5, P, s_val, s_idx);
    if (tid < 2) s_fix[tid] = 0;
    for (int gi = 0; gi < G; ++gi) {
        // first arg-max over the tiles: (max value, lowest tile) == lowest anchor index among equal maxima
        float bv = -2.f; int bt = 0x7fffffff;
        for (int t = tid; t < ntile; t += RL_THREADS) {
            const float v = part_iou[((size_t)n * ntile + t) * P + gi];
            if (v > bv) { bv = v; bt = t; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o); const int t2 = __shfl_xor(bt, o);
            if (v2 > bv || (v2 == bv && t2 < bt)) { bv = v2; bt = t2; }
        }
        if (lane == 0) { s_pv[wave] = bv; s_pi[wave] = bt; }
        __syncthreads();
        if (tid == 0) {
            float fv = s_pv[0]; int ft = s_pi[0];
            for (int w = 1; w < RL_THREADS / 64; ++w)
                if (s_pv[w] > fv || (s_pv[w] == fv && s_pi[w] < ft)) { fv = s_pv[w]; ft = s_pi[w]; }
            const int b = part_idx[((size_t)n * ntile + ft) * P + gi];
            s_best[gi] = b;
            best[(size_t)n * P + gi] = b;
        }
        __syncthreads();
    }
    for (int gi = tid; gi < G; gi += RL_THREADS) {
        const int b = s_best[gi];
        bool first = true;
        for (int k = 0; k < gi; ++k) first = first && s_best[k] != b;
        if (first) {
            const float m = maxiou[(size_t)n * A + b];
            if (m > 0.5f) atomicAdd(&s_fix[0], 1);
            else if (m < 0.4f) atomicAdd(&s_fix[1], 1);
        }
        status[(size_t)n * A + b] = 3;
    }
    __syncthreads();
    if (tid == 0) {
        ngt[n] = G;
        const int np_ = G + counts[n * 4 + 0] - s_fix[0], nn_ = counts[n * 4 + 1] - s_fix[1];
        counts[n * 4 + 0] = np_;
        counts[n * 4 + 1] = nn_;
        counts[n * 4 + 2] = 3 * np_ < nn_ ? 3 * np_ : nn_;         // hard-negative budget of the detectors that mine (RefineDet.py:521)
        counts[n * 4 + 3] = 0;
    }
}

struct RLossArgs {
    const float* pconf; const float* pbox; int N, A, C;
    const float* yx; const float* hw; const float* gt; int P;
    const int* ngt; const int* best; const unsigned char* status; const int* rgindex; const int* counts;
    float alpha, gamma, grad_scale;
    float* loss_parts; float* dconf; float* dbox;
    float* parts;        // [N][workgroups of the loss pass][2] loss sums of the workgroups: added in workgroup order by retina_best_rows_kernel (round 6: the loss
                         // VALUE used to leave by float atomics -- the gradients never did -- and differed in the last bit from run to run)
};

// focal term of one row and its gradient w.r.t. the logits (RetinaNet.py:457-474); `add` accumulates (best rows).
__device__ __forceinline__ float focal_row(const RLossArgs& a, const float* z, float* dz, int label, float scale, bool add) {
    float e[RL_MAXC];                                   // fully unrolled below: stays in registers
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < RL_MAXC; ++c) { e[c] = c < a.C ? z[c] : -INFINITY; m = fmaxf(m, e[c]); }
    float s = 0.f, el = 0.f;
#pragma unroll
    for (int c = 0; c < RL_MAXC; ++c) {
        e[c] = c < a.C ? expf(e[c] - m) : 0.f;
        s += e[c];
        el = c == label ? e[c] : el;
    }
    const float inv = 1.f / s;
    const float pu = el * inv;
    const float pt = fminf(fmaxf(pu, 1e-8f), 1.f);
    const float om = 1.f - pt;
    const float lg = logf(pt);
    const float loss = -a.alpha * powf(om, a.gamma) * lg;
    // d loss / d pt (zero where the clip is active from below)
    float dpt = 0.f;
    if (pu >= 1e-8f) dpt = a.alpha * (a.gamma * powf(om, a.gamma - 1.f) * lg - powf(om, a.gamma) / pt);
    const float k = dpt * pu * scale;
#pragma unroll
    for (int c = 0; c < RL_MAXC; ++c) {
        if (c < a.C) {
            const float g = k * ((c == label ? 1.f : 0.f) - e[c] * inv);
            if (add) atomicAdd(dz + c, g); else dz[c] = g;
        }
    }
    return loss;
}

// smooth-L1 of one positive row vs GT g (encode :441-442, loss :443-445) and its gradient
__device__ __forceinline__ float box_row(const RLossArgs& a, int n, int anchor, int g, const float* pb, float* db, float scale, bool add) {
    const float* gb = a.gt + ((size_t)n * a.P + g) * 5;
    const float ayc = a.yx[2 * anchor], axc = a.yx[2 * anchor + 1], ah = a.hw[2 * anchor], aw = a.hw[2 * anchor + 1];
    const float t[4] = {(gb[0] - ayc) / ah, (gb[1] - axc) / aw, logf(gb[2] / ah), logf(gb[3] / aw)};
    float sum = 0.f;
    for (int k = 0; k < 4; ++k) {
        const float d = pb[k] - t[k];
        const float ad = fabsf(d);
        sum += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        const float gr = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * scale;
        if (add) atomicAdd(db + k, gr); else db[k] = gr;
    }
    return sum;
}

// One thread per anchor row; the 256 x C logits of a workgroup (and the gradients on the way out) move through
// LDS so that global accesses are contiguous 16-byte-per-lane streams instead of 64 lanes x (84-byte stride).
// Row pitch C (odd for 21 classes) keeps the per-lane LDS walks conflict-free.
__global__ void __launch_bounds__(RL_THREADS) retina_loss_kernel(const RLossArgs a) {
    __shared__ float s_red[2][RL_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float s_z[RL_THREADS * RL_MAXC];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int num_pos = a.counts[n * 4 + 0];
    const float inv_np = 1.f / (float)num_pos;
    const float gs = a.grad_scale * inv_np;
    float conf = 0.f, coord = 0.f;
    const int ntile = (a.A + RL_THREADS - 1) / RL_THREADS;
    for (int tile = blockIdx.x * RL_LOSS_TILES; tile < ntile && tile < (int)(blockIdx.x + 1) * RL_LOSS_TILES; ++tile) {
    const int a0 = tile * RL_THREADS;
    const int rows = min(RL_THREADS, a.A - a0);
    const size_t base = ((size_t)n * a.A + a0) * a.C;
    const int nflt = rows * a.C;
    // global side: 16-byte accesses from the first aligned element (the row pitch C = 21 floats makes `base` land on
    // any 4-byte phase), scalar head / tail; LDS side: scalar (the LDS image keeps the global phase)
    const int head = min(nflt, (int)((4 - (base & 3)) & 3));
    const int nvec = (nflt - head) / 4;
    const int tail0 = head + nvec * 4;
    {
        const float4* src = reinterpret_cast<const float4*>(a.pconf + base + head);
        for (int i = tid; i < nvec; i += RL_THREADS) {
            const float4 v = src[i];
            float* d = s_z + head + 4 * i;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (tid < head) s_z[tid] = a.pconf[base + tid];
        if (tid < nflt - tail0) s_z[tail0 + tid] = a.pconf[base + tail0 + tid];
    }
    __syncthreads();
    const int an = a0 + tid;
    if (tid < rows) {
        const size_t row = (size_t)n * a.A + an;
        float* z = s_z + tid * a.C;                 // logits in, gradient out (in place)
        const float* pb = a.pbox + row * 4;
        float4 dbv = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned char st = a.status[row];
        if (st == 1) {
            const int g = a.rgindex[row];
            const int label = (int)a.gt[((size_t)n * a.P + g) * 5 + 4];
            conf += focal_row(a, z, z, label, gs, false);
            float db[4];
            coord += box_row(a, n, an, g, pb, db, gs, false);
            dbv = make_float4(db[0], db[1], db[2], db[3]);
        } else if (st == 2) {
            conf += focal_row(a, z, z, a.C - 1, gs, false);
        } else {                                   // ignore band, or a best anchor (its rows are added afterwards)
            for (int c = 0; c < a.C; ++c) z[c] = 0.f;
        }
        *reinterpret_cast<float4*>(a.dbox + row * 4) = dbv;
    }
    __syncthreads();
    // stage out
    {
        float4* dst = reinterpret_cast<float4*>(a.dconf + base + head);
        for (int i = tid; i < nvec; i += RL_THREADS) {
            const float* s = s_z + head + 4 * i;
            dst[i] = make_float4(s[0], s[1], s[2], s[3]);
        }
        if (tid < head) a.dconf[base + tid] = s_z[tid];
        if (tid < nflt - tail0) a.dconf[base + tail0 + tid] = s_z[tail0 + tid];
    }
    __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { conf += __shfl_xor(conf, o); coord += __shfl_xor(coord, o); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = conf; s_red[1][tid >> 6] = coord; }
    __syncthreads();
    if (tid == 0) {
        float c0 = 0.f, c1 = 0.f;
        for (int w = 0; w < RL_THREADS / 64; ++w) { c0 += s_red[0][w]; c1 += s_red[1][w]; }
        float* pp = a.parts + ((size_t)n * gridDim.x + blockIdx.x) * 2;
        pp[0] = c0 * inv_np;
        pp[1] = c1 * inv_np;
    }
}

// the G "best" rows of every image (duplicates allowed: two GT may pick one anchor) -- accumulated on top of
// the zeros written by retina_loss_kernel, hence a second launch
__global__ void __launch_bounds__(RL_MAX_GT) retina_best_rows_kernel(const RLossArgs a, const int nparts) {
    __shared__ float s_w[2][RL_MAX_GT / 64];
    const int n = blockIdx.x, g = threadIdx.x;
    const int G = a.ngt[n];
    float conf = 0.f, coord = 0.f;
    const float inv_np = 1.f / (float)a.counts[n * 4 + 0];
    if (g < G) {
        const int an = a.best[(size_t)n * a.P + g];
        const size_t row = (size_t)n * a.A + an;
        const int label = (int)a.gt[((size_t)n * a.P + g) * 5 + 4];
        const float gs = a.grad_scale * inv_np;
        conf = focal_row(a, a.pconf + row * a.C, a.dconf + row * a.C, label, gs, true);
        coord = box_row(a, n, an, g, a.pbox + row * 4, a.dbox + row * 4, gs, true);
    }
    for (int o = 32; o > 0; o >>= 1) { conf += __shfl_xor(conf, o); coord += __shfl_xor(coord, o); }
    if ((g & 63) == 0) { s_w[0][g >> 6] = conf * inv_np; s_w[1][g >> 6] = coord * inv_np; }
    __syncthreads();
    if (g < 2) {                                       // fixed order: the loss pass's workgroups, then this launch's waves
        float t = 0.f;
        const float* pp = a.parts + (size_t)n * nparts * 2 + g;
        for (int b = 0; b < nparts; ++b) t += pp[2 * b];
        for (int w = 0; w < RL_MAX_GT / 64; ++w) t += s_w[g][w];
        a.loss_parts[n * 2 + g] = t;
    }
}

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" int odtk_retina_anchors(int input_dim, int nlevels, const int* fh, const int* fw, const int* na,
                                   const float* prior_hw, float* y1x1, float* y2x2, float* yx, float* hw, void* stream) {
    ODTK_REQUIRE(fh && fw && na && prior_hw && y1x1 && y2x2 && yx && hw, "retina_anchors: null pointer");
    ODTK_REQUIRE(nlevels > 0 && nlevels <= RL_MAX_LEVELS, "retina_anchors: nlevels=%d out of range", nlevels);
    AnchorArgs p;
    p.nlevels = nlevels; p.input_dim = (float)input_dim;
    int off = 0, k = 0;
    for (int l = 0; l < nlevels; ++l) {
        ODTK_REQUIRE(na[l] > 0 && na[l] <= RL_MAX_NA && fh[l] > 0 && fw[l] > 0, "retina_anchors: bad level %d", l);
        p.fh[l] = fh[l]; p.fw[l] = fw[l]; p.na[l] = na[l]; p.off[l] = off;
        for (int i = 0; i < na[l]; ++i, ++k) { p.hw[l][i][0] = prior_hw[2 * k]; p.hw[l][i][1] = prior_hw[2 * k + 1]; }
        off += fh[l] * fw[l] * na[l];
    }
    p.off[nlevels] = off; p.total = off;
    hipLaunchKernelGGL(retina_anchors_kernel, dim3(ceil_div(off, 256)), dim3(256), 0, (hipStream_t)stream, p, y1x1, y2x2, yx, hw);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" long long odtk_retina_match_workspace_bytes(int A, int N, int P) {
    const long long ntile = (A + RL_THREADS - 1) / RL_THREADS;
    return (long long)N * A * 4 + 2ll * N * ntile * P * 4;
}

extern "C" int odtk_retina_match(const float* y1x1, const float* y2x2, const float* hw, int A, const float* gt, int N,
                                 int P, int* ngt, int* best, unsigned char* status, int* rgindex, int* counts,
                                 void* workspace, void* stream) {
    ODTK_REQUIRE(y1x1 && y2x2 && hw && gt && ngt && best && status && rgindex && counts && workspace, "retina_match: null pointer");
    ODTK_REQUIRE(A > 0 && N > 0 && P > 0 && P <= RL_MAX_GT, "retina_match: A=%d N=%d P=%d out of range", A, N, P);
    const int ntile = ceil_div(A, RL_THREADS);
    float* maxiou = (float*)workspace;
    float* part_iou = maxiou + (size_t)N * A;
    int* part_idx = (int*)(part_iou + (size_t)N * ntile * P);
    hipStream_t st = (hipStream_t)stream;
    if (int e = zero_async(counts, (size_t)N * 4 * sizeof(int), st)) return e;
    hipLaunchKernelGGL(retina_iou_kernel, dim3(ceil_div(ntile, RL_TILES_PER_BLOCK), N), dim3(RL_THREADS), 0, st, y1x1, y2x2, hw, A, gt, P, maxiou, rgindex,
                       part_iou, part_idx, status, counts);
    hipLaunchKernelGGL(retina_status_kernel, dim3(N), dim3(RL_THREADS), 0, st, gt, P, A, ntile, part_iou, part_idx, maxiou,
                       ngt, best, status, counts);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_retina_loss(const float* pconf, const float* pbox, int N, int A, int C, const float* yx,
                                const float* hw, const float* gt, int P, const int* ngt, const int* best,
                                const unsigned char* status, const int* rgindex, const int* counts, float alpha,
                                float gamma, float grad_scale, float* loss_parts, float* dconf, float* dbox, void* stream) {
    ODTK_REQUIRE(pconf && pbox && yx && hw && gt && ngt && best && status && rgindex && counts && loss_parts && dconf && dbox,
                 "retina_loss: null pointer");
    ODTK_REQUIRE(C > 1 && C <= RL_MAXC && P > 0 && P <= RL_MAX_GT, "retina_loss: C=%d P=%d unsupported", C, P);
    hipStream_t st = (hipStream_t)stream;
    const int nparts = ceil_div(ceil_div(A, RL_THREADS), RL_LOSS_TILES);
    char* parts = nullptr;
    if (int e = cv::misc_scratch((size_t)N * nparts * 2 * sizeof(float), st, &parts)) return e;
    RLossArgs a;
    a.parts = (float*)parts;
    a.pconf = pconf; a.pbox = pbox; a.N = N; a.A = A; a.C = C; a.yx = yx; a.hw = hw; a.gt = gt; a.P = P;
    a.ngt = ngt; a.best = best; a.status = status; a.rgindex = rgindex; a.counts = counts;
    a.alpha = alpha; a.gamma = gamma; a.grad_scale = grad_scale; a.loss_parts = loss_parts; a.dconf = dconf; a.dbox = dbox;
    hipLaunchKernelGGL(retina_loss_kernel, dim3(nparts, N), dim3(RL_THREADS), 0, st, a);
    hipLaunchKernelGGL(retina_best_rows_kernel, dim3(N), dim3(RL_MAX_GT), 0, st, a, nparts);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
