// Implicit-GEMM convolution on CDNA4 matrix cores (gfx950 only).
//
// Replaces tf.nn.conv2d / tf.layers.conv2d forward (reference SSD300.py:519, :524)
// and the Conv2DBackpropInput / Conv2DBackpropFilter ops added by
// optimizer.minimize (SSD300.py:154).
//
// One kernel family, two operand dtypes:
//   bf16 storage -> v_mfma_f32_32x32x16_bf16   (throughput path)
//   f32  storage -> v_mfma_f32_32x32x2_f32     (exact-f32 parity path, same indexing)
//
// GEMM view (all three passes): D[p][q] = sum_k P[p][k] * Q[q][k]
//   fwd / dgrad : p = output channel, q = output pixel, k = (r, s, c)    "gather" kernel
//   wgrad       : p = output channel, q = (r, s, c),    k = pixel        "transpose" kernel
// A workgroup (256 threads = 4 wave64, 2x2) owns a PT x 128 tile of D (PT = 64 | 128);
// each k-step stages a PT x 128 B and a 128 x 128 B slab (k-contiguous rows) in LDS.
// LDS rows are 128 B = eight 16-B slots; slot s of row r lives at slot s ^ swz(r), which
// makes every ds_read_b128 lane-group of the 32x32 fragment reads conflict-free.
// MFMA operand 1 (D rows) = channels, operand 2 (D cols) = pixels / (r,s,c): each lane then
// owns 4 consecutive output channels of one pixel per accumulator quad -> 8/16-byte stores.
#include "conv_common.h"

namespace odtk {
using namespace cv;
namespace {


// ---------------------------------------------------------------------------------------
// gather kernel: forward conv and dgrad
// ---------------------------------------------------------------------------------------

template <typename T, typename TO, int PT>
__global__ void __launch_bounds__(256) conv_gather_kernel(const GatherArgs a) {
    constexpr int KCH = Mma<T>::KCH;
    constexpr int BKE = 8 * KCH;
    constexpr int PI = PT / 64, QI = 2, PL = PT / 32;
    __shared__ __attribute__((aligned(16))) char smem[(PT + 128) * 128];
    char* sP = smem;
    char* sQ = smem + PT * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave & 1, wq = wave >> 1;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = vb / a.tiles_p, tp = vb - tq * a.tiles_p;
    const int p0 = tp * PT, q0 = tq * 128;

    const int cc = tid & 7, r0 = tid >> 3;
    int hb[4], wb[4], nb[4];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = q0 + r0 + 32 * i;
        if (m < a.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            hb[i] = ho * a.ostride - a.pad_t;
            wb[i] = wo * a.ostride - a.pad_l;
            nb[i] = n * a.H;
        } else {
            hb[i] = -(1 << 28); wb[i] = 0; nb[i] = 0;
        }
    }
    int klin = cc * KCH;
    int kc, ks, kr;
    {
        const int rs = klin / a.C;
        kc = klin - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }
    uint4 rq[4], rp[PL];

    auto load_tile = [&]() {
        const bool kv = kr < a.R;
        const int dh = kr * a.dil, dw = ks * a.dil;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hn = hb[i] + dh, wn = wb[i] + dw;
            int hi, wi;
            bool ok;
            if (a.idiv == 1) {
                hi = hn; wi = wn;
                ok = kv && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            } else {
                ok = kv && hn >= 0 && wn >= 0 && (hn % a.idiv) == 0 && (wn % a.idiv) == 0;
                hi = hn / a.idiv; wi = wn / a.idiv;
                ok = ok && hi < a.H && wi < a.W;
            }
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const size_t off = ((size_t)((nb[i] + hi) * a.W + wi) * a.ldx + kc) * sizeof(T);
                v = *reinterpret_cast<const uint4*>(a.x + off);
            }
            rq[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int row = p0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kv && row < a.K)
                v = *reinterpret_cast<const uint4*>(a.w + ((size_t)row * a.ldw + klin) * sizeof(T));
            rp[i] = v;
        }
        klin += BKE;
        kc += BKE;
        while (kc >= a.C) {
            kc -= a.C;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sQ + row * 128 + ((cc ^ swz(row)) << 4)) = rq[i];
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int row = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sP + row * 128 + ((cc ^ swz(row)) << 4)) = rp[i];
        }
    };

    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.Kdim + BKE - 1) / BKE;
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile();
        mma_slab<T, PI, QI>(sP, sQ, wp * (PT / 2), wq * 64, lane, acc);
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: lane (l31, hi) owns pixel q, channels base + 8*g + 4*hi + {0..3}
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int m = q0 + wq * 64 + j * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = p0 + wp * (PT / 2) + i * 32 + 8 * g + 4 * hi;
                if (c >= a.K) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                const bool full = (c + 3 < a.K);
                TO* yp = reinterpret_cast<TO*>(a.y) + (size_t)m * a.ldy + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!full && c + e >= a.K) continue;
                    float t = v[e];
                    if (a.bias) t += a.bias[c + e];
                    if (a.accumulate) t += elem<TO>::load(yp[e]);
                    if (a.relu) t = fmaxf(t, 0.f);
                    if (a.mask) {
                        const T mv = reinterpret_cast<const T*>(a.mask)[(size_t)m * a.ldmask + c + e];
                        if (!(elem<T>::load(mv) > 0.f)) t = 0.f;
                    }
                    v[e] = t;
                }
                if (full) {
                    if (sizeof(TO) == 2) {
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(yp) = o;
                    } else {
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.K) yp[e] = elem<TO>::store(v[e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// gather kernel, LDS-DMA version (stride-1 input sampling, i.e. every forward conv and the dgrad
// of stride-1 convs): operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write), two LDS stages, one barrier per k-slab.  The DMA writes lane-linear (wave base +
// lane*16), so the XOR swizzle is applied to the SOURCE chunk each lane fetches; padded / out of
// range chunks fetch from a 16-byte zero page instead of branching.
// ---------------------------------------------------------------------------------------

template <typename T, typename TO, int PT>
__global__ void __launch_bounds__(256) conv_gather_glds_kernel(const GatherArgs a) {
    constexpr int KCH = Mma<T>::KCH;
    constexpr int BKE = 8 * KCH;
    // wave layout: 2 x 2 waves of (PT / 2) x 64 for PT = 64 | 128; PT = 32 (round 4: Cout <= 32, the 7..28-channel layers of RetinaNet.py:27's widths, which
    // wasted half to 8/9 of a 64-row filter tile): 1 x 4 waves of 32 x 32
    constexpr int WP = PT >= 64 ? 2 : 1, WQ = 4 / WP;
    constexpr int PI = PT / (32 * WP), QI = 4 / WQ, PL = PT / 32;
    constexpr int STAGE = (PT + 128) * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wq = wave / WP;
    const int prow0 = wp * (PT / WP), qrow0 = wq * (128 / WQ);
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tq = vb / a.tiles_p, tp = vb - tq * a.tiles_p;
    const int p0 = tp * PT, q0 = tq * 128;

    const int r0 = tid >> 3;                         // rows r0 + 32*i
    const int cc = (tid & 7) ^ swz_g(r0);            // logical chunk this lane fetches (same for all i)
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // per pixel row: byte offset of tap (0,0) channel 0, and the bit mask of in-range taps
    long long qoff[4];
    unsigned qmask[4];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = q0 + r0 + 32 * i;
        qoff[i] = 0; qmask[i] = 0;
        if (m < a.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int hb = ho * a.ostride - a.pad_t, wb = wo * a.ostride - a.pad_l;
            qoff[i] = ((long long)(n * a.H + hb) * a.W + wb) * a.ldx * (long long)sizeof(T);
            unsigned mk = 0;
            for (int r = 0; r < a.R; ++r)
                for (int s2 = 0; s2 < a.S; ++s2) {
                    const int hi = hb + r * a.dil, wi = wb + s2 * a.dil;
                    if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) mk |= 1u << (r * a.S + s2);
                }
            qmask[i] = mk;
        }
    }
    long long poff[PL];
    bool pok[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int row = p0 + r0 + 32 * i;
        pok[i] = row < a.K;
        poff[i] = (long long)row * a.ldw * (long long)sizeof(T);
    }
    int klin = cc * KCH;
    int kc, ks, kr;
    {
        const int rs = klin / a.C;
        kc = klin - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }

    auto issue = [&](int stage) {
        const unsigned sP = smem_base + (unsigned)stage * STAGE + wave_u * 1024u;
        const unsigned sQ = sP + PT * 128;
        const bool kv = kr < a.R;
        const int tap = kr * a.S + ks;
        const long long toff = ((long long)(kr * a.dil) * a.W + ks * a.dil) * a.ldx * (long long)sizeof(T) +
                               (long long)kc * (long long)sizeof(T);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = kv && ((qmask[i] >> tap) & 1u);
            const char* src = ok ? a.x + qoff[i] + toff : zero;
            glds16(src, sQ + i * 4096u);
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const bool ok = kv && pok[i];
            const char* src = ok ? a.w + poff[i] + (long long)klin * (long long)sizeof(T) : zero;
            glds16(src, sP + i * 4096u);
        }
        klin += BKE;
        kc += BKE;
        while (kc >= a.C) {
            kc -= a.C;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };

    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.Kdim + BKE - 1) / BKE;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of slab kt landed
        __syncthreads();                                      // ... and everybody else's; slab kt-1 fully consumed
        if (kt + 1 < nk) issue((kt + 1) & 1);
        const char* sP = smem + (kt & 1) * STAGE;
        mma_slab<T, PI, QI, true>(sP, sP + PT * 128, prow0, qrow0, lane, acc);
    }

    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int m = q0 + qrow0 + j * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = p0 + prow0 + i * 32 + 8 * g + 4 * hi;
                if (c >= a.K) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                const bool full = (c + 3 < a.K);
                TO* yp = reinterpret_cast<TO*>(a.y) + (size_t)m * a.ldy + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!full && c + e >= a.K) continue;
                    float t = v[e];
                    if (a.bias) t += a.bias[c + e];
                    if (a.accumulate) t += elem<TO>::load(yp[e]);
                    if (a.relu) t = fmaxf(t, 0.f);
                    if (a.mask) {
                        const T mv = reinterpret_cast<const T*>(a.mask)[(size_t)m * a.ldmask + c + e];
                        if (!(elem<T>::load(mv) > 0.f)) t = 0.f;
                    }
                    v[e] = t;
                }
                if (full) {
                    if (sizeof(TO) == 2) {
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(yp) = o;
                    } else {
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.K) yp[e] = elem<TO>::store(v[e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// transpose kernel: wgrad.  k = pixel; both operands are read pixel-major from HBM (16-B
// channel chunks of 4 consecutive pixels per thread) and transposed in registers on the way
// into the k-contiguous LDS rows.
// ---------------------------------------------------------------------------------------

template <typename T, int PT>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs a) {
    constexpr int KCH = Mma<T>::KCH;          // channels per 16-B chunk
    constexpr int PKE = 8 * KCH;              // pixels per k-slab (128-B LDS row)
    constexpr int NCC = 128 / KCH;            // chunks across a 128-row operand tile
    constexpr int PI = PT / 64, QI = 2;
    __shared__ __attribute__((aligned(16))) char smem[(PT + 128) * 128];
    char* sP = smem;
    char* sQ = smem + PT * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave & 1, wq = wave >> 1;
    const int tile = blockIdx.x;
    const int tq = tile / a.tiles_p, tp = tile - tq * a.tiles_p;
    const int p0 = tp * PT, q0 = tq * 128;
    const int split = blockIdx.y;

    const int cc = tid % NCC, pg = tid / NCC;
    // P operand (dy): channel chunk cc of the PT-row tile (threads with cc*KCH >= PT idle)
    const int pch = p0 + cc * KCH;
    const bool p_active = (cc * KCH < PT) && (pch < a.lddy);
    // Q operand (x gathered): column j0 = q0 + cc*KCH -> fixed (r, s, c0)
    const int j0 = q0 + cc * KCH;
    const bool q_active = j0 < a.RSC;
    int qr = 0, qs = 0, qc = 0;
    if (q_active) {
        const int rs = j0 / a.C;
        qc = j0 - rs * a.C;
        qr = rs / a.S;
        qs = rs - qr * a.S;
    }
    const int dh = qr * a.dil - a.pad_t, dw_ = qs * a.dil - a.pad_l;

    const int it0 = split * a.iters_per_split;
    int it1 = it0 + a.iters_per_split;
    const int iters_total = (a.P + PKE - 1) / PKE;
    if (it1 > iters_total) it1 = iters_total;

    uint4 vp[4], vq[4];
    auto load_tile = [&](int it) {
        const int pb = it * PKE + 4 * pg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = pb + i;
            uint4 u = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);
            if (p < a.P) {
                if (p_active)
                    u = *reinterpret_cast<const uint4*>(a.dy + ((size_t)p * a.lddy + pch) * sizeof(T));
                if (q_active) {
                    const unsigned n = fdiv((unsigned)p, a.div_howo);
                    const unsigned rem = (unsigned)p - n * (unsigned)(a.Ho * a.Wo);
                    const unsigned ho = fdiv(rem, a.div_wo);
                    const unsigned wo = rem - ho * (unsigned)a.Wo;
                    const int hi = (int)ho * a.stride + dh, wi = (int)wo * a.stride + dw_;
                    if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
                        const size_t off = ((size_t)(((int)n * a.H + hi) * a.W + wi) * a.ldx + qc) * sizeof(T);
                        v = *reinterpret_cast<const uint4*>(a.x + off);
                    }
                }
            }
            vp[i] = u;
            vq[i] = v;
        }
    };
    // 4 pixels x KCH channels -> KCH rows of 4 pixels
    auto store_op = [&](char* s, const uint4 (&v)[4], int rows_valid) {
        if (sizeof(T) == 2) {
            const unsigned* d0 = reinterpret_cast<const unsigned*>(&v[0]);
            const unsigned* d1 = reinterpret_cast<const unsigned*>(&v[1]);
            const unsigned* d2 = reinterpret_cast<const unsigned*>(&v[2]);
            const unsigned* d3 = reinterpret_cast<const unsigned*>(&v[3]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = cc * 8 + j;
                if (row >= rows_valid) break;
                const int h = j >> 1;
                uint2 o;
                if ((j & 1) == 0) {
                    o.x = (d0[h] & 0xffffu) | (d1[h] << 16);
                    o.y = (d2[h] & 0xffffu) | (d3[h] << 16);
                } else {
                    o.x = (d0[h] >> 16) | (d1[h] & 0xffff0000u);
                    o.y = (d2[h] >> 16) | (d3[h] & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(s + row * 128 + (((pg >> 1) ^ swz(row)) << 4) + ((pg & 1) << 3)) = o;
            }
        } else {
            const unsigned* d0 = reinterpret_cast<const unsigned*>(&v[0]);
            const unsigned* d1 = reinterpret_cast<const unsigned*>(&v[1]);
            const unsigned* d2 = reinterpret_cast<const unsigned*>(&v[2]);
            const unsigned* d3 = reinterpret_cast<const unsigned*>(&v[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = cc * 4 + j;
                if (row >= rows_valid) break;
                *reinterpret_cast<uint4*>(s + row * 128 + ((pg ^ swz(row)) << 4)) =
                    make_uint4(d0[j], d1[j], d2[j], d3[j]);
            }
        }
    };

    // fused bias gradient: the q-tile-0 blocks also column-sum their dy operand
    const bool do_bias = a.dbias != nullptr && tq == 0 && p_active;
    float bsum[KCH];
#pragma unroll
    for (int e = 0; e < KCH; ++e) bsum[e] = 0.f;
    auto acc_bias = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned* d = reinterpret_cast<const unsigned*>(&vp[i]);
            if (sizeof(T) == 2) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    bsum[(2 * h) % KCH] += __uint_as_float(d[h] << 16);
                    bsum[(2 * h + 1) % KCH] += __uint_as_float(d[h] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; ++h) bsum[h % KCH] += __uint_as_float(d[h]);
            }
        }
    };

    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (it0 < it1) {
        load_tile(it0);
        if (do_bias) acc_bias();
        store_op(sP, vp, PT);
        store_op(sQ, vq, 128);
        __syncthreads();
        for (int it = it0; it < it1; ++it) {
            const bool more = it + 1 < it1;
            if (more) load_tile(it + 1);
            mma_slab<T, PI, QI>(sP, sQ, wp * (PT / 2), wq * 64, lane, acc);
            __syncthreads();
            if (more) {
                if (do_bias) acc_bias();
                store_op(sP, vp, PT);
                store_op(sQ, vq, 128);
                __syncthreads();
            }
        }
    }
    if (a.dbias != nullptr && tq == 0) {      // block-uniform: reduce over the pixel groups in LDS first
        constexpr int NPG = 256 / NCC;
        float* red = reinterpret_cast<float*>(smem);            // [NPG][PT]; operand slabs are dead now
        if (cc * KCH < PT) {
#pragma unroll
            for (int e = 0; e < KCH; ++e) red[pg * PT + cc * KCH + e] = bsum[e];
        }
        __syncthreads();
        if (tid < PT && p0 + tid < a.K) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < NPG; ++g) t += red[g * PT + tid];
            if (a.bws) a.bws[(size_t)split * a.K + p0 + tid] = t;      // deterministic mode: one bias slot per pixel split
            else atomicAdd(a.dbias + p0 + tid, t);
        }
    }

    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int col = q0 + wq * 64 + j * 32 + l31;
        if (col >= a.RSC) continue;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = p0 + wp * (PT / 2) + i * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                if (k < a.K) {
                    // deterministic mode (round 6: the f32 engines and the odd bf16 layouts that end up here): plain stores of the partial tile to ws[split],
                    // summed in split order by the reduction launch
                    if (a.ws) a.ws[((size_t)split * a.K + k) * a.RSC + col] = acc[i][j][e];
                    else atomicAdd(a.dw + (size_t)k * a.RSC + col, acc[i][j][e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// wgrad, LDS-DMA + transpose-read version (bf16, 128 x 128 tile).  Both operands stay in their
// natural [pixel][channel] layout: a k-slab is 64 pixels x 128 channels (256-B rows) per operand,
// moved HBM -> LDS as sixteen 1-KiB DMA pieces of 4 pixel rows.  The MFMA fragments (8 consecutive
// k = pixels per lane) are produced by ds_read_b64_tr_b16, which hands each lane a COLUMN of a
// 4 (pixels) x 16 (channels) block.  Inside a piece the 16-B chunk `ch` of pixel row r sits at
// slot ch ^ (4*r) so that the 4 rows x 64 B a half-wave reads cover all 64 banks exactly once
// (the swizzle is applied to the DMA source address; the destination is lane-linear).
// ---------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) conv_wgrad_dma_kernel(const WgradArgs a) {
    constexpr int PT = 128, PI = 2, QI = 2;
    constexpr int PKE = 64;                         // pixels per k-slab
    constexpr int OPB = PKE * 256;                  // bytes per operand slab
    constexpr int STAGE = 2 * OPB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave & 1, wq = wave >> 1;
    const int tile = blockIdx.x;
    const int tq = tile / a.tiles_p, tp = tile - tq * a.tiles_p;
    const int p0 = tp * PT, q0 = tq * 128;
    const int split = blockIdx.y;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // DMA lane role: pixel row r = lane>>4 of the piece, logical 16-B chunk ch = (lane&15) ^ (4*r)
    const int dr = lane >> 4;
    const int dch = (lane & 15) ^ (dr << 2);
    const int pch = p0 + dch * 8;                   // dy channel of this lane's chunk
    const bool p_col_ok = pch < a.lddy;
    const int j0 = q0 + dch * 8;                    // (r,s,c) column of this lane's chunk
    const bool q_col_ok = j0 < a.RSC;
    int qr = 0, qs = 0, qc = 0;
    if (q_col_ok) {
        const int rs = j0 / a.C;
        qc = j0 - rs * a.C;
        qr = rs / a.S;
        qs = rs - qr * a.S;
    }
    const int dh = qr * a.dil - a.pad_t, dw_ = qs * a.dil - a.pad_l;
    const int HoWo = a.Ho * a.Wo;

    const int iters_total = (a.P + PKE - 1) / PKE;
    const int it0 = split * a.iters_per_split;
    int it1 = it0 + a.iters_per_split;
    if (it1 > iters_total) it1 = iters_total;

    auto issue = [&](int it, int stage) {
        const unsigned sP = smem_base + (unsigned)stage * STAGE;
        const unsigned sQ = sP + OPB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = (int)wave_u + 4 * i;
            const int p = it * PKE + piece * 4 + dr;
            const bool pin = p < a.P;
            const char* src = (pin && p_col_ok) ? a.dy + ((size_t)p * a.lddy + pch) * 2 : zero;
            if ((a.dbg & 1) && it != it0) src = zero;
            glds16(src, sP + (unsigned)piece * 1024u);
            const char* srcq = zero;
            if (pin && q_col_ok) {
                const unsigned n = fdiv((unsigned)p, a.div_howo);
                const unsigned rem = (unsigned)p - n * (unsigned)HoWo;
                const unsigned ho = fdiv(rem, a.div_wo);
                const unsigned wo = rem - ho * (unsigned)a.Wo;
                const int hi = (int)ho * a.stride + dh, wi = (int)wo * a.stride + dw_;
                if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                    srcq = a.x + ((size_t)(((int)n * a.H + hi) * a.W + wi) * a.ldx + qc) * 2;
            }
            if ((a.dbg & 1) && it != it0) srcq = zero;
            glds16(srcq, sQ + (unsigned)piece * 1024u);
        }
    };

    // transpose-read lane role: group g = lane>>4 -> channel sub-block 16*(g&1), k half g>>1;
    // lane c = lane&15 supplies the address of (pixel row c>>2, 8-B column c&3) of that block
    const int g = lane >> 4, c = lane & 15;
    const int rr = c >> 2;                                   // pixel row inside the piece
    unsigned pfo[PI], qfo[QI];                               // byte offset inside a piece (+ k-half piece)
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int ch = (wp * 64 + i * 32 + 16 * (g & 1)) / 8 + ((c & 3) >> 1);
        pfo[i] = (unsigned)((2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int ch = (wq * 64 + j * 32 + 16 * (g & 1)) / 8 + ((c & 3) >> 1);
        qfo[j] = (unsigned)((2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8);
    }

    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool do_bias = a.dbias != nullptr && tq == 0 && wq == 0;     // wave-uniform
    float bsum[PI] = {0.f, 0.f};

    if (it0 < it1) {
        issue(it0, 0);
        for (int it = it0; it < it1; ++it) {
            const int st = (it - it0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < it1 && !((a.dbg & 4) && it > it0)) issue(it + 1, st ^ 1);
            const unsigned sP = smem_base + (unsigned)st * STAGE;
            const unsigned sQ = sP + OPB;
            // two fragment register sets: the transpose reads of sub-step ks+1 are in flight under the MFMAs of ks
            uint4 pf[2][PI], qf[2][QI];
            auto ldf = [&](int ks, uint4 (&p)[PI], uint4 (&q)[QI]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < PI; ++i) {
                    const uint2 lo = lds_tr16(sP + ks * 4096u + pfo[i]);
                    const uint2 hi2 = lds_tr16(sP + ks * 4096u + 1024u + pfo[i]);
                    p[i] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
                }
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const uint2 lo = lds_tr16(sQ + ks * 4096u + qfo[j]);
                    const uint2 hi2 = lds_tr16(sQ + ks * 4096u + 1024u + qfo[j]);
                    q[j] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
                }
            };
            ldf(0, pf[0], qf[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) ldf(ks + 1, pf[(ks + 1) & 1], qf[(ks + 1) & 1]);
#pragma unroll
                for (int i = 0; i < PI; ++i)
#pragma unroll
                    for (int j = 0; j < QI; ++j) Mma<bf16_t>::run(pf[ks & 1][i], qf[ks & 1][j], acc[i][j]);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < PI; ++i) {
                        const unsigned* d = reinterpret_cast<const unsigned*>(&pf[ks & 1][i]);
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            bsum[i] += __uint_as_float(d[h] << 16) + __uint_as_float(d[h] & 0xffff0000u);
                    }
                }
            }
        }
    }

    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int col = q0 + wq * 64 + j * 32 + l31;
        if (col >= a.RSC) continue;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = p0 + wp * 64 + i * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                if (k < a.K && !(a.dbg & 16)) {
                    if (a.ws) a.ws[((size_t)split * a.K + k) * a.RSC + col] = acc[i][j][e];      // deterministic mode: see conv_wgrad_kernel
                    else atomicAdd(a.dw + (size_t)k * a.RSC + col, acc[i][j][e]);
                }
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const float t = bsum[i] + __shfl_xor(bsum[i], 32);       // both k halves
            const int k = p0 + wp * 64 + i * 32 + l31;
            if (hi == 0 && k < a.K) {
                if (a.bws) a.bws[(size_t)split * a.K + k] = t;
                else atomicAdd(a.dbias + k, t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// filter preparation: f32 master [K][RS][C] -> cast copy and flipped/transposed dgrad copy
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void filter_prepare_kernel(const float* __restrict__ w, int K, int RS, int C, int Kp,
                                      T* __restrict__ wc, T* __restrict__ wt) {
    // grid: x over (rs, c-tiles), y over k-tiles ; 32x32 tile transpose through LDS
    __shared__ float tile[32][33];
    const int ctiles = (C + 31) / 32;
    const int rs = blockIdx.x / ctiles, ct = blockIdx.x - rs * ctiles;
    const int c0 = ct * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        float v = 0.f;
        if (k < K && c < C) {
            v = w[((size_t)k * RS + rs) * C + c];
            if (wc) wc[((size_t)k * RS + rs) * C + c] = elem<T>::store(v);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (wt) {
        const int rsf = RS - 1 - rs;   // flip both taps: (R-1-r)*S + (S-1-s) = RS-1-(r*S+s)
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, k = k0 + tx;
            if (c < C && k < Kp) wt[((size_t)c * RS + rsf) * Kp + k] = elem<T>::store(k < K ? tile[tx][i] : 0.f);
        }
    }
}

// Batched form: one launch refreshes the dgrad-layout filters of every layer (32 launches -> 1 per step).
struct FpItem {                      // mirrors odtk_fp_item (include/odtk.h)
    const float* w; void* w_t;
    int K, RS, C, Kp;
    int block_begin, ctiles, ktiles, pad_;
};
template <typename T>
__global__ void __launch_bounds__(256) filter_prepare_batched_kernel(const FpItem* __restrict__ items, int n_items) {
    // one workgroup = one tile of 64 k x 32 c for one tap: 128-byte row reads (32 f32 along c), 128-byte row writes (64 bf16 along k)
    __shared__ float tile[64][33];
    // wave-uniform binary search of the item that owns this block (block_begin ascending)
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= items[mid].block_begin) lo = mid; else hi = mid - 1;
    }
    const FpItem d = items[lo];
    const int lb = blockIdx.x - d.block_begin;
    const int kt = lb / (d.RS * d.ctiles), rem = lb - kt * (d.RS * d.ctiles);
    const int rs = rem / d.ctiles, ct = rem - rs * d.ctiles;
    const int c0 = ct * 32, k0 = kt * 64;
    {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int i = ty; i < 64; i += 8) {
            const int k = k0 + i, c = c0 + tx;
            tile[i][tx] = (k < d.K && c < d.C) ? d.w[((size_t)k * d.RS + rs) * d.C + c] : 0.f;
        }
    }
    __syncthreads();
    const int rsf = d.RS - 1 - rs;
    T* wt = reinterpret_cast<T*>(d.w_t);
    const int tk = threadIdx.x & 63, tc = threadIdx.x >> 6;
    for (int i = tc; i < 32; i += 4) {
        const int cc = c0 + i, k = k0 + tk;
        if (cc < d.C && k < d.Kp) wt[((size_t)cc * d.RS + rsf) * d.Kp + k] = elem<T>::store(k < d.K ? tile[tk][i] : 0.f);
    }
}

static bool g_force_regstage = false;   // debugging knob (odtk_debug_set key 0)
static thread_local const char* g_last_kernel = "";   // name of the conv kernel the last conv call launched (odtk_conv_last_kernel)
static int g_dbg = 0;                   // key 2: perf-experiment bits forwarded to the kernels (results are wrong when set)
static int g_dbg2 = 0;                  // key 6: dispatch A/B switches of the conv kernels that leave results intact
static int g_v3_mode = 0;               // key 1: 0 = auto, 1 = legacy 4-wave kernels only, 2 = 8-wave v3 wherever supported

template <typename T, typename TO>
int launch_gather(const GatherArgs& a, int PT, hipStream_t st) {
    const int grid = a.tiles_p * a.tiles_q;
    const bool dma = a.idiv == 1 && a.R * a.S <= 32 && !g_force_regstage;
    if (dma) {
        g_last_kernel = PT == 32 ? "conv_gather_glds_kernel<32>" : PT == 64 ? "conv_gather_glds_kernel<64>" : "conv_gather_glds_kernel<128>";
        if (PT == 32)
            hipLaunchKernelGGL((conv_gather_glds_kernel<T, TO, 32>), dim3(grid), dim3(256), 0, st, a);
        else if (PT == 64)
            hipLaunchKernelGGL((conv_gather_glds_kernel<T, TO, 64>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((conv_gather_glds_kernel<T, TO, 128>), dim3(grid), dim3(256), 0, st, a);
    } else {
        g_last_kernel = PT == 64 ? "conv_gather_kernel<64>" : "conv_gather_kernel<128>";
        if (PT == 64)
            hipLaunchKernelGGL((conv_gather_kernel<T, TO, 64>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((conv_gather_kernel<T, TO, 128>), dim3(grid), dim3(256), 0, st, a);
    }
    return 0;
}

// auto policy: the 8-wave / 3-stage kernel (one 512-thread block per CU; split-K below ~128 tiles) wins on
// every SSD300 layer (tools/conv_bench.py); the 4-wave kernels stay as the f32 / odd-layout path
bool gather_v3_auto(const GatherArgs&) { return true; }

int dispatch_gather(GatherArgs& a, int dtype, int out_dtype, hipStream_t st) {
    a.div_howo = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.div_wo = make_fastdiv((unsigned)a.Wo);
    a.dbg = g_dbg;
    a.dbg2 = g_dbg2;
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * dtype_size(dtype));
    a.w_bytes = (unsigned)((size_t)ceil_div(a.K, 1) * a.ldw * dtype_size(dtype));
    if (!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && gather_c64_supported(a, dtype, out_dtype)) {
        launch_gather_c64(a, st);
        g_last_kernel = "conv3x3_c64k64_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && gather_c8_supported(a, dtype, out_dtype)) {
        launch_gather_c8(a, st);
        g_last_kernel = "conv3x3_c8k64_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (!g_force_regstage && g_v3_mode != 1 && gather_v3_supported(a, dtype, out_dtype) &&
        (g_v3_mode >= 2 || gather_v3_auto(a) || a.pool_mode)) {              // (a fused pool that reaches this point was promised by gather_v6_pool_variant)
        a.ksplit = 1;
        if (int e = launch_gather_v3(a, st)) return e;
        g_last_kernel = a.ksplit > 1 ? (a.K <= 64 ? "conv_gather_v3_kernel<64>+splitk" : "conv_gather_v3_kernel<128>+splitk")
                        : a.ksplit == -9 ? "conv_gather_v9_kernel"
                        : a.ksplit == -6 ? "conv_gather_v6_kernel+splitk"
                        : a.ksplit < 0 ? "conv_gather_v6_kernel"
                                       : (a.K <= 64 ? "conv_gather_v3_kernel<64>" : "conv_gather_v3_kernel<128>");
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    // (PT = 32: the f32 LDS-DMA kernel only -- dbg2 bit 4 = off, A/B)
    const bool pt32 = a.K <= 32 && dtype == ODTK_F32 && a.idiv == 1 && a.R * a.S <= 32 && !g_force_regstage && !(g_dbg2 & 16);
    const int PT = pt32 ? 32 : a.K <= 64 ? 64 : 128;
    a.tiles_p = ceil_div(a.K, PT);
    a.tiles_q = ceil_div(a.M, 128);
    if (dtype == ODTK_BF16 && out_dtype == ODTK_BF16) launch_gather<bf16_t, bf16_t>(a, PT, st);
    else if (dtype == ODTK_BF16 && out_dtype == ODTK_F32) launch_gather<bf16_t, float>(a, PT, st);
    else if (dtype == ODTK_F32 && out_dtype == ODTK_F32) launch_gather<float, float>(a, PT, st);
    else {
        set_error("conv: unsupported dtype combination in=%d out=%d", dtype, out_dtype);
        return ODTK_ERR_ARG;
    }
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

int check_desc(const odtk_conv_desc* d) {
    ODTK_REQUIRE(d != nullptr, "conv: null descriptor");
    ODTK_REQUIRE(d->dtype == ODTK_BF16 || d->dtype == ODTK_F32 || d->dtype == ODTK_F32X3, "conv: bad dtype %d", d->dtype);
    ODTK_REQUIRE(d->dtype != ODTK_F32X3 || d->out_dtype == ODTK_F32X3, "conv: dtype ODTK_F32X3 goes with out_dtype ODTK_F32X3 (f32 tensors on both sides)");
    const int kch = d->dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(d->C % kch == 0 && d->ldx % kch == 0 && d->ldx >= d->C,
                 "conv: C=%d / ldx=%d must be multiples of %d", d->C, d->ldx, kch);
    ODTK_REQUIRE(d->ldy % 4 == 0 && d->ldy >= d->K, "conv: ldy=%d must be a multiple of 4 and >= K=%d", d->ldy, d->K);
    ODTK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->K > 0 && d->R > 0 && d->S > 0 &&
                 d->stride > 0 && d->dil > 0, "conv: non-positive dimension");
    ODTK_REQUIRE((long long)d->N * d->H * d->W < (1ll << 31) && (long long)d->N * d->Ho * d->Wo < (1ll << 31),
                 "conv: too many pixels");
    return ODTK_OK;
}

}  // namespace
}  // namespace odtk

using namespace odtk;

extern "C" const char* odtk_conv_last_kernel(void) { return g_last_kernel; }

// ODTK_F32X3 descriptors (defined further down, next to the argument builders they use)
static bool x3_runs(const odtk_conv_desc* d, int pass);
static odtk_conv_desc as_f32(const odtk_conv_desc* d);
static int conv2d_fwd_x3(const odtk_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu, hipStream_t st);
static int conv2d_dgrad_x3(const odtk_conv_desc* d, const float* dy, int lddy, const float* w_t, const float* relu_src, float* dx, int accumulate, hipStream_t st);

extern "C" int odtk_scratch_slot(int slot) {
    ODTK_REQUIRE(set_scratch_slot(slot) == 0, "scratch_slot: slot %d out of range (0..3)", slot);
    return ODTK_OK;
}

extern "C" int odtk_debug_set(int key, int value) {
    if (key == 0) { g_force_regstage = value != 0; return ODTK_OK; }
    if (key == 1) { g_v3_mode = value; return ODTK_OK; }
    if (key == 2) { g_dbg = value; return ODTK_OK; }
    if (key == 6) { g_dbg2 = value; cv::set_x3_zero_lo((value & (1 << 17)) != 0); return ODTK_OK; }
    if (key == 3) { set_nms_legacy(value != 0); return ODTK_OK; }
    if (key == 4) { set_bn_small_rows(value); return ODTK_OK; }
    if (key == 5) { cv::set_wgrad_deterministic(value != 0); return ODTK_OK; }
    if (key == 7) { set_gn_small_rows(value); return ODTK_OK; }
    set_error("debug_set: unknown key %d", key);
    return ODTK_ERR_ARG;
}

extern "C" int odtk_conv2d_fwd(const odtk_conv_desc* d, const void* x, const void* w, const float* bias,
                               void* y, int relu, void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    if (d->dtype == ODTK_F32X3) {
        if (x3_runs(d, 0)) return conv2d_fwd_x3(d, (const float*)x, (const float*)w, bias, (float*)y, relu, (hipStream_t)stream);
        const odtk_conv_desc f = as_f32(d);
        return odtk_conv2d_fwd(&f, x, w, bias, y, relu, stream);
    }
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.w = (const char*)w; a.bias = bias; a.mask = nullptr; a.y = (char*)y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.K = d->K; a.ldy = d->ldy; a.ldmask = 0;
    a.R = d->R; a.S = d->S; a.ostride = d->stride; a.dil = d->dil; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.idiv = 1;
    a.M = d->N * d->Ho * d->Wo; a.Kdim = d->R * d->S * d->C; a.ldw = a.Kdim;
    a.relu = relu; a.accumulate = 0;
    a.rev = 1;             // (only the persistent 64 -> 64 halo kernel looks at it: conv1_2 walks DOWN the map conv1_1 has just written upwards)
    return dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream);
}

static void fwd_args(GatherArgs& a, const odtk_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int relu) {
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.w = (const char*)w; a.bias = bias; a.mask = nullptr; a.y = (char*)y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.K = d->K; a.ldy = d->ldy; a.ldmask = 0;
    a.R = d->R; a.S = d->S; a.ostride = d->stride; a.dil = d->dil; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.idiv = 1;
    a.M = d->N * d->Ho * d->Wo; a.Kdim = d->R * d->S * d->C; a.ldw = a.Kdim;
    a.relu = relu; a.accumulate = 0;
    a.rev = 1;
}

extern "C" int odtk_conv2d_fwd_pool2x2_fused(const odtk_conv_desc* d) {
    if (check_desc(d) || d->dtype == ODTK_F32X3) return 0;
    GatherArgs a;
    fwd_args(a, d, nullptr, nullptr, nullptr, nullptr, 1);
    a.dbg = g_dbg;
    if (!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && gather_c64_supported(a, d->dtype, d->out_dtype)) return 1;
    return (!g_force_regstage && g_v3_mode != 1 && gather_v6_pool_variant(a, d->dtype, d->out_dtype)) ? 1 : 0;
}

extern "C" int odtk_maxpool2x2_fwd_idx(const void* x, void* y, void* idx, int N, int H, int W, int C, int ld, int Ho, int Wo, int dtype, void* stream);
extern "C" int odtk_maxpool_fwd(const void* x, void* y, int N, int H, int W, int C, int ld, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                                int dtype, void* stream);

extern "C" int odtk_conv2d_fwd_pool2x2(const odtk_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int relu,
                                       void* y_pool, int ld_pool, void* idx, void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(d->dtype != ODTK_F32X3, "conv2d_fwd_pool2x2: ODTK_F32X3 descriptors go to odtk_conv2d_fwd / _dgrad / _wgrad only");
    ODTK_REQUIRE(x && w && y_pool, "conv2d_fwd_pool2x2: null pointer");
    ODTK_REQUIRE(ld_pool % 8 == 0 && ld_pool >= d->K, "conv2d_fwd_pool2x2: ld_pool=%d must be a multiple of 8 and >= K=%d", ld_pool, d->K);
    GatherArgs a;
    fwd_args(a, d, x, w, bias, y, relu);
    if (relu && odtk_conv2d_fwd_pool2x2_fused(d) && idx != nullptr && ld_pool % 8 == 0) {        // (the fused pooling compares ReLU outputs as integers)
        a.ypool = (char*)y_pool; a.pidx = (unsigned short*)idx; a.ldpool = ld_pool; a.pool_mode = y ? 1 : 2;
        return dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream);
    }
    ODTK_REQUIRE(y != nullptr, "conv2d_fwd_pool2x2: this shape runs as conv + pool and needs the un-pooled output buffer y");
    ODTK_REQUIRE(ld_pool == d->ldy, "conv2d_fwd_pool2x2: the two-launch path needs ld_pool == ldy");
    if (int e = dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream)) return e;
    const int Hp = (d->Ho + 1) / 2, Wp = (d->Wo + 1) / 2;
    if (idx) return odtk_maxpool2x2_fwd_idx(y, y_pool, idx, d->N, d->Ho, d->Wo, d->K, d->ldy, Hp, Wp, d->out_dtype, stream);
    return odtk_maxpool_fwd(y, y_pool, d->N, d->Ho, d->Wo, d->K, d->ldy, Hp, Wp, 2, 2, 0, 0, d->out_dtype, stream);
}

static void dgrad_args(GatherArgs& a, const odtk_conv_desc* d, const void* dy, int lddy, const void* w_t, const void* relu_src, void* dx, int accumulate) {
    memset(&a, 0, sizeof(a));
    a.x = (const char*)dy; a.w = (const char*)w_t; a.bias = nullptr; a.mask = (const char*)relu_src; a.y = (char*)dx;
    a.N = d->N; a.H = d->Ho; a.W = d->Wo; a.C = lddy; a.ldx = lddy;
    a.Ho = d->H; a.Wo = d->W; a.K = d->C; a.ldy = d->ldx; a.ldmask = d->ldx;
    a.R = d->R; a.S = d->S; a.ostride = 1; a.dil = d->dil;
    a.pad_t = (d->R - 1) * d->dil - d->pad_t;
    a.pad_l = (d->S - 1) * d->dil - d->pad_l;
    a.idiv = d->stride;
    a.M = d->N * d->H * d->W; a.Kdim = d->R * d->S * lddy; a.ldw = a.Kdim;
    a.relu = 0; a.accumulate = accumulate;
}

extern "C" int odtk_conv2d_dgrad(const odtk_conv_desc* d, const void* dy, int lddy, const void* w_t,
                                 const void* relu_src, void* dx, int accumulate, void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(dy && w_t && dx, "conv2d_dgrad: null pointer");
    const int kch = d->dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(lddy % kch == 0 && lddy >= d->K, "conv2d_dgrad: lddy=%d must be a multiple of %d", lddy, kch);
    if (d->dtype == ODTK_F32X3) {
        if (x3_runs(d, 2)) return conv2d_dgrad_x3(d, (const float*)dy, lddy, (const float*)w_t, (const float*)relu_src, (float*)dx, accumulate, (hipStream_t)stream);
        const odtk_conv_desc f = as_f32(d);
        return odtk_conv2d_dgrad(&f, dy, lddy, w_t, relu_src, dx, accumulate, stream);
    }
    GatherArgs a;
    dgrad_args(a, d, dy, lddy, w_t, relu_src, dx, accumulate);      // the "input" of this conv is dy [N][Ho][Wo][lddy], its "output" is dx [N][H][W][C]
    return dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream);
}

// ---- ReLU mask as sign bits between a producer's forward pass and the consumer's input-gradient pass (conv1_1 -> conv1_2 of SSD300.py:193-208)
extern "C" int odtk_conv2d_relu_bits_supported(const odtk_conv_desc* producer, const odtk_conv_desc* consumer, int consumer_lddy) {
    if (check_desc(producer) || check_desc(consumer)) return 0;
    if (producer->dtype == ODTK_F32X3 || consumer->dtype == ODTK_F32X3) return 0;
    if (g_force_regstage || g_v3_mode == 1 || (g_dbg & 2048)) return 0;
    GatherArgs f, b;
    fwd_args(f, producer, nullptr, nullptr, nullptr, nullptr, 1);
    dgrad_args(b, consumer, nullptr, consumer_lddy, nullptr, nullptr, nullptr, 0);
    return (gather_c8_supported(f, producer->dtype, producer->out_dtype) && gather_c64_supported(b, consumer->dtype, consumer->out_dtype) &&
            producer->ldy == consumer->ldx && producer->N == consumer->N && producer->Ho == consumer->H && producer->Wo == consumer->W) ? 1 : 0;
}

extern "C" int odtk_conv2d_fwd_bits(const odtk_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int relu, void* relu_bits, void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(d->dtype != ODTK_F32X3, "conv2d_fwd_bits: ODTK_F32X3 descriptors go to odtk_conv2d_fwd / _dgrad / _wgrad only");
    ODTK_REQUIRE(x && w && y && relu_bits, "conv2d_fwd_bits: null pointer");
    GatherArgs a;
    fwd_args(a, d, x, w, bias, y, relu);
    ODTK_REQUIRE(!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && gather_c8_supported(a, d->dtype, d->out_dtype),
                 "conv2d_fwd_bits: only the first-layer kernel writes sign bits (odtk_conv2d_relu_bits_supported)");
    a.ybits = (unsigned char*)relu_bits;
    return dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream);
}

extern "C" int odtk_conv2d_dgrad_bits(const odtk_conv_desc* d, const void* dy, int lddy, const void* w_t, const void* relu_bits, void* dx, int accumulate,
                                      void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(d->dtype != ODTK_F32X3, "conv2d_dgrad_bits: ODTK_F32X3 descriptors go to odtk_conv2d_fwd / _dgrad / _wgrad only");
    ODTK_REQUIRE(dy && w_t && dx && relu_bits, "conv2d_dgrad_bits: null pointer");
    ODTK_REQUIRE(lddy % 8 == 0 && lddy >= d->K, "conv2d_dgrad_bits: lddy=%d must be a multiple of 8", lddy);
    GatherArgs a;
    dgrad_args(a, d, dy, lddy, w_t, nullptr, dx, accumulate);
    ODTK_REQUIRE(!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && gather_c64_supported(a, d->dtype, d->out_dtype),
                 "conv2d_dgrad_bits: only the 64 -> 64 halo kernel reads sign bits (odtk_conv2d_relu_bits_supported)");
    a.mask_bits = (const unsigned char*)relu_bits;
    return dispatch_gather(a, d->dtype, d->out_dtype, (hipStream_t)stream);
}

// ---- ODTK_F32X3: the f32 engine's convolutions on the bf16 MFMA kernels by operand splitting (conv_v3.hip; include/odtk.h) ------------------------------
static inline int pad8(int v) { return (v + 7) / 8 * 8; }
static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
// pitch (elements) of a pixel row that stores two split parts of ldc channels: 128 bytes more than the parts need, so that consecutive rows do not sit a power
// of two apart (measured: with 2 ldc exactly -- 1 KiB rows for 256 channels -- the halo kernels' patch DMA ran 5-7 % slower than on the 1.5 KiB rows of three parts)
static inline int x3_pitch(int ldc) { return 2 * ldc + 64; }
static odtk_conv_desc as_f32(const odtk_conv_desc* d) { odtk_conv_desc f = *d; f.dtype = f.out_dtype = ODTK_F32; return f; }

// Would this pass of the layer run as split bf16 products?  pass 0 = forward, 1 = filter gradient, 2 = input gradient.
// Policy (profiles/r04x_retinanet_f32_vs_f32x3_per_layer.md): below ~20 000 multiply-adds per output element row the split passes cost more than the exact
// f32 MFMA kernel takes -- those layers stay exact.
static bool x3_runs(const odtk_conv_desc* d, int pass) {
    if (d->dtype != ODTK_F32X3 || g_force_regstage || g_v3_mode == 1 || (g_dbg2 & 4)) return false;
    const long long Min = (long long)d->N * d->H * d->W, Mout = (long long)d->N * d->Ho * d->Wo;
    const int ldc = pad8(d->C), ldk = pad8(d->K);
    const long long lim = (1ll << 31) - (1ll << 21);
    // (a stride-2 INPUT gradient is the exception: its exact kernel is the register-staged gather -- no LDS-DMA for the parity walk in f32 -- at 8-10 TFLOP/s;
    //  RetinaNet's 28 -> 56 shortcut at 200 x 200: 450 -> 259 us split)
    if ((long long)d->C * d->K * d->R * d->S < 20000 && !(g_dbg2 & 8) && !(pass == 2 && d->stride == 2 && d->R * d->S > 1)) return false;
    if (!(d->R * d->S <= 32 && (d->stride == 1 || (d->stride == 2 && d->dil == 1)))) return false;
    if (!(Min * (2 * ldc + 64) * 2 < lim && Mout * (2 * ldk + 64) * 2 < lim && 3 * Min * ldc * 2 < lim && 3 * Mout * ldk * 2 < lim && (long long)d->K * d->R * d->S * 3 * ldc * 2 < lim && (long long)d->C * d->R * d->S * 3 * ldk * 2 < lim &&
          3 * Min < (1ll << 31) && 3 * Mout < (1ll << 31) && Mout * d->ldy * 4 < (1ll << 31) && Min * d->ldx * 4 < (1ll << 31))) return false;
    if (pass == 1 && d->C % 8 != 0) return false;          // the filter gradient's rows are [K][R][S][C]: the bf16 kernels write whole 8-channel chunks
    return true;
}

extern "C" int odtk_conv2d_x3_supported(const odtk_conv_desc* d) {
    if (check_desc(d)) return 0;
    return (x3_runs(d, 0) ? 1 : 0) | (x3_runs(d, 1) ? 2 : 0) | (x3_runs(d, 2) ? 4 : 0);
}

static void x3_finish_args(GatherArgs& a) {
    a.div_howo = make_fastdiv((unsigned)(a.Ho * a.Wo)); a.div_wo = make_fastdiv((unsigned)a.Wo);
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 2);
    a.w_bytes = (unsigned)((size_t)a.K * a.ldw * 2);
    a.dbg = g_dbg; a.dbg2 = g_dbg2;
}

static int conv2d_fwd_x3(const odtk_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu, hipStream_t st) {
    const long long Min = (long long)d->N * d->H * d->W, Mout = (long long)d->N * d->Ho * d->Wo;
    const int ldc = pad8(d->C);
    odtk_conv_desc b = as_f32(d);
    b.C = 3 * ldc;                                          // the reduction sees [hi | hi | lo] ...
    b.ldx = x3_pitch(ldc);                                  // ... of pixel rows that store [hi | lo] (GatherArgs::x3c)
    GatherArgs a;
    fwd_args(a, &b, nullptr, nullptr, nullptr, nullptr, 0);
    a.x3c = ldc;
    x3_finish_args(a);
    const int ks = cv::gather_x3_ksplit(a);
    const size_t xs_b = up256((size_t)Min * x3_pitch(ldc) * 2), w3_b = up256((size_t)d->K * d->R * d->S * 3 * ldc * 2);
    char* base = nullptr;
    if (int e = cv::x3_scratch(xs_b + w3_b + (ks > 1 ? (size_t)ks * Mout * d->ldy * 4 : 0), st, &base)) return e;
    // pixels [hi | lo] (read as [hi | hi | lo]) and filters [K][R][S][hi | lo | hi]: ONE launch (round 6)
    cv::launch_split3_chan2(x, Min, d->C, d->ldx, base, ldc, 2, 2, x3_pitch(ldc),
                            w, (long long)d->K * d->R * d->S, d->C, d->C, base + xs_b, ldc, 2, 3, 3 * ldc, st);
    a.x = base; a.w = base + xs_b;
    cv::launch_gather_x3(a, y, (float*)(base + xs_b + w3_b), bias, relu, nullptr, 0, 0, st);
    g_last_kernel = a.ksplit == -9 ? "conv_gather_v9_kernel<x3>" : a.ksplit < 0 ? "conv_gather_v6_kernel<x3>" : "conv_gather_v3_kernel<x3>";
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

static int conv2d_dgrad_x3(const odtk_conv_desc* d, const float* dy, int lddy, const float* w_t, const float* relu_src, float* dx, int accumulate, hipStream_t st) {
    const long long Min = (long long)d->N * d->H * d->W, Mout = (long long)d->N * d->Ho * d->Wo;
    const int ldk = pad8(d->K);
    GatherArgs a;
    odtk_conv_desc f = as_f32(d);
    dgrad_args(a, &f, nullptr, 3 * ldk, nullptr, nullptr, nullptr, 0);
    a.ldx = x3_pitch(ldk);                                  // dy rows store [hi | lo], the reduction (a.C = 3 ldk) reads [hi | hi | lo]
    a.x3c = ldk;
    x3_finish_args(a);
    const int ks = cv::gather_x3_ksplit(a);
    const size_t dys_b = up256((size_t)Mout * x3_pitch(ldk) * 2), wt3_b = up256((size_t)d->C * d->R * d->S * 3 * ldk * 2);
    char* base = nullptr;
    if (int e = cv::x3_scratch(dys_b + wt3_b + (ks > 1 ? (size_t)ks * Min * d->ldx * 4 : 0), st, &base)) return e;
    // dy [hi | lo] (read as [hi | hi | lo]) and the caller's [C][R][S][lddy] flipped filters -> [hi | lo | hi]: ONE launch (round 6)
    cv::launch_split3_chan2(dy, Mout, d->K, lddy, base, ldk, 2, 2, x3_pitch(ldk),
                            w_t, (long long)d->C * d->R * d->S, d->K, lddy, base + dys_b, ldk, 2, 3, 3 * ldk, st);
    a.x = base; a.w = base + dys_b;
    cv::launch_gather_x3(a, dx, (float*)(base + dys_b + wt3_b), nullptr, 0, relu_src, d->ldx, accumulate, st);
    g_last_kernel = a.ksplit == -9 ? "conv_gather_v9_kernel<x3>" : a.ksplit < 0 ? "conv_gather_v6_kernel<x3>" : "conv_gather_v3_kernel<x3>";
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_colsum(const void* dy, int M, int C, int ld, int dtype, float* out, int accumulate, void* workspace, void* stream);
extern "C" int odtk_conv2d_wgrad(const odtk_conv_desc* d, const void* x, const void* dy, int lddy, float* dw, float* dbias, void* stream);

static int conv2d_wgrad_x3(const odtk_conv_desc* d, const float* x, const float* dy, int lddy, float* dw, float* dbias, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const long long Min = (long long)d->N * d->H * d->W, Mout = (long long)d->N * d->Ho * d->Wo;
    const int ldk = pad8(d->K);
    const size_t xr_b = up256((size_t)3 * Min * d->C * 2), dyr_b = up256((size_t)3 * Mout * ldk * 2), cs_b = (size_t)2 * 256 * ((d->K + 63) / 64 * 64) * 4;
    char* base = nullptr;
    if (int e = cv::x3_scratch(xr_b + dyr_b + cs_b, st, &base)) return e;
    // x images [hi ; lo ; hi] and dy images [hi ; hi ; lo]: ONE launch (round 6)
    cv::launch_split3_rows2(x, Min, d->C, d->ldx, base, d->C, 2, dy, Mout, d->K, lddy, base + xr_b, ldk, 4, st);
    odtk_conv_desc b = *d;
    b.N = 3 * d->N; b.dtype = b.out_dtype = ODTK_BF16; b.ldx = d->C; b.ldy = ldk;
    if (int e = odtk_conv2d_wgrad(&b, base, base + xr_b, ldk, dw, nullptr, stream)) return e;      // the bf16 filter-gradient kernels; their f32 result is dW
    if (dbias)                                                                                       // the bias gradient is a plain f32 column sum of dy
        return odtk_colsum(dy, (int)Mout, d->K, lddy, ODTK_F32, dbias, 1, base + xr_b + dyr_b, stream);
    return ODTK_OK;
}

extern "C" int odtk_conv2d_wgrad(const odtk_conv_desc* d, const void* x, const void* dy, int lddy,
                                 float* dw, float* dbias, void* stream) {
    if (int e = check_desc(d)) return e;
    ODTK_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
    const int kch = d->dtype == ODTK_BF16 ? 8 : 4;
    ODTK_REQUIRE(lddy % kch == 0 && lddy >= d->K, "conv2d_wgrad: lddy=%d must be a multiple of %d", lddy, kch);
    if (d->dtype == ODTK_F32X3) {
        if (x3_runs(d, 1)) return conv2d_wgrad_x3(d, (const float*)x, (const float*)dy, lddy, dw, dbias, stream);
        const odtk_conv_desc f = as_f32(d);
        return odtk_conv2d_wgrad(&f, x, dy, lddy, dw, dbias, stream);
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.dy = (const char*)dy; a.dw = dw; a.dbias = dbias;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.K = d->K; a.lddy = lddy;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.dil = d->dil; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.P = d->N * d->Ho * d->Wo; a.RSC = d->R * d->S * d->C;
    a.div_howo = make_fastdiv((unsigned)(d->Ho * d->Wo));
    a.div_wo = make_fastdiv((unsigned)d->Wo);
    a.dbg = g_dbg;
    a.dbg2 = g_dbg2;
    if ((uintptr_t)dw & 15) {
        // the single-split flush and the fixed-order reduction move dW as 16-byte rows (round-5 advisory): a gradient buffer that is not 16-byte aligned takes
        // the float-atomic flush, and the deterministic mode refuses it
        ODTK_REQUIRE(!cv::get_wgrad_deterministic(), "conv2d_wgrad: deterministic mode needs dw aligned to 16 bytes (got %p)", (void*)dw);
        a.dbg2 |= 4096;
    }
    if (!g_force_regstage && g_v3_mode != 1 && wgrad_f32_narrow_supported(a, d->dtype)) {
        launch_wgrad_f32_narrow(a, (hipStream_t)stream);
        g_last_kernel = "wgrad_f32_narrow_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && wgrad_c64_supported(a, d->dtype)) {
        launch_wgrad_c64(a, (hipStream_t)stream);
        g_last_kernel = "wgrad3x3_c64k64_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (!g_force_regstage && g_v3_mode != 1 && !(g_dbg & 2048) && wgrad_c8_supported(a, d->dtype)) {
        launch_wgrad_c8(a, (hipStream_t)stream);
        g_last_kernel = "wgrad3x3_c8k64_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    if (!g_force_regstage && g_v3_mode != 1 && wgrad_v3_supported(a, d->dtype)) {
        launch_wgrad_v3(a, (hipStream_t)stream);
        g_last_kernel = a.which == 8 ? "conv_wgrad_v8_kernel" : "conv_wgrad_v3_kernel";
        ODTK_LAUNCH_CHECK();
        return ODTK_OK;
    }
    const int PT = d->K <= 64 ? 64 : 128;
    a.tiles_p = ceil_div(d->K, PT);
    a.tiles_q = ceil_div(a.RSC, 128);
    const int pke = 8 * kch;
    const int iters_total = ceil_div(a.P, pke);
    const int tiles = a.tiles_p * a.tiles_q;
    int splits = 1024 / tiles;
    if (splits < 1) splits = 1;
    int max_splits = iters_total / 8;
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    a.iters_per_split = ceil_div(iters_total, splits);
    splits = ceil_div(iters_total, a.iters_per_split);
    a.div_howo = make_fastdiv((unsigned)(d->Ho * d->Wo));
    a.div_wo = make_fastdiv((unsigned)d->Wo);
    a.dbg = g_dbg;
    dim3 grid(tiles, splits);
    hipStream_t st = (hipStream_t)stream;
    if (int e = cv::wgrad_split_scratch(a, splits, dbias ? splits : 0, st)) return e;      // deterministic mode (key 5) with > 1 pixel split: partial tiles + reduction launch
    g_last_kernel = (d->dtype == ODTK_BF16 && !g_force_regstage && !(PT == 64 && g_v3_mode == 1)) ? "conv_wgrad_dma_kernel" : "conv_wgrad_kernel";
    if (d->dtype == ODTK_BF16) {
        if (PT == 64 && (g_v3_mode == 1 || g_force_regstage)) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 64>), grid, dim3(256), 0, st, a);
        else if (!g_force_regstage) hipLaunchKernelGGL(conv_wgrad_dma_kernel, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 128>), grid, dim3(256), 0, st, a);
    } else {
        if (PT == 64) hipLaunchKernelGGL((conv_wgrad_kernel<float, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<float, 128>), grid, dim3(256), 0, st, a);
    }
    cv::wgrad_split_reduce(a, st);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_filter_prepare(const float* w, int K, int R, int S, int C, int Kp, int dtype,
                                   void* w_c, void* w_t, void* stream) {
    ODTK_REQUIRE(w && (w_c || w_t), "filter_prepare: null pointer");
    ODTK_REQUIRE(Kp >= K, "filter_prepare: Kp < K");
    const int RS = R * S;
    const int kmax = w_t ? Kp : K;
    dim3 grid(RS * ceil_div(C, 32), ceil_div(kmax, 32));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_BF16)
        hipLaunchKernelGGL(filter_prepare_kernel<bf16_t>, grid, dim3(256), 0, st, w, K, RS, C, Kp, (bf16_t*)w_c, (bf16_t*)w_t);
    else if (dtype == ODTK_F32)
        hipLaunchKernelGGL(filter_prepare_kernel<float>, grid, dim3(256), 0, st, w, K, RS, C, Kp, (float*)w_c, (float*)w_t);
    else ODTK_REQUIRE(false, "filter_prepare: bad dtype %d", dtype);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}

extern "C" int odtk_filter_prepare_batched(const void* items_dev, int n_items, int total_blocks, int dtype,
                                           void* stream) {
    ODTK_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "filter_prepare_batched: bad argument");
    static_assert(sizeof(FpItem) == 48, "FpItem layout is part of the C-ABI");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ODTK_BF16)
        hipLaunchKernelGGL(filter_prepare_batched_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, (const FpItem*)items_dev, n_items);
    else if (dtype == ODTK_F32)
        hipLaunchKernelGGL(filter_prepare_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, st, (const FpItem*)items_dev, n_items);
    else ODTK_REQUIRE(false, "filter_prepare_batched: bad dtype %d", dtype);
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
