// Implicit-GEMM convolution, 8-wave / 3-stage LDS-ring generation (gfx950 only, bf16 operands).
//
// Same GEMM view and the same swizzled 128-byte-row slab layout as conv.hip, re-tiled for one
// 512-thread workgroup per CU:
//   * tile = PT channels x 256 pixels (PT = 64 | 128), 8 wave64 as 2 (channels) x 4 (pixels),
//     every wave owns (PT/2) x 64 of D -> the 2x2 / 1x2 grid of 32x32x16 MFMAs per k-substep;
//   * a k-slab is 64 k-elements (128 B per row); THREE slabs live in LDS (3 x 48 KiB), filled by
//     LDS-DMA (global_load_lds_dwordx4).  Slab t+2 is issued while slab t is consumed and the
//     wait in front of the slab barrier is a COUNTED s_waitcnt vmcnt(#DMA per slab): one whole
//     slab stays in flight across the barrier, so an HBM/L2 round trip has two slab-times to land;
//   * epilogue: accumulators (+bias) are rounded to bf16 into an XOR-swizzled [pixel][channel]
//     LDS image and leave as full 16-byte-per-lane row segments (256 B contiguous per pixel);
//     accumulate / ReLU / ReLU-mask are applied on that coalesced pass.
#include <mutex>
#include "conv_common.h"

namespace odtk {
namespace cv {
namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// Coalesced epilogue shared by the gather kernels: acc -> (bias) -> bf16 -> (ReLU) -> swizzled LDS image
// [QT pixels][PT channels] -> 16 B per lane -> (accumulate, relu, mask) -> global.
// `smem` must be free (all slab reads done, barrier passed) and hold QT*PT*2 bytes.
//
// Round 4 rewrite.  Switching the epilogue off showed what it costs with one workgroup per CU and nothing to overlap it: 31 % of conv2_1, 23-31 % of conv2_2,
// 12-19 % of conv3_2, 6-10 % of conv4_2 -- and two thirds of that was NOT the global stores but the instruction stream around them (1 376 VALU in the compiled
// epilogue of the 128 x 256 variant: 64-bit address arithmetic and bounds tests per 16-byte chunk, f32 bias add + f32 ReLU per value).  Now:
//   * BIAS_IN_ACC (raster-run halo kernel): the accumulators START at the bias, nothing is added here;
//   * ReLU is a packed signed-16-bit max on the ROUNDED pair (rounding keeps the sign; -0 -> +0): one op per two values;
//   * the image addresses of a lane are 4 PI swizzle terms + instruction-offset immediates (a 32-pixel step leaves (pixel & 15) alone);
//   * the store pass is buffer-addressed: one per-lane byte offset, the iteration's row step in an SGPR, rows >= M and pitch-tail chunks dropped by the
//     hardware range check -- per iteration one ds_read_b128 and one buffer_store_dwordx4 (plus the loads / chunk post-ops of a masked or accumulating pass).
// ---------------------------------------------------------------------------------------
typedef unsigned u32x4_v __attribute__((ext_vector_type(4)));
typedef short s16x2e_v __attribute__((ext_vector_type(2)));

// POOL (raster-run halo kernel on row-pair tiles): the tile holds `rows_here` whole rows of image `pn` starting at an even row (pooled row ph0); only its first
// rows_here * W pixels are this tile's, and after the store pass (skipped when pool_mode == 2: the un-pooled map is never stored) the 2 x 2 / stride-2
// 'SAME' max pooling of those rows leaves from the same LDS image -- pooled chunk + recorded first arg-max, the arithmetic of conv3x3_c64k64_kernel's fused pool.
// PHASE (round 5; the 8-wave kernel on the parity phases of a stride-2 input gradient): the tile's pixels are a run of ONE phase's pixel grid (GatherArgs::v9[pn]); pixel
// ml of it is output row ((n Ho + 2 i + ph) Wo + 2 j + pw) -- the store pass (and its accumulate / mask loads) addresses every row through that map.
template <int PT, int QT, int NTHR, int PI, int QI, bool BIAS_IN_ACC = false, bool POOL = false, bool PHASE = false>
__device__ __forceinline__ void epilogue_bf16(const GatherArgs& a, char* smem, f32x16_v (&acc)[PI][QI],
                                              int p0, int q0, int prow0, int qrow0, int tid, int pn = 0, int ph0 = 0, int rows_here = 0) {
    constexpr int RB = PT * 2;            // bytes per pixel row of the image
    constexpr int NCH = RB / 16;          // 16-B chunks per row (8 | 16)
    constexpr int NIT = (QT * NCH) / NTHR;
    constexpr int RPI = NTHR / NCH;       // pixel rows per store-pass iteration (a multiple of NCH: the read swizzle is the same in every iteration)
    static_assert(RPI % NCH == 0 && (QT * NCH) % NTHR == 0, "store pass: whole iterations, iteration step a multiple of the swizzle period");
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    float4 bv[PI][4];
    if (!BIAS_IN_ACC) {
        // (every global load of the epilogue is issued BEFORE its first use: the bias of this lane's 4 * PI channel groups here, the accumulate /
        //  ReLU-mask operands of the store pass below in one batch)
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = p0 + prow0 + i * 32 + 8 * g + 4 * hi;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) {
                    if (c + 3 < a.K) b = *reinterpret_cast<const float4*>(a.bias + c);
                    else {
                        if (c < a.K) b.x = a.bias[c];
                        if (c + 1 < a.K) b.y = a.bias[c + 1];
                        if (c + 2 < a.K) b.z = a.bias[c + 2];
                    }
                }
                bv[i][g] = b;
            }
    }
    // ReLU without accumulate is applied here, on the rounded pair (with accumulate it follows the sum, in post_chunk)
    const s16x2e_v floor16 = (a.relu && !a.accumulate) ? (s16x2e_v)(0) : (s16x2e_v)(-32768);
    {
        const int qq = (qrow0 + l31) & (NCH - 1);                     // the pixel's swizzle bits: the same for every j (32-pixel steps)
        char* rowp = smem + (qrow0 + l31) * RB;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = prow0 + i * 32 + 8 * g + 4 * hi;      // channel inside the tile
                char* dst = rowp + ((((cl >> 3) ^ qq) & (NCH - 1)) << 4) + ((cl & 4) << 1);
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                    if (!BIAS_IN_ACC) { v0 += bv[i][g].x; v1 += bv[i][g].y; v2 += bv[i][g].z; v3 += bv[i][g].w; }
                    uint2 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2e_v, cvt_pk_bf16(v0, v1)), floor16));
                    o.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2e_v, cvt_pk_bf16(v2, v3)), floor16));
                    *reinterpret_cast<uint2*>(dst + j * 32 * RB) = o;
                }
            }
        }
    }
    // store pass: chunk (pixel qt + it * RPI, 16-byte column ch) per lane and iteration
    const int qt = tid / NCH, ch = tid % NCH;
    const int c0 = p0 + ch * 8;
    const long long ybytes = (long long)a.M * a.ldy * 2;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, (unsigned)ybytes);
    // (the row step goes through the per-lane offset, not the instruction's scalar offset: the range check compares offset with num_records - soffset, which
    //  wraps for tiles that reach more than a whole tensor past the end; a pitch-tail lane keeps its out-of-range sentinel by stepping 0)
    const unsigned voff = c0 < a.ldy ? (unsigned)(((long long)(q0 + qt) * a.ldy + c0) * 2) : 0xFFFFFFF0u;      // (>= ybytes: rows past M are dropped / read as 0)
    const unsigned sstep = c0 < a.ldy ? (unsigned)(RPI * a.ldy * 2) : 0u;
    unsigned prow[PHASE ? NIT : 1];                      // PHASE: output row of iteration it (0xFFFFFFFF: past the phase's pixels)
    if (PHASE) {
        const GatherArgs::V9Phase& P = a.v9[pn];
        const int ph = pn >> 1, pw = pn & 1, hw = P.Hq * P.Wq;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ml = q0 + qt + it * RPI;
            prow[it] = 0xFFFFFFFFu;
            if (ml < P.Mq) {
                const int n = (int)fdiv((unsigned)ml, P.d_hw), rem = ml - n * hw;
                const int hi_ = (int)fdiv((unsigned)rem, P.d_w), wi_ = rem - hi_ * P.Wq;
                prow[it] = (unsigned)((n * a.Ho + 2 * hi_ + ph) * a.Wo + 2 * wi_ + pw);
            }
        }
    }
    auto yoff = [&](int it) __attribute__((always_inline)) -> unsigned {
        if constexpr (!PHASE) return voff + it * sstep;
        else return (prow[it] != 0xFFFFFFFFu && c0 < a.ldy) ? (prow[it] * (unsigned)a.ldy + (unsigned)c0) * 2u : 0xFFFFFFF0u;
    };
    const bool post = a.accumulate || a.mask;
    u32x4_v oldv[NIT], mkv[NIT];
    if (post) {
        // operands of the store pass (independent of the image): issued now, they land under the barrier
        const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.mask ? a.mask : a.y, (unsigned)((long long)a.M * a.ldmask * 2));
        const unsigned moff = c0 < a.ldmask ? (unsigned)(((long long)(q0 + qt) * a.ldmask + c0) * 2) : 0xFFFFFFF0u;
        const unsigned mstep = c0 < a.ldmask ? (unsigned)(RPI * a.ldmask * 2) : 0u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            oldv[it] = (u32x4_v)(0u); mkv[it] = (u32x4_v)(0u);
            if (a.accumulate) oldv[it] = __builtin_amdgcn_raw_buffer_load_b128(ry, yoff(it), 0, 0);
            if (a.mask) {
                unsigned mo = moff + it * mstep;
                if constexpr (PHASE) mo = (prow[it] != 0xFFFFFFFFu && c0 < a.ldmask) ? (prow[it] * (unsigned)a.ldmask + (unsigned)c0) * 2u : 0xFFFFFFF0u;
                mkv[it] = __builtin_amdgcn_raw_buffer_load_b128(rm, mo, 0, 0);
            }
        }
    }
    __syncthreads();
    const char* src = smem + qt * RB + (((ch ^ qt) & (NCH - 1)) << 4);
    // in batches of EB chunks: EB LDS reads in flight, then EB stores (chunk by chunk the compiler waits for every read: ~120 exposed cycles x NIT per lane)
    constexpr int EB = NIT % 8 == 0 ? 8 : (NIT % 4 == 0 ? 4 : 1);
    if (!POOL || a.pool_mode != 2) {
        const int valid_px = POOL ? rows_here * a.W : QT;       // (row-pair tiles: the pixels behind the tile's rows belong to the next tile)
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += EB) {
            uint4 v[EB];
#pragma unroll
            for (int e = 0; e < EB; ++e) v[e] = *reinterpret_cast<const uint4*>(src + (it0 + e) * RPI * RB);
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                if (post) post_chunk(v[e], a.accumulate != 0, a.relu != 0, __builtin_bit_cast(uint4, oldv[it0 + e]), a.mask != nullptr, __builtin_bit_cast(uint4, mkv[it0 + e]));
                const unsigned off = (!POOL || qt + (it0 + e) * RPI < valid_px) ? yoff(it0 + e) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_v, v[e]), ry, off, 0, 0);
            }
        }
    }
    if (POOL) {
        typedef unsigned short u16x2p_v __attribute__((ext_vector_type(2)));
        const int Wp = (a.W + 1) >> 1, Hp = (a.H + 1) >> 1;
        const int ntask = ((rows_here + 1) >> 1) * Wp * NCH;
        const int kchunks = a.K >> 3;
        for (int task = tid; task < ntask; task += NTHR) {
            const int pch = task % NCH, pp = task / NCH;
            const int pr = (int)fdiv((unsigned)pp, a.div_wp), pw = pp - pr * Wp;
            const int hl = 2 * pr, w = 2 * pw;
            if (p0 + pch * 8 >= a.K) continue;
            // candidates outside the image are zeroed (they can tie but never win: the values are ReLU outputs) and read the window's first pixel instead
            const bool b1 = w + 1 < a.W, b2 = hl + 1 < rows_here;
            const unsigned in1 = b1 ? 0xFFFFFFFFu : 0u, in2 = b2 ? 0xFFFFFFFFu : 0u;
            const int qa0 = hl * a.W + w;
            const int qs[4] = {qa0, b1 ? qa0 + 1 : qa0, b2 ? qa0 + a.W : qa0, (b1 && b2) ? qa0 + a.W + 1 : qa0};
            uint4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const uint4*>(smem + qs[t] * RB + (((pch ^ qs[t]) & (NCH - 1)) << 4));
            const unsigned* u0 = reinterpret_cast<const unsigned*>(&v[0]);
            const unsigned* u1 = reinterpret_cast<const unsigned*>(&v[1]);
            const unsigned* u2 = reinterpret_cast<const unsigned*>(&v[2]);
            const unsigned* u3 = reinterpret_cast<const unsigned*>(&v[3]);
            uint4 best;
            unsigned* ub = reinterpret_cast<unsigned*>(&best);
            unsigned code = 0;
            const u16x2p_v one = (u16x2p_v)(1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // non-negative bf16 order like unsigned 16-bit integers; FIRST maximum in scan order: row = (max of row 1 > max of row 0), column = (right > left)
                const u16x2p_v x0 = __builtin_bit_cast(u16x2p_v, u0[q]), x1 = __builtin_bit_cast(u16x2p_v, u1[q] & in1);
                const u16x2p_v x2 = __builtin_bit_cast(u16x2p_v, u2[q] & in2), x3 = __builtin_bit_cast(u16x2p_v, u3[q] & in1 & in2);
                const u16x2p_v m01 = __builtin_elementwise_max(x0, x1), m23 = __builtin_elementwise_max(x2, x3);
                const u16x2p_v row = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m23, m01), one);
                const u16x2p_v c01 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(x1, x0), one);
                const u16x2p_v c23 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(x3, x2), one);
                const u16x2p_v rmask = (u16x2p_v)(0) - row;
                const u16x2p_v am = ((c23 & rmask) | (c01 & ~rmask)) | (row << 1);
                const unsigned cu = __builtin_bit_cast(unsigned, am);
                code |= ((cu | (cu >> 14)) & 0xFu) << (4 * q);
                ub[q] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(m01, m23));
            }
            const size_t mo = (size_t)(pn * Hp + ph0 + pr) * Wp + pw;
            *reinterpret_cast<uint4*>(a.ypool + (mo * a.ldpool + p0 + pch * 8) * 2) = best;
            a.pidx[mo * kchunks + (p0 >> 3) + pch] = (unsigned short)code;
        }
    }
}

// ---------------------------------------------------------------------------------------
// gather kernel (forward conv, stride-1 dgrad)
// ---------------------------------------------------------------------------------------
// PHASE (round 5; with BUF, C64, no split-K): the four parity phases of a stride-2 input gradient in one launch (GatherArgs::v9, the table the small-map kernel uses):
// a tile's pixels belong to one phase, only that phase's taps are walked -- 9 tap-slabs per four output pixels instead of 36, a quarter of the MFMA work.
template <int PT, bool DB, bool EARLY, bool BUF, bool C64 = false, bool SPLIT = false, bool ILV = false, bool PHASE = false>
__global__ void __launch_bounds__(512) conv_gather_v3_kernel(const GatherArgs a) {
    constexpr int QT = 256;
    constexpr int PI = PT / 64, QI = 2, PL = PT / 64;
    constexpr int STAGE = (PT + QT) * 128;
    constexpr int NST = 3;
    constexpr int NDMA = 4 + PL;                       // LDS-DMA pieces per wave per slab
    static_assert(QT * PT * 2 <= NST * STAGE, "epilogue image must fit");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave & 1, wq = wave >> 1;
    // SPLIT (split-K for the small, latency-bound layers): `ksplit` consecutive blocks share one tile, block
    // `part` reduces k-slabs [part*nk/ksplit, (part+1)*nk/ksplit) and writes its f32 partial tile to a.ws[part];
    // splitk_finish_kernel sums the parts in fixed order and applies bias / ReLU / mask / accumulate
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int part = SPLIT ? lin % a.ksplit : 0;
    const int vb = SPLIT ? lin / a.ksplit : lin;
    int tq = vb / a.tiles_p;
    const int tp = vb - tq * a.tiles_p;
    int pi = 0, ph = 0, pw = 0, Hq = a.Ho, Wq = a.Wo, Mq = a.M, r_first = 0, s_first = 0;
    FastDiv d_hw = a.div_howo, d_w = a.div_wo;
    int nk_all = (a.Kdim + 63) >> 6;
    if (PHASE) {
        pi = (tq >= a.v9[1].tile0) + (tq >= a.v9[2].tile0) + (tq >= a.v9[3].tile0);
        const GatherArgs::V9Phase& P = a.v9[pi];
        ph = pi >> 1; pw = pi & 1;
        tq -= P.tile0;
        Hq = P.Hq; Wq = P.Wq; Mq = P.Mq; r_first = P.r0; s_first = P.s0; nk_all = P.nk;
        d_hw = P.d_hw; d_w = P.d_w;
    }
    const int p0 = tp * PT, q0 = tq * QT;
    const int ks0 = SPLIT ? part * nk_all / a.ksplit : 0;
    const int nk = SPLIT ? (part + 1) * nk_all / a.ksplit - ks0 : ((a.dbg & 8) ? 1 : nk_all);

    const int r0 = tid >> 3;                           // DMA rows r0 + 64*i
    const int cc = (tid & 7) ^ swz_g(r0);              // logical 16-B chunk this lane fetches
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // per pixel row: byte offset of tap (0,0) channel 0, and the bit mask of in-range taps
    long long qoff[4];
    unsigned qoff32[4];
    unsigned qmask[4];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rw = make_rsrc(a.w, a.w_bytes);
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = q0 + r0 + 64 * i;
        qoff[i] = 0; qmask[i] = 0; qoff32[i] = 0;
        if (m < Mq) {
            const int n = (int)fdiv((unsigned)m, d_hw), rem = m - n * (PHASE ? Hq * Wq : HoWo);
            const int ho_ = (int)fdiv((unsigned)rem, d_w), wo_ = rem - ho_ * Wq;
            const int ho = PHASE ? 2 * ho_ + ph : ho_, wo = PHASE ? 2 * wo_ + pw : wo_;
            const int hb = ho * a.ostride - a.pad_t, wb = wo * a.ostride - a.pad_l;
            // stride-2 dgrad (idiv == 2, dil == 1): tap r reads dy row (hb + r) / 2 when that is an integer
            // = (hb >> 1) + ((r + 1) >> 1) for every parity-matching r, so the "base + tap offset" form survives
            // with the halved base and tap steps ((r + 1) >> 1); mismatching taps are masked like padding
            const int hq = a.idiv == 2 ? hb >> 1 : hb, wq = a.idiv == 2 ? wb >> 1 : wb;
            qoff[i] = ((long long)(n * a.H + hq) * a.W + wq) * a.ldx * 2ll;
            qoff32[i] = (unsigned)(((n * a.H + hq) * a.W + wq) * a.ldx * 2);     // may wrap below 0: only used with in-range taps
            unsigned rm = 0, cm = 0;
            for (int r = 0; r < a.R; ++r) {
                const int hn = hb + r * a.dil;
                if (a.idiv == 2 ? (hn >= 0 && !(hn & 1) && (hn >> 1) < a.H) : (unsigned)hn < (unsigned)a.H) rm |= 1u << r;
            }
            for (int s2 = 0; s2 < a.S; ++s2) {
                const int wn = wb + s2 * a.dil;
                if (a.idiv == 2 ? (wn >= 0 && !(wn & 1) && (wn >> 1) < a.W) : (unsigned)wn < (unsigned)a.W) cm |= 1u << s2;
            }
            unsigned mk = 0;
            for (int r = 0; r < a.R; ++r)
                if ((rm >> r) & 1u) mk |= cm << (r * a.S);
            qmask[i] = mk;
        }
    }
    long long poff[PL];
    unsigned poff32[PL];
    bool pok[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int row = p0 + r0 + 64 * i;
        pok[i] = row < a.K;
        poff[i] = (long long)row * a.ldw * 2ll;
        poff32[i] = (unsigned)(row * a.ldw * 2);
    }
    int klin = ks0 * 64 + cc * 8;
    int kc, ks, kr;
    {
        const int rs = klin / a.C;
        kc = klin - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }

    // C64 (C % 64 == 0, buffer addressing): a k-slab never straddles a tap, so the tap walk is wave-uniform
    // (SALU) and a piece costs this lane an AND, a compare, an add and a select.
    int s_klin = ks0 * 64, s_kc, s_ks, s_kr;
    {
        const int rs = s_klin / a.C;
        s_kc = s_klin - rs * a.C;
        s_kr = rs / a.S;
        s_ks = rs - s_kr * a.S;
    }
    if (PHASE) { s_kr = r_first; s_ks = s_first; s_kc = 0; }     // the phase's first tap; taps advance by 2 (same parity)
    if (C64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qoff32[i] += (unsigned)(cc * 16);
#pragma unroll
        for (int i = 0; i < PL; ++i) poff32[i] += (unsigned)(cc * 16);
    }
    auto issue = [&](int stage) {
        const unsigned sP = smem_base + (unsigned)stage * STAGE + wave_u * 1024u;
        const unsigned sQ = sP + PT * 128;
        if (C64) {
            const int sr = a.idiv == 2 ? (s_kr + 1) >> 1 : s_kr * a.dil, ss = a.idiv == 2 ? (s_ks + 1) >> 1 : s_ks * a.dil;
            const int xkc = (SPLIT && a.x3c && s_kc >= a.x3c) ? s_kc - a.x3c : s_kc;       // x3 (round 6: the phase launch): the third part re-reads the first part's channels
            const unsigned toff32 = (unsigned)((sr * a.W + ss) * a.ldx * 2 + xkc * 2);
            const unsigned tapbit = 1u << (s_kr * a.S + s_ks);
            const unsigned woff = PHASE ? (unsigned)(((s_kr * a.S + s_ks) * a.C + s_kc) * 2) : (unsigned)(s_klin * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned addr = qoff32[i] + toff32;
                glds16_buf(rx, (qmask[i] & tapbit) ? addr : 0xFFFFFFF0u, sQ + i * 8192u);
            }
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                const unsigned addr = poff32[i] + woff;
                glds16_buf(rw, pok[i] ? addr : 0xFFFFFFF0u, sP + i * 8192u);
            }
            s_klin += 64;
            s_kc += 64;
            if (s_kc >= a.C) {
                s_kc = 0;
                if (PHASE) { s_ks += 2; if (s_ks >= a.S) { s_ks = s_first; s_kr += 2; } }
                else if (++s_ks == a.S) { s_ks = 0; ++s_kr; }
            }
            return;
        }
        const bool kv = kr < a.R;
        const int tap = kr * a.S + ks;
        const int tr = a.idiv == 2 ? (kr + 1) >> 1 : kr * a.dil, ts = a.idiv == 2 ? (ks + 1) >> 1 : ks * a.dil;
        const long long toff = ((long long)tr * a.W + ts) * a.ldx * 2ll + (long long)kc * 2ll;
        const bool first = klin < 64;
        if (BUF) {
            const int kcx = (SPLIT && a.x3c && kc >= a.x3c) ? kc - a.x3c : kc;           // x3: the pixel operand holds [hi | lo], read as [hi | hi | lo]
            const unsigned toff32 = (unsigned)((tr * a.W + ts) * a.ldx * 2 + kcx * 2);
            const unsigned tapbit = kv ? (1u << tap) : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                glds16_buf(rx, (qmask[i] & tapbit) ? qoff32[i] + toff32 : 0xFFFFFFF0u, sQ + i * 8192u);
#pragma unroll
            for (int i = 0; i < PL; ++i)      // rows >= K and the k tail fetch zeros
                glds16_buf(rw, (kv && pok[i]) ? poff32[i] + (unsigned)(klin * 2) : 0xFFFFFFF0u, sP + i * 8192u);
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = kv && ((qmask[i] >> tap) & 1u);
            const char* src = ok ? a.x + qoff[i] + toff : zero;
            if ((a.dbg & 1) && !first) src = zero;
            glds16(src, sQ + i * 8192u);
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const bool ok = kv && pok[i];
            const char* src = ok ? a.w + poff[i] + (long long)klin * 2ll : zero;
            if ((a.dbg & 2) && !first) src = zero;
            glds16(src, sP + i * 8192u);
        }
        }
        klin += 64;
        kc += 64;
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

    if (EARLY) {
        // "Landed one slab early" protocol: at the top of iteration kt slabs kt AND kt+1 are published, so the
        // first fragments of slab kt+1 are read BEFORE barrier kt+1 and the matrix pipe restarts right behind
        // the barrier instead of waiting for an LDS round trip.  Price: a slab has one iteration (not two) to land.
        const int l31 = lane & 31, hi = lane >> 5;
        const int prow0 = wp * (PT / 2), qrow0 = wq * 64;
        uint4 pfA[PI], qfA[QI], pfB[PI], qfB[QI];
        auto ldf = [&](const char* sP, int ks, uint4 (&pf)[PI], uint4 (&qf)[QI]) __attribute__((always_inline)) {
            const char* sQ = sP + PT * 128;
            const int slot = ks * 2 + hi;
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int row = prow0 + i * 32 + l31;
                pf[i] = *reinterpret_cast<const uint4*>(sP + row * 128 + ((slot ^ swz_g(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < QI; ++j) {
                const int row = qrow0 + j * 32 + l31;
                qf[j] = *reinterpret_cast<const uint4*>(sQ + row * 128 + ((slot ^ swz_g(row)) << 4));
            }
        };
        auto mm = [&](uint4 (&pf)[PI], uint4 (&qf)[QI]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < PI; ++i)
#pragma unroll
                for (int j = 0; j < QI; ++j) Mma<bf16_t>::run(pf[i], qf[j], acc[i][j]);
        };
        const bool late = wave_u >= 4;
        issue(0);
        if (nk > 1) issue(1);
        wait_vmcnt<0>();
        block_barrier();                                 // slabs 0 and 1 published
        if (nk > 2) issue(2);
        ldf(smem, 0, pfA, qfA);
        int st_c = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const char* sP = smem + st_c * STAGE;
            const int st_1 = st_c == 2 ? 0 : st_c + 1;
            const bool do_issue = kt > 0 && kt + 2 < nk;          // refills stage of slab kt-1 with slab kt+2
            const int st_prev = st_c == 0 ? 2 : st_c - 1;
            if (do_issue && !late) issue(st_prev);
            ldf(sP, 1, pfB, qfB);
            mm(pfA, qfA);
            ldf(sP, 2, pfA, qfA);
            mm(pfB, qfB);
            if (do_issue && late) issue(st_prev);
            ldf(sP, 3, pfB, qfB);
            mm(pfA, qfA);
            if (kt + 1 < nk) ldf(smem + st_1 * STAGE, 0, pfA, qfA);   // slab kt+1 was published one barrier ago
            mm(pfB, qfB);
            wait_vmcnt<0>();                              // my pieces of slab kt+2 landed
            block_barrier();                              // slab kt+2 published; slab kt fully consumed
            st_c = st_1;
        }
    } else if (ILV && C64 && BUF) {
    // Interleaved variant: the NDMA pieces of slab kt+2 and the fragment reads of sub-step ks+1 are dealt out
    // one at a time BETWEEN the MFMAs of slab kt (sched_barrier pins the order), so no wave ever sits in a burst
    // of DMA issues while the matrix pipe of its SIMD drains, and the L1/TA path sees a steady trickle.
    issue(0);
    if (nk > 1) issue(1);
    const int l31 = lane & 31, hi = lane >> 5;
    const int prow0 = wp * (PT / 2), qrow0 = wq * 64;
    unsigned pofs[PI], qofs[QI];                        // LDS byte offsets of this lane's fragment rows (slot 0)
#pragma unroll
    for (int i = 0; i < PI; ++i) { const int row = prow0 + i * 32 + l31; pofs[i] = (unsigned)(row * 128) | ((unsigned)swz_g(row) << 16); }
#pragma unroll
    for (int j = 0; j < QI; ++j) { const int row = qrow0 + j * 32 + l31; qofs[j] = (unsigned)(row * 128) | ((unsigned)swz_g(row) << 16); }
    int st_c = 0, st_n = 2;
    auto slab = [&](auto ISSUE, int stage_c, int stage_n) __attribute__((always_inline)) {
        constexpr bool do_issue = decltype(ISSUE)::value;
        const char* sP = smem + stage_c * STAGE;
        const char* sQ = sP + PT * 128;
        uint4 pf[2][PI], qf[2][QI];
        auto rd = [&](int ks, int r) __attribute__((always_inline)) {       // read #r of sub-step ks: P0.., then Q0..
            const int slot = ks * 2 + hi;
            if (r < PI) pf[ks & 1][r] = *reinterpret_cast<const uint4*>(sP + (pofs[r] & 0xFFFFu) + (((unsigned)slot ^ (pofs[r] >> 16)) << 4));
            else qf[ks & 1][r - PI] = *reinterpret_cast<const uint4*>(sQ + (qofs[r - PI] & 0xFFFFu) + (((unsigned)slot ^ (qofs[r - PI] >> 16)) << 4));
        };
        // DMA pieces of this wave: 0..3 pixel rows r0 + 64 i, 4.. filter rows
        const unsigned dP = smem_base + (unsigned)stage_n * STAGE + wave_u * 1024u;
        const unsigned dQ = dP + PT * 128;
        const int sr = a.idiv == 2 ? (s_kr + 1) >> 1 : s_kr * a.dil, ss = a.idiv == 2 ? (s_ks + 1) >> 1 : s_ks * a.dil;
        const unsigned toff32 = (unsigned)((sr * a.W + ss) * a.ldx * 2 + s_kc * 2);
        const unsigned tapbit = 1u << (s_kr * a.S + s_ks);
        const unsigned woff = (unsigned)(s_klin * 2);
        auto piece = [&](int q) __attribute__((always_inline)) {
            if (q < 4) {
                const unsigned addr = qoff32[q] + toff32;
                glds16_buf_nc(rx, (qmask[q] & tapbit) ? addr : 0xFFFFFFF0u, dQ + q * 8192u);
            } else {
                const unsigned addr = poff32[q - 4] + woff;
                glds16_buf_nc(rw, pok[q - 4] ? addr : 0xFFFFFFF0u, dP + (q - 4) * 8192u);
            }
        };
#pragma unroll
        for (int r = 0; r < PI + QI; ++r) rd(0, r);
        static_for<4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int i = 0; i < PI; ++i)
#pragma unroll
                for (int j = 0; j < QI; ++j) {
                    const int mi = i * QI + j;                       // MFMA # inside the sub-step
                    const int slotno = ks * (PI * QI) + mi;          // ... inside the slab (0..4*PI*QI-1)
                    Mma<bf16_t>::run(pf[ks & 1][i], qf[ks & 1][j], acc[i][j]);
                    if (ks < 3) {
#pragma unroll
                        for (int r = 0; r < PI + QI; ++r)
                            if (r * (PI * QI) / (PI + QI) == mi) rd(ks + 1, r);
                    }
                    if (do_issue) {                                  // piece q goes out behind MFMA slot q * T / NDMA
#pragma unroll
                        for (int q = 0; q < NDMA; ++q)
                            if (q * (4 * PI * QI) / NDMA == slotno) piece(q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        });
        if (do_issue) {
            s_klin += 64;
            s_kc += 64;
            if (s_kc >= a.C) {
                s_kc = 0;
                if (++s_ks == a.S) { s_ks = 0; ++s_kr; }
            }
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
        block_barrier();
        if (kt + 2 < nk) slab(std::integral_constant<bool, true>{}, st_c, st_n);
        else slab(std::integral_constant<bool, false>{}, st_c, st_n);
        st_c = st_c == 2 ? 0 : st_c + 1;
        st_n = st_n == 2 ? 0 : st_n + 1;
    }
    block_barrier();
    } else {
    issue(0);
    if (nk > 1) issue(1);
    int st_c = 0, st_n = 2;                             // stage consumed now / stage to refill
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();   // this wave's pieces of slab kt landed
        block_barrier();            // ... everybody's did; everybody finished reading slab kt-1 (= stage st_n)
        // Waves w and w+4 share a SIMD.  The first half issues its LDS-DMA pieces BEFORE its MFMAs, the second
        // half AFTER them, so that on every SIMD one wave feeds the matrix pipe while its partner is blocked in
        // the (100+ cycles per piece) DMA issue path -- otherwise all eight waves leave the barrier in the same
        // phase and the two phases serialise (dbg bit 6 restores that order for A/B runs).
        const bool do_issue = kt + 2 < nk && !((a.dbg & 4) && kt > 0);
        const bool late = wave_u >= 4 && !(a.dbg & 64);
        if (do_issue && !late) issue(st_n);
        const char* sP = smem + st_c * STAGE;
        const bool prio = (a.dbg & 1024) != 0;          // A/B: raise the wave's priority over its MFMA cluster
        if (prio) __builtin_amdgcn_s_setprio(1);
        if (DB) mma_slab_db<bf16_t, PI, QI>(sP, sP + PT * 128, wp * (PT / 2), wq * 64, lane, acc);
        else mma_slab<bf16_t, PI, QI, true>(sP, sP + PT * 128, wp * (PT / 2), wq * 64, lane, acc);
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (do_issue && late) issue(st_n);
        st_c = st_c == 2 ? 0 : st_c + 1;
        st_n = st_n == 2 ? 0 : st_n + 1;
    }
    if (PHASE && nk == 0) wait_vmcnt<0>();               // (a phase without taps -- 1 x 1 / stride 2 -- still issued its first pieces: they must land before the image)
    block_barrier();                                    // all slab reads done: LDS is free for the output image
    }
    if (SPLIT) {
        // f32 partial tile -> a.ws[part][m][ldy]: 16 B (4 channels) per lane and accumulator group
        const int l31 = lane & 31, hi = lane >> 5;
        float* wsp = a.ws + (size_t)part * a.M * a.ldy;
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            int m = q0 + wq * 64 + j * 32 + l31;
            if (m >= Mq) continue;
            if (PHASE) {                                  // x3 phase launch (round 6, single part): the tile's pixels are a run of ONE phase's grid -> output row
                const int n = (int)fdiv((unsigned)m, d_hw), rem = m - n * (Hq * Wq);
                const int hq_ = (int)fdiv((unsigned)rem, d_w), wq_ = rem - hq_ * Wq;
                m = (n * a.Ho + 2 * hq_ + ph) * a.Wo + 2 * wq_ + pw;
            }
#pragma unroll
            for (int i = 0; i < PI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = p0 + wp * (PT / 2) + i * 32 + 8 * g + 4 * hi;
                    if (c < a.ldy) {
                        float o[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        if (a.ksplit == 1) x3_store4(a, m, c, o);      // the x3 engine's single-part launch writes the f32 OUTPUT: no finish pass
                        else *reinterpret_cast<float4*>(wsp + (size_t)m * a.ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
        }
        return;
    }
    epilogue_bf16<PT, QT, 512, PI, QI, false, false, PHASE>(a, smem, acc, p0, q0, wp * (PT / 2), wq * 64, tid, pi);
}

// ---------------------------------------------------------------------------------------
// "v6": raster-run halo gather for 3x3 / stride 1 / pad 1 layers (C % 64 == 0, W <= 79): the v5 tile and schedule, but
// the PIXEL operand is no longer gathered once per tap.  The 256 output pixels of a tile are a contiguous run m0.. of the
// (n, h, w) raster, so the inputs of ALL nine taps lie in the contiguous run m0 - d(W + 1) .. m0 + 255 + d(W + 1): that patch
// (<= 416 rows x 64 channels) is LDS-DMA'd ONCE per 64-channel chunk (double buffered, 13 pieces per wave spread over
// the nine tap slabs of the previous chunk) and tap (dr, ds) reads its fragments at row + dr*W + ds.  Row wrap, image
// borders and image-to-image seams are handled at fragment-read time: a lane whose (pixel, tap) is padding reads a
// zero row instead (per-pixel 9-bit tap masks, computed once).  Only the 16-KiB filter slab is fetched per tap:
// 196 KiB instead of 432 KiB of LDS-DMA per chunk.  slot ^ ((row >> 1) & 7) keeps ds_read_b128 conflict-free for any
// start row.  k order = (chunk, tap); the filter k index stays tap * C + channel.
// ---------------------------------------------------------------------------------------
// Template: NPP patch pieces per wave (patch rows = 32 NPP).  DBUF = true (NPP = 13, W <= 79): two patch buffers, the next
// chunk's pieces ride on the tap slabs of the current one.  DBUF = false (NPP = 18, W <= 159: conv2_x at 150 x 150): ONE patch
// buffer of 576 rows; the next chunk's patch is written into rows that are already DEAD -- tap row dr only reads patch rows
// >= dr * dil * W, so groups i < G1 = floor(dil W / 32) are issued on the slab of tap 3, groups < G2 = floor(2 dil W / 32) on
// tap 6, and only the last NPP - G2 groups (258 rows) are exposed between two chunks.
// WP x (4 / WP) waves, every wave PI x QI accumulator tiles of 32 x 32: 2, 2, 4 = the 128 x 256 tile (0.75 fragment reads per MFMA);
// 1, 4, 4 = a 128 x 512 tile of four 128 x 128 wave tiles (0.5 reads per MFMA, round 2: the 75- and 150-pixel maps, whose time is LDS
// bandwidth); 2, 2, 3 = 128 x 192 (conv5_x: 244 instead of 184 tiles for 256 CUs); 1, 2, 4 = 64 x 512 for Cout <= 64
// (conv2_1's input gradient, 128 -> 64 channels at W = 150: it ran on the 8-wave gather kernel at 490 TFLOP/s).
// (Round 4, measured and removed: a FOURTH filter stage where the patch leaves 16 KiB free -- conv4_x on an 11-group patch pair, the single-buffer variants of
// W = 75 -- with slab kt + 3 issued and two slabs' pieces in flight across a barrier: bit-identical, 1-3 % SLOWER on every layer, profiles/r04k_*.)
// (Round 4, measured and removed: a 128 x 128-tile instantiation -- four waves of 64 x 64, single-buffered 7-group patch, 76 KiB of LDS, <= 256 registers -- so that TWO
// workgroups share a CU and fill each other's epilogue / prologue / barrier gaps: conv4_1 forward -8 %, conv4_2 -3..-4 %, conv6 forward -6 % in isolation, +10 % on
// short launches, and 0.0 % on the step in the same-process A/B: profiles/r04t_*.)
// F32OUT (round 4, the x3 engine): the f32 accumulators leave as f32 rows of a.ws [M][ldy] (+ bias, ReLU) straight from the registers, 16 bytes per lane and
// accumulator group -- the bf16 image in LDS and everything the bf16 epilogue does (mask, accumulate, pool) do not exist in this form.
template <int NPP, bool DBUF, int G1, int G2, int WP = 2, int PI = 2, int QI = 4, bool POOL = false, bool F32OUT = false>
__global__ void __launch_bounds__(256) conv_gather_v6_kernel(const GatherArgs a) {
    constexpr int PT = WP * PI * 32, QT = (4 / WP) * QI * 32, NTHR = 256;
    static_assert(PT == 128 || PT == 64, "the filter slab is 128 (or 64: Cout <= 64) rows");
    constexpr int NP = PT / 32;                          // filter DMA pieces per wave and slab
    constexpr int PROWS = NPP * 32;                      // patch rows (416 | 576)
    constexpr int PATCH = PROWS * 128, WST = PT * 128;
    constexpr int WBASE = (DBUF ? 2 : 1) * PATCH;        // the three filter stages follow the patch buffer(s)
    constexpr int ZOFF = WBASE + 3 * WST;
    __shared__ __attribute__((aligned(16))) char smem[ZOFF + 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wq = wave / WP;
    // chunk-range split-K (round 5, F32OUT instantiations only: GatherArgs::cs_split): `cs_split` consecutive blocks share a tile
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = (F32OUT && a.cs_split > 1) ? a.cs_split : 1;
    const int part = nsplit > 1 ? lin % nsplit : 0;
    const int vb = nsplit > 1 ? lin / nsplit : lin;
    const int tq = vb / a.tiles_p, tp = vb - tq * a.tiles_p;
    const int p0 = tp * PT;
    // POOL (round 4): a tile = pool_rpt whole rows of ONE image (<= QT pixels; the rest of the tile is computed and dropped), so that the epilogue can pool them
    int q0 = tq * QT, pn = 0, ph0 = 0, rows_here = 0;
    if (POOL) {
        pn = (int)fdiv((unsigned)tq, a.div_ptpi);
        const int ti = tq - pn * a.pool_tpi;
        q0 = pn * a.H * a.W + ti * a.pool_rpt * a.W;
        rows_here = a.H - ti * a.pool_rpt < a.pool_rpt ? a.H - ti * a.pool_rpt : a.pool_rpt;
        ph0 = (ti * a.pool_rpt) >> 1;
    }
    const int ncs = (a.C + 63) >> 6;                     // 64-channel chunks (9 tap slabs each); round 5: the last one may be partial (C % 8 == 0)
    const int cs_begin = nsplit > 1 ? part * ncs / nsplit : 0, cs_end = nsplit > 1 ? (part + 1) * ncs / nsplit : ncs;     // this block's chunks
    const int x3nc = F32OUT ? (a.x3c >> 6) : 0;          // x3 engine: chunks per split part of the pixel operand (0 = plain; the bf16 instantiations compile it away)
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rw = make_rsrc(a.w, a.w_bytes);
    if (tid < 16) reinterpret_cast<uint4*>(smem + ZOFF)[tid] = make_uint4(0u, 0u, 0u, 0u);      // two zero rows
    // ---- filter DMA rows r0 + 32 i
    const int r0 = tid >> 3;
    const int cc = (tid & 7) ^ swz_g(r0);
    unsigned poff32[NP];
    bool pok[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = p0 + r0 + 32 * i;
        pok[i] = row < a.K;
        poff32[i] = (unsigned)(row * a.ldw * 2 + cc * 16);
    }
    // ---- patch DMA: piece q = wave + 4 i covers patch rows 8q .. 8q+7; row r holds raster pixel q0 - W - 1 + r
    const int total_px = a.N * a.H * a.W;
    // (piece i of a wave is 32 rows below piece i - 1 and 32 i does not touch (row >> 1) & 7: one base offset + i * xstep)
    unsigned xok = 0;
    const int xrow0 = wave * 8 + (lane >> 3);
    const unsigned xoff0 = (unsigned)((q0 - a.dil * (a.W + 1) + xrow0) * a.ldx * 2 + (((lane & 7) ^ ((xrow0 >> 1) & 7)) * 16));
    const unsigned xstep = (unsigned)(64 * a.ldx);
    // channels left of this lane's 16-byte column in a chunk: chunk cs is fetched iff cs * 64 < xcmax.  With C % 64 != 0 (the heads' input gradients: 104, 152
    // channels of dy) the columns past C of the last chunk are zero-filled by the range check; what the filter slab holds there (the next tap's rows) meets zeros.
    const int xcmax = a.C - (((lane & 7) ^ ((xrow0 >> 1) & 7)) << 3);
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int g = q0 - a.dil * (a.W + 1) + xrow0 + 32 * i;
        if ((unsigned)g < (unsigned)total_px) xok |= 1u << i;
    }
    // ---- fragment rows of this lane: pixel rows qrow0 + 32 j + l31, their 9-bit tap masks
    const int l31 = lane & 31, hi = lane >> 5;
    const int prow0 = wp * (PI * 32), qrow0 = wq * (QI * 32);
    unsigned qmask[QI];
    const int HW = a.H * a.W;
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int m = q0 + qrow0 + j * 32 + l31;
        unsigned mk = 0;
        if (m < a.M) {
            const int n = (int)fdiv((unsigned)m, a.div_howo), rem = m - n * HW;
            const int h = (int)fdiv((unsigned)rem, a.div_wo), w = rem - h * a.W;
            unsigned rm = 0, cm = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if ((unsigned)(h + (d - 1) * a.dil) < (unsigned)a.H) rm |= 1u << d;
                if ((unsigned)(w + (d - 1) * a.dil) < (unsigned)a.W) cm |= 1u << d;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if ((rm >> d) & 1u) mk |= cm << (3 * d);
        }
        qmask[j] = mk;
    }
    unsigned pofs[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) { const int row = prow0 + i * 32 + l31; pofs[i] = (unsigned)(row * 128) | ((unsigned)swz_g(row) << 16); }

    auto issue_w = [&](int kt, int stage) __attribute__((always_inline)) {        // filter slab kt = cs*9 + tap
        const int cs = kt / 9, tap = kt - cs * 9;
        const unsigned woff = (unsigned)((tap * a.C + cs * 64) * 2);
        const unsigned dP = smem_base + (unsigned)WBASE + (unsigned)stage * WST + wave_u * 1024u;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned addr = poff32[i] + woff;
            glds16_buf_nc(rw, pok[i] ? addr : 0xFFFFFFF0u, dP + i * 4096u);
        }
    };
    auto issue_x = [&](int i, int cs, int buf) __attribute__((always_inline)) {    // patch piece i of chunk cs
        const int xcs = (x3nc && cs >= x3nc) ? cs - x3nc : cs;           // x3: chunks of the third part re-read the first part's
        const unsigned addr = xoff0 + (unsigned)i * xstep + (unsigned)(xcs * 128);
        glds16_buf_nc(rx, (((xok >> i) & 1u) && xcs * 64 < xcmax) ? addr : 0xFFFFFFF0u, smem_base + (unsigned)buf * PATCH + (wave_u + 4u * (unsigned)i) * 1024u);
    };

#pragma unroll
    for (int i = 0; i < NPP; ++i) issue_x(i, cs_begin, 0);
    issue_w(cs_begin * 9, 0);
    issue_w(cs_begin * 9 + 1, 1);
    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // One tap slab.  Every slab issues the SAME number of pieces (filter slab kt+2: NP; patch pieces of chunk cs+1 by tap
    // position: 2,2,2,2,2,1,1,1,0 for 13 pieces), so the counted vmcnt in front of each barrier is a compile-time constant; past the end
    // of the k loop the pieces carry out-of-range offsets (zero fill into stages nobody reads again).
    auto slab = [&](auto TAPC, int kt, int cs, int st_c, int st_n) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int dr = tap / 3, ds = tap - dr * 3;
        // (double buffer: the NPP pieces of the next chunk dealt over taps 0..7, NPP = 13 -> 2,2,2,2,2,1,1,1,0)
        constexpr int NX = DBUF ? (tap < 8 ? NPP / 8 + (tap < NPP % 8 ? 1 : 0) : 0) : (tap == 3 ? G1 : (tap == 6 ? G2 - G1 : 0));
        constexpr int XBASE = DBUF ? tap * (NPP / 8) + (tap < NPP % 8 ? tap : NPP % 8) : (tap == 3 ? 0 : G1);
        constexpr int NPC = NP + NX;
        const char* sP = smem + WBASE + st_c * WST;
        const unsigned pbase = smem_base + (DBUF ? (unsigned)(((cs - cs_begin) & 1) * PATCH) : 0u);
        const unsigned zrow = smem_base + ZOFF;
        const bool more_x = cs + 1 < cs_end && (cs + 1) * 64 < xcmax, more_w = kt + 2 < 9 * cs_end;
        // byte offset of the next chunk's channels in a pixel row (x3: chunks of the third part re-read the first part's), wave-uniform, once per slab
        const unsigned xchunk_next = __builtin_amdgcn_readfirstlane((unsigned)(((x3nc && cs + 1 >= x3nc) ? cs + 1 - x3nc : cs + 1) * 128));
        // filter slab kt+2 = (chunk, tap) two positions ahead
        const int csn = tap < 7 ? cs : cs + 1;
        constexpr int tn = (tap + 2) % 9;
        const unsigned woff = (unsigned)((tn * a.C + csn * 64) * 2);
        const unsigned dW = smem_base + (unsigned)WBASE + (unsigned)st_n * WST + wave_u * 1024u;
        // fragment addressing of this tap: row + dr*W + ds, slot ^ ((row >> 1) & 7); padding lanes -> the zero row
        unsigned qa[QI], qx[QI];
        const int shift = (dr * a.W + ds) * a.dil;
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const unsigned row = (unsigned)(qrow0 + j * 32 + l31 + shift);
            const bool ok = (qmask[j] >> tap) & 1u;
            qa[j] = ok ? pbase + row * 128u : zrow + (row & 1u) * 128u;       // same banks as the real row: no new conflicts
            qx[j] = (unsigned)(hi * 16) ^ (((row >> 1) & 7u) << 4);
        }
        uint4 pf[2][PI], qf[2][QI];
        auto rd = [&](int ks, int r) __attribute__((always_inline)) {
            if (r < PI) {
                const int slot = ks * 2 + hi;
                pf[ks & 1][r] = *reinterpret_cast<const uint4*>(sP + (pofs[r] & 0xFFFFu) + (((unsigned)slot ^ (pofs[r] >> 16)) << 4));
            } else {
                const int j = r - PI;
                const unsigned ad = qa[j] + (qx[j] ^ (unsigned)(ks * 32));
                qf[ks & 1][j] = *reinterpret_cast<const uint4*>(smem + (ad - smem_base));
            }
        };
#pragma unroll
        for (int r = 0; r < PI + QI; ++r) rd(0, r);
        static_for<4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int j = 0; j < QI; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i) {
                    const int mi = j * PI + i;
                    const int slotno = ks * (PI * QI) + mi;
                    Mma<bf16_t>::run(pf[ks & 1][i], qf[ks & 1][j], acc[i][j]);
                    if (ks < 3) {
#pragma unroll
                        for (int r = 0; r < PI + QI; ++r)
                            if (r * (PI * QI) / (PI + QI) == mi) rd(ks + 1, r);
                    }
#pragma unroll
                    for (int q = 0; q < NPC; ++q)
                        if (q * (2 * PI * QI) / NPC == slotno) {          // over the FIRST half of the slab (round 3: +0.6 % over dealing them across all of it)
                            if (q < NP) {
                                const unsigned addr = poff32[q] + woff;
                                glds16_buf_nc(rw, (pok[q] && more_w) ? addr : 0xFFFFFFF0u, dW + q * 4096u);
                            } else {
                                const int xi = XBASE + q - NP;
                                const unsigned addr = xoff0 + (unsigned)xi * xstep + xchunk_next;
                                glds16_buf_nc(rx, (((xok >> xi) & 1u) && more_x) ? addr : 0xFFFFFFF0u,
                                              smem_base + (DBUF ? (unsigned)(((cs + 1 - cs_begin) & 1) * PATCH) : 0u) + (wave_u + 4u * (unsigned)xi) * 1024u);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
        });
    };
    int st_c = 0, st_n = 2, kt = cs_begin * 9;
    for (int cs = cs_begin; cs < cs_end; ++cs) {
        static_for<9>([&](auto TAPC) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAPC)::value;
            // may stay in flight: what the PREVIOUS slab issued (filter slab kt+1 and the patch pieces of its position)
            constexpr int PREV = (tap + 8) % 9;
            constexpr int NXP = DBUF ? (PREV < 8 ? NPP / 8 + (PREV < NPP % 8 ? 1 : 0) : 0) : (PREV == 3 ? G1 : (PREV == 6 ? G2 - G1 : 0));
            if (!DBUF && tap == 0 && cs > cs_begin) {
                // single patch buffer: every wave is past its last read of the old chunk (barrier), the rest of the new
                // chunk's patch (groups G2 .. NPP-1) goes out now and must land before the first tap reads it
                block_barrier();
#pragma unroll
                for (int i = G2; i < NPP; ++i) issue_x(i, cs, 0);
                wait_vmcnt<0>();
            } else {
                wait_vmcnt<NP + NXP>();
            }
            block_barrier();
            slab(TAPC, kt, cs, st_c, st_n);
            st_c = st_c == 2 ? 0 : st_c + 1;
            st_n = st_n == 2 ? 0 : st_n + 1;
            ++kt;
        });
    }
    wait_vmcnt<0>();                                    // the trailing (zero-fill) pieces must land before the image overwrites LDS
    if constexpr (F32OUT) {
        const int hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const int m = q0 + qrow0 + j * 32 + l31;
            if (m >= a.M) continue;
#pragma unroll
            for (int i = 0; i < PI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = p0 + prow0 + i * 32 + 8 * g + 4 * hi;
                    if (c >= a.ldy) continue;
                    float o[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (nsplit > 1) *reinterpret_cast<float4*>(a.ws + ((size_t)part * a.M + m) * a.ldy + c) = make_float4(o[0], o[1], o[2], o[3]);   // f32 partial tile
                    else x3_store4(a, m, c, o);
                }
        }
        return;
    }
    block_barrier();
    epilogue_bf16<PT, QT, NTHR, PI, QI, false, POOL>(a, smem, acc, p0, q0, prow0, qrow0, tid, pn, ph0, rows_here);
}

// sum of the split-K partial tiles (fixed order -> deterministic) + the epilogue of epilogue_bf16, 8 channels per thread
__global__ void __launch_bounds__(256) splitk_finish_kernel(const GatherArgs a) {
    const int cpr = a.ldy >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.M * cpr) return;
    const int m = idx / cpr, c0 = (idx - m * cpr) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const size_t stride = (size_t)a.M * a.ldy;
    const float* wp = a.ws + (size_t)m * a.ldy + c0;
    for (int p = 0; p < a.ksplit; ++p) {
        const float4 lo = *reinterpret_cast<const float4*>(wp + p * stride), h4 = *reinterpret_cast<const float4*>(wp + p * stride + 4);
        v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += h4.x; v[5] += h4.y; v[6] += h4.z; v[7] += h4.w;
    }
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (c0 + e < a.K) v[e] += a.bias[c0 + e];
    }
    if (a.relu && !a.accumulate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    uint4 o = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
    char* yp = a.y + ((size_t)m * a.ldy + c0) * 2;
    if (a.accumulate || a.mask) {
        uint4 old = make_uint4(0, 0, 0, 0), mk = make_uint4(0, 0, 0, 0);
        if (a.accumulate) old = *reinterpret_cast<const uint4*>(yp);
        if (a.mask) mk = *reinterpret_cast<const uint4*>(a.mask + ((size_t)m * a.ldmask + c0) * 2);
        post_chunk(o, a.accumulate != 0, a.relu != 0, old, a.mask != nullptr, mk);
    }
    *reinterpret_cast<uint4*>(yp) = o;
}


// Partial filter-gradient tile of one wave -> ws[split] in full 16-byte, row-contiguous stores: the MFMA accumulator layout has a
// lane own 4 consecutive K ROWS of one column, so storing it directly is one dword per lane and instruction (256 per lane for a
// 128 x 128 wave tile -- measured 3.5 % of the step slower than float atomics); 32 rows at a time go through a wave-private LDS
// patch [32][QI * 32] instead and leave as QI * 8 float4 per lane.  `lds` is free for the wave: the caller has passed a block
// barrier after the last slab and the trailing LDS-DMA has landed.
// ADD (round 5): dst is dW itself and the tile is ADDED to what it holds -- the flush of a launch with ONE pixel split, where no other workgroup touches the tile: plain
// 16-byte read-add-store rows instead of 32 K float atomics per workgroup (deterministic, and the atomics were ~16 us of a 40-us launch on DarkNet-53's 13 x 13 maps).
template <int PI, int QI, bool ADD = false>
__device__ __forceinline__ void store_partial_tile(const f32x16_v (&acc)[PI][QI], float* lds, float* __restrict__ dst /* ws[split] */,
                                                   int k0, int col0, int K, int RSC, int lane) {
    constexpr int W = QI * 32;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < PI; ++i) {
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[(8 * (e >> 2) + 4 * hi + (e & 3)) * W + j * 32 + l31] = acc[i][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: LDS executes in order, this pins the compiler
#pragma unroll
        for (int t = 0; t < W / 8; ++t) {
            const int idx = t * 64 + lane;
            const int row = idx / (W / 4), c4 = idx % (W / 4);
            const float4 v = *reinterpret_cast<const float4*>(lds + row * W + c4 * 4);
            const int k = k0 + i * 32 + row, col = col0 + c4 * 4;
            if (k < K && col < RSC) {
                float4* d = reinterpret_cast<float4*>(dst + (size_t)k * RSC + col);
                if (ADD) { const float4 o = *d; *d = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w); }
                else *d = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------
// wgrad, 8-wave / 3-stage generation: dW[k][(r,s,c)] += sum_pixels dy[p][k] * x[p(r,s)][c]
// Same operand handling as conv_wgrad_dma_kernel (conv.hip): both slabs stay [pixel][channel],
// LDS-DMA pieces of 4 pixel rows x 256 B with the chunk ^ 4*row source swizzle, fragments by
// ds_read_b64_tr_b16.  Re-tiled as 128 (k) x 256 (columns) per 512-thread workgroup -- the column
// operand is two 128-column sub-slabs -- with the counted-vmcnt ring, the early/late DMA issue
// stagger between SIMD partner waves and buffer-addressed DMA (range check = zero fill).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) conv_wgrad_v3_kernel(const WgradArgs a) {
    constexpr int PI = 2, QI = 2;
    constexpr int PKE = 64;                          // pixels per k-slab
    constexpr int OPB = PKE * 256;                   // bytes per 128-channel operand slab (16 KiB)
    constexpr int STAGE = 3 * OPB;                   // P + two Q sub-slabs
    constexpr int NST = 3;
    constexpr int NDMA = 6;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    // 1-D grid, XCD-aware order: an XCD's contiguous range of virtual ids covers whole pixel splits, so the
    // tiles that re-read the same dy / x pixel slabs (same split, different tile) share one L2
    const int ntiles = a.tiles_p * a.tiles_q;
    const int vb = (a.dbg & 512) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int split = vb / ntiles, tile = vb - split * ntiles;
    const int tq = tile / a.tiles_p, tp = tile - tq * a.tiles_p;
    const int p0 = tp * 128, q0 = tq * 256;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rdy = make_rsrc(a.dy, a.dy_bytes);

    // DMA lane role: pixel row dr of the piece, logical 16-B chunk dch (source-side swizzle)
    const int dr = lane >> 4;
    const int dch = (lane & 15) ^ (dr << 2);
    const int pch = p0 + dch * 8;
    const bool p_col_ok = pch < a.lddy;
    int dh[2], dw_[2], qc[2];
    bool q_col_ok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j0 = q0 + s * 128 + dch * 8;
        q_col_ok[s] = j0 < a.RSC;
        int qr = 0, qs = 0; qc[s] = 0;
        if (q_col_ok[s]) {
            const int rs = j0 / a.C;
            qc[s] = j0 - rs * a.C;
            qr = rs / a.S;
            qs = rs - qr * a.S;
        }
        dh[s] = qr * a.dil - a.pad_t;
        dw_[s] = qs * a.dil - a.pad_l;
    }
    const int HoWo = a.Ho * a.Wo;
    const int iters_total = (a.P + PKE - 1) / PKE;
    const int it0 = split * a.iters_per_split;
    int it1 = it0 + a.iters_per_split;
    if (it1 > iters_total) it1 = iters_total;
    if (it0 >= it1) return;

    auto issue = [&](int it, int stage) __attribute__((always_inline)) {
        const unsigned sP = smem_base + (unsigned)stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 8 * i;
            const int p = it * PKE + piece * 4 + dr;
            const bool pin = p < a.P;
            glds16_buf(rdy, (pin && p_col_ok) ? (unsigned)((p * a.lddy + pch) * 2) : 0xFFFFFFF0u, sP + (unsigned)piece * 1024u);
            const unsigned n = fdiv((unsigned)p, a.div_howo);
            const unsigned rem = (unsigned)p - n * (unsigned)HoWo;
            const unsigned ho = fdiv(rem, a.div_wo);
            const unsigned wo = rem - ho * (unsigned)a.Wo;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int hi = (int)ho * a.stride + dh[s], wi = (int)wo * a.stride + dw_[s];
                const bool ok = pin && q_col_ok[s] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                glds16_buf(rx, ok ? (unsigned)(((((int)n * a.H + hi) * a.W + wi) * a.ldx + qc[s]) * 2) : 0xFFFFFFF0u,
                           sP + (unsigned)(OPB + s * OPB) + (unsigned)piece * 1024u);
            }
        }
    };

    // transpose-read lane role (see conv_wgrad_dma_kernel): group g = lane>>4, c = lane&15
    const int g = lane >> 4, c = lane & 15;
    const int rr = c >> 2;
    unsigned pfo[PI], qfo[QI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int ch = (wp * 64 + i * 32 + 16 * (g & 1)) / 8 + ((c & 3) >> 1);
        pfo[i] = (unsigned)((2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int ch = ((wq & 1) * 64 + j * 32 + 16 * (g & 1)) / 8 + ((c & 3) >> 1);
        qfo[j] = (unsigned)(OPB + (wq >> 1) * OPB + (2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8);
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

    const int nk = it1 - it0;
    issue(it0, 0);
    if (nk > 1) issue(it0 + 1, 1);
    int st_c = 0, st_n = 2;
    const bool late = wave >= 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
        block_barrier();
        const bool do_issue = kt + 2 < nk;
        if (do_issue && !late) issue(it0 + kt + 2, st_n);
        const unsigned sS = smem_base + (unsigned)st_c * STAGE;
        uint4 pf[2][PI], qf[2][QI];
        auto ldf = [&](int ks, uint4 (&p)[PI], uint4 (&q)[QI]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const uint2 lo = lds_tr16(sS + ks * 4096u + pfo[i]);
                const uint2 hi2 = lds_tr16(sS + ks * 4096u + 1024u + pfo[i]);
                p[i] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
            }
#pragma unroll
            for (int j = 0; j < QI; ++j) {
                const uint2 lo = lds_tr16(sS + ks * 4096u + qfo[j]);
                const uint2 hi2 = lds_tr16(sS + ks * 4096u + 1024u + qfo[j]);
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
                    for (int h = 0; h < 4; ++h) bsum[i] += bf16_lo(d[h]) + bf16_hi(d[h]);
                }
            }
        }
        if (do_issue && late) issue(it0 + kt + 2, st_n);
        st_c = st_c == 2 ? 0 : st_c + 1;
        st_n = st_n == 2 ? 0 : st_n + 1;
    }

    const int l31 = lane & 31, hi = lane >> 5;
    if (a.ws) {
        __syncthreads();                                   // every wave is past its last fragment read
        store_partial_tile<PI, QI>(acc, reinterpret_cast<float*>(smem) + wave * (32 * QI * 32), a.ws + (size_t)split * a.K * a.RSC,
                                   p0 + wp * 64, q0 + wq * 64, a.K, a.RSC, lane);
    } else if (a.nsplit == 1 && !(a.dbg2 & 4096)) {        // one pixel split: this workgroup owns its tile of dW (dbg2 bit 12 = atomics, A/B)
        __syncthreads();
        store_partial_tile<PI, QI, true>(acc, reinterpret_cast<float*>(smem) + wave * (32 * QI * 32), a.dw, p0 + wp * 64, q0 + wq * 64, a.K, a.RSC, lane);
    } else {
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const int col = q0 + wq * 64 + j * 32 + l31;
            if (col >= a.RSC) continue;
#pragma unroll
            for (int i = 0; i < PI; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = p0 + wp * 64 + i * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                    if (k < a.K) atomicAdd(a.dw + (size_t)k * a.RSC + col, acc[i][j][e]);
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
                if (a.bws) a.bws[(size_t)split * a.K + k] = t;       // one bias slot per pixel split
                else atomicAdd(a.dbias + k, t);
            }
        }
    }
}

// Sum of the per-split partial filter gradients in split order (deterministic), added into dw; the last blocks do the same
// for the bias slots.  n4 = K * RSC / 4 (RSC is a multiple of 8).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, long long n4, float* __restrict__ dw,
                                                           const float* __restrict__ bws, int nbslot, int K, float* __restrict__ dbias,
                                                           int wblocks) {
    if ((int)blockIdx.x >= wblocks) {
        // bias slots (round 6): 16 lanes per output -- lane group g sums the slots g, g + 16, ... in order, the 16 group sums meet in LDS and are added in group
        // order (a fixed order).  One thread per output walked up to splits x tiles_q x 2 = 504 slots (conv3_2) as a chain of dependent-latency loads in ONE
        // workgroup: 66-125 us per launch, the whole deficit of the deterministic mode on the 256 x 256-tile layers (rocprof: profiles/r06d_det_wgrad_trace.md).
        constexpr int G = 16, OPB = 256 / G;
        __shared__ float smb[256];
        const int tid = threadIdx.x, g = tid / OPB, o = tid % OPB;
        const int k = ((int)blockIdx.x - wblocks) * OPB + o;
        float t = 0.f;
        if (k < K)
            for (int s = g; s < nbslot; s += G) t += bws[(size_t)s * K + k];
        smb[tid] = t;
        __syncthreads();
        if (g == 0 && k < K) {
            float r = smb[o];
#pragma unroll
            for (int q = 1; q < G; ++q) r += smb[q * OPB + o];
            dbias[k] += r;
        }
        return;
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {                      // eight 16-byte loads in flight per lane; the adds stay in split order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w4[(long long)(s + u) * n4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
    for (; s + 4 <= nsplit; s += 4) {
        const float4 v0 = w4[(long long)s * n4 + i], v1 = w4[(long long)(s + 1) * n4 + i];
        const float4 v2 = w4[(long long)(s + 2) * n4 + i], v3 = w4[(long long)(s + 3) * n4 + i];
        t.x += v0.x; t.y += v0.y; t.z += v0.z; t.w += v0.w;
        t.x += v1.x; t.y += v1.y; t.z += v1.z; t.w += v1.w;
        t.x += v2.x; t.y += v2.y; t.z += v2.z; t.w += v2.w;
        t.x += v3.x; t.y += v3.y; t.z += v3.z; t.w += v3.w;
    }
    for (; s < nsplit; ++s) {
        const float4 v = w4[(long long)s * n4 + i];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float4* d4 = reinterpret_cast<float4*>(dw);
    float4 o = d4[i];
    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    d4[i] = o;
}

// The same for MANY partials of a SMALL gradient (round 5: the 64 -> 64 kernel's 256 per-workgroup partials of 64 x 576, the first-layer kernel's 1 024
// per-wave partials of 64 x 72): one thread per output walked a chain of nsplit dependent-latency loads in 5 .. 36 blocks (64 + 256 us on the step's critical
// path).  Here 16 lanes share an output: lane group g sums the partials g, g + 16, .. in order, the 16 group sums meet in LDS and are added in group order --
// a fixed order again, 16 x the loads in flight and 16 x the blocks.  The bias slots likewise.
__global__ void __launch_bounds__(256) wgrad_reduce_wide_kernel(const float* __restrict__ ws, int nsplit, long long n4, float* __restrict__ dw,
                                                                const float* __restrict__ bws, int nbslot, int K, float* __restrict__ dbias,
                                                                int wblocks) {
    constexpr int G = 16, OPB = 256 / G;
    __shared__ float4 sm[256];
    const int tid = threadIdx.x, g = tid / OPB, o = tid % OPB;
    if ((int)blockIdx.x >= wblocks) {
        const int k = ((int)blockIdx.x - wblocks) * OPB + o;
        float t = 0.f;
        if (k < K)
            for (int s = g; s < nbslot; s += G) t += bws[(size_t)s * K + k];
        sm[tid].x = t;
        __syncthreads();
        if (g == 0 && k < K) {
            float r = sm[o].x;
#pragma unroll
            for (int q = 1; q < G; ++q) r += sm[q * OPB + o].x;
            dbias[k] += r;
        }
        return;
    }
    const long long i = (long long)blockIdx.x * OPB + o;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4)
        for (int s = g; s < nsplit; s += G) {
            const float4 v = w4[(long long)s * n4 + i];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
    sm[tid] = t;
    __syncthreads();
    if (g == 0 && i < n4) {
        float4 r = sm[o];
#pragma unroll
        for (int q = 1; q < G; ++q) { const float4 v = sm[q * OPB + o]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        float4* d4 = reinterpret_cast<float4*>(dw);
        float4 cur = d4[i];
        cur.x += r.x; cur.y += r.y; cur.z += r.z; cur.w += r.w;
        d4[i] = cur;
    }
}

// ---------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (conv1_2 forward and its dgrad:
// 300x300 maps, 2.9 M pixels -- HBM-heavy, and a 64-wide tile starves the generic kernels).
// Persistent 4-wave workgroup per CU:
//   * the WHOLE filter (9 taps x [64 rows][128 B] = 72 KiB) is loaded into LDS once;
//   * the image is walked in 8 x 32-pixel tiles; the 10 x 34-pixel input patch (halo included, out-of-image
//     pixels zero-filled by the buffer range check) is LDS-DMA'd ONCE per tile, double buffered, and all nine
//     taps read their fragments from it at shifted positions -- 1.33x input traffic instead of 9x gathers;
//     chunk slot ^ ((pixel >> 1) & 7) keeps ds_read_b128 conflict-free for ANY start pixel;
//   * wave w owns tile rows 2w, 2w+1 (64 pixels) x all 64 output channels: 144 MFMAs per tile, fragments double
//     buffered in registers, the next patch's DMA pieces interleaved into the first taps;
//   * epilogue through the just-consumed patch buffer: full 128-B lines, 1 KiB contiguous per store instruction.
// ---------------------------------------------------------------------------------------
template <int ABL>
__global__ void __launch_bounds__(256) conv3x3_c64k64_kernel(const GatherArgs a, const int tiles_r, const int tiles_c,
                                                             const int total_tiles, const FastDiv div_tpi, const FastDiv div_tc) {
    constexpr int PW = 34, PPX = 340, NPIECE = 43, PBUF = NPIECE * 1024;
    constexpr int WBYTES = 9 * 8192;
    __shared__ __attribute__((aligned(16))) char smem[WBYTES + 2 * PBUF];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rw = make_rsrc(a.w, a.w_bytes);
    const int grid = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, grid);
    const int my_tiles = slot < total_tiles ? (total_tiles - slot + grid - 1) / grid : 0;
    if (my_tiles == 0) return;

    // the filter: 72 pieces of 8 rows x 128 B
    for (int q = wave; q < 72; q += 4) {
        const int tap = q >> 3, row = (q & 7) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        glds16_buf(rw, (unsigned)(row * a.ldw * 2 + tap * 128 + lc * 16), smem_base + (unsigned)q * 1024u);
    }
    float4 bias[8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias[i * 4 + g] = a.bias ? *reinterpret_cast<const float4*>(a.bias + i * 32 + 8 * g + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int tiles_per_img = tiles_r * tiles_c;
    auto decode = [&](int v, int& n, int& h0, int& w0) __attribute__((always_inline)) {       // (round 3: no integer division in the tile loop)
        if (a.rev) v = total_tiles - 1 - v;
        n = (int)fdiv((unsigned)v, div_tpi);
        const int rem = v - n * tiles_per_img;
        const int tr = (int)fdiv((unsigned)rem, div_tc);
        h0 = tr * 8; w0 = (rem - tr * tiles_c) * 32;
    };
    // patch pieces of this wave: q = wave + 4 t, t = 0..10 (43 pieces of 8 pixels; wave 3 has ten real ones).  Everything that depends on the lane only is
    // hoisted -- patch row / column of the piece's pixel and its source offset relative to the tile origin -- so a piece costs ~8 VALU and no branch.
    int x_rel[11], x_prc[11];
#pragma unroll
    for (int t = 0; t < 11; ++t) {
        const int q = wave + 4 * t;
        const int px = q * 8 + (lane >> 3);
        const int pr = px / PW, pc = px - pr * PW;
        const int lc = (lane & 7) ^ ((px >> 1) & 7);
        x_rel[t] = ((pr * a.W + pc) * 64 + lc * 8) * 2;
        x_prc[t] = ((px < PPX && q < NPIECE) ? pr : 0x7FFF) | (pc << 16);
    }
    // perf experiments only (odtk_debug_set key 2, separate instantiations; results are garbage): bit 0 no patch DMA after the first tile, bit 10 no
    // fragment reads / MFMAs, bit 6 no epilogue
    constexpr bool abl_dma = (ABL & 1) != 0, abl_mma = (ABL & 2) != 0, abl_epi = (ABL & 4) != 0;
    auto issue_piece = [&](int n, int h0, int w0, int t, int buf, bool en) __attribute__((always_inline)) {
        const int q = wave + 4 * t;
        const int base = (((n * a.H + h0 - 1) * a.W + w0 - 1) * 64) * 2;
        const int h = (x_prc[t] & 0xFFFF) + h0 - 1, w = (x_prc[t] >> 16) + w0 - 1;
        const bool ok = en && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        if (t < 10 || wave < 3)                                    // wave-uniform (only wave 3, t = 10, has no piece: q = 43)
            glds16_buf(rx, ok ? (unsigned)(base + x_rel[t]) : 0xFFFFFFF0u, smem_base + (unsigned)(WBYTES + buf * PBUF) + (unsigned)q * 1024u);
    };
    int tn, th0, tw0;
    decode(slot, tn, th0, tw0);
#pragma unroll
    for (int t = 0; t < 11; ++t) issue_piece(tn, th0, tw0, t, 0, true);
    // fragment addressing: weights row k = i*32 + l31; patch pixel of (tile row 2w+j, column l31).
    // Physical 16-byte slot = (2 ks + hi) ^ s = (2 ks) ^ (hi ^ s) with s = (row >> 1) & 7: the lane-dependent part u = row * 128 + ((hi ^ s) << 4) is computed once
    // per (operand row | tap and tile row); the four k sub-steps then cost one v_xor each: address = (u & ~127) + ((u & 127) ^ (ks << 5)) = u ^ (ks << 5)
    // (the XOR only touches bits 5 and 6, which belong to the slot part of u).
    const unsigned wu0 = (unsigned)(l31 * 128) + ((unsigned)(hi ^ ((l31 >> 1) & 7)) << 4);
    const unsigned wu1 = (unsigned)((32 + l31) * 128) + ((unsigned)(hi ^ (((32 + l31) >> 1) & 7)) << 4);
    const int pxb0 = (2 * wave) * PW + l31, pxb1 = (2 * wave + 1) * PW + l31;
    unsigned pu[9][2];                                      // per tap and tile row: u of the patch pixel this lane reads
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int px0 = pxb0 + (tap / 3) * PW + (tap % 3), px1 = pxb1 + (tap / 3) * PW + (tap % 3);
        pu[tap][0] = (unsigned)(px0 * 128) + ((unsigned)(hi ^ ((px0 >> 1) & 7)) << 4);
        pu[tap][1] = (unsigned)(px1 * 128) + ((unsigned)(hi ^ ((px1 >> 1) & 7)) << 4);
    }
    char* stg = smem + WBYTES + wave * 8192;               // + buf * PBUF

    for (int it = 0; it < my_tiles; ++it) {
        const int buf = it & 1;
        const bool has_next = it + 1 < my_tiles;
        int nn, nh0, nw0;
        decode(has_next ? slot + (it + 1) * grid : slot, nn, nh0, nw0);
        if (it == 0) { wait_vmcnt<0>(); }
        block_barrier();            // patch `buf` (and at it = 0 the filter) visible; everybody left epilogue it-1
        f32x16_v acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const char* patch = smem + WBYTES + buf * PBUF;
        int cn, ch0, cw0;
        decode(slot + it * grid, cn, ch0, cw0);
        uint4 mk[8];                                     // ReLU-mask chunks of this tile, prefetched under the last taps
#pragma unroll
        for (int r = 0; r < 8; ++r) mk[r] = make_uint4(0, 0, 0, 0);
        // fragments run TWO k-steps ahead of the MFMAs in three register sets, and the order is pinned (sched_barrier): left alone, the scheduler sinks the
        // ds_reads next to their MFMAs and every step eats an LDS round trip (round 3: the loop ran at 45 % of the MFMA rate)
        uint4 pf[3][2], qf[3][2];
        auto ldf = [&](int step, uint4 (&p)[2], uint4 (&q)[2]) __attribute__((always_inline)) {
            const int tap = step >> 2, ks = step & 3;
            const char* wt = smem + tap * 8192;
            p[0] = *reinterpret_cast<const uint4*>(wt + (wu0 ^ (unsigned)(ks << 5)));
            q[0] = *reinterpret_cast<const uint4*>(patch + (pu[tap][0] ^ (unsigned)(ks << 5)));
            q[1] = *reinterpret_cast<const uint4*>(patch + (pu[tap][1] ^ (unsigned)(ks << 5)));
            p[1] = *reinterpret_cast<const uint4*>(wt + (wu1 ^ (unsigned)(ks << 5)));
        };
        if (!abl_mma) { ldf(0, pf[0], qf[0]); ldf(1, pf[1], qf[1]); }
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int cur = step % 3;
            if (step < 34 && !abl_mma) ldf(step + 2, pf[(step + 2) % 3], qf[(step + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (!abl_mma) Mma<bf16_t>::run(pf[cur][0], qf[cur][0], acc[0][0]);
            if ((step & 3) == 0 && 2 * (step >> 2) < 11) {
                // the next tile's patch: 11 pieces per wave, two per tap over the first taps (last tile: out-of-range offsets = zero fill of the idle buffer)
                issue_piece(nn, nh0, nw0, 2 * (step >> 2), buf ^ 1, has_next && !abl_dma);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!abl_mma) Mma<bf16_t>::run(pf[cur][0], qf[cur][1], acc[0][1]);
            if ((step & 3) == 0 && 2 * (step >> 2) + 1 < 11) {
                issue_piece(nn, nh0, nw0, 2 * (step >> 2) + 1, buf ^ 1, has_next && !abl_dma);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (step == 24 && a.mask_bits) {            // ReLU mask as sign bits: one byte per chunk instead of its 16 bytes (round 4)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int idx = r * 64 + lane;
                    const int pxl = idx >> 3, ch = idx & 7;
                    const int h = ch0 + 2 * wave + (pxl >> 5), w = cw0 + (pxl & 31);
                    if (h < a.H && w < a.W) mk[r].x = a.mask_bits[((size_t)(cn * a.H + h) * a.W + w) * 8 + ch];
                }
            } else if (step == 24 && a.mask) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int idx = r * 64 + lane;
                    const int pxl = idx >> 3, ch = idx & 7;
                    const int h = ch0 + 2 * wave + (pxl >> 5), w = cw0 + (pxl & 31);
                    if (h < a.H && w < a.W)
                        mk[r] = *reinterpret_cast<const uint4*>(a.mask + (((size_t)(cn * a.H + h) * a.W + w) * a.ldmask + ch * 8) * 2);
                }
            }
            if (!abl_mma) {
                Mma<bf16_t>::run(pf[cur][1], qf[cur][0], acc[1][0]);
                Mma<bf16_t>::run(pf[cur][1], qf[cur][1], acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vmcnt<0>();            // my pieces of the next patch landed (they had >= 3 taps of MFMA time)
        block_barrier();            // everybody is done reading patch `buf`: it becomes the output staging area
        if (abl_epi) {               // keep the accumulators alive
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
            continue;
        }
        // ---- epilogue
        char* sg = stg + buf * PBUF;
        typedef short s16x2_v __attribute__((ext_vector_type(2)));
        typedef unsigned short u16x2_v __attribute__((ext_vector_type(2)));
        const s16x2_v relu_floor = a.relu != 0 ? (s16x2_v)(0) : (s16x2_v)(-32768);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pxl = j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * hi;
                    const float4 b = bias[i * 4 + g];
                    const float v0 = acc[i][j][4 * g] + b.x, v1 = acc[i][j][4 * g + 1] + b.y;
                    const float v2 = acc[i][j][4 * g + 2] + b.z, v3 = acc[i][j][4 * g + 3] + b.w;
                    // ReLU on the ROUNDED pair as a signed 16-bit max against 0 (against -32768 = identity when there is no ReLU): rounding keeps the sign,
                    // so this equals rounding max(v, 0) -- two packed ops instead of four, and no -0 ever reaches the integer compares of the pooling below
                    uint2 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_v, cvt_pk_bf16(v0, v1)), relu_floor));
                    o.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_v, cvt_pk_bf16(v2, v3)), relu_floor));
                    *reinterpret_cast<uint2*>(sg + pxl * 128 + ((((cl >> 3) ^ pxl) & 7) << 4) + ((cl & 4) << 1)) = o;
                }
        }
        asm volatile("" ::: "memory");          // wave-private patch, in-order LDS: only pins the compiler (TBAA)
        if (a.pool_mode != 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int idx = r * 64 + lane;
                const int pxl = idx >> 3, ch = idx & 7;
                const int h = ch0 + 2 * wave + (pxl >> 5), w = cw0 + (pxl & 31);
                uint4 v = *reinterpret_cast<const uint4*>(sg + pxl * 128 + (((ch ^ pxl) & 7) << 4));
                if (h < a.H && w < a.W) {
                    const size_t m = (size_t)(cn * a.H + h) * a.W + w;
                    if (a.mask_bits) {
                        unsigned* u = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = keep_where_bits(u[q], mk[r].x >> (2 * q));
                    } else if (a.mask) post_chunk(v, false, false, mk[r], true, mk[r]);
                    *reinterpret_cast<uint4*>(a.y + (m * a.ldy + ch * 8) * 2) = v;
                }
            }
        }
        if (a.pool_mode) {
            // fused tf.layers.max_pooling2d(2, 2, 'same') (SSD300.py:209): this wave's image holds tile rows 2w, 2w+1 -- both rows of 16 pooling
            // windows (tile origins are even).  16 windows x 8 chunks = 128 tasks, two per lane: the four window pixels' chunks come back from
            // the image, the FIRST maximum in scan order wins (strict >, as odtk_maxpool2x2_fwd_idx records it) and its position goes to pidx.
            const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int task = r * 64 + lane;
                const int pp = task >> 3, ch = task & 7;
                const int h = ch0 + 2 * wave, w = cw0 + 2 * pp;
                if (h < a.H && w < a.W) {
                    uint4 v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int pxl = (t >> 1) * 32 + 2 * pp + (t & 1);
                        v[t] = *reinterpret_cast<const uint4*>(sg + pxl * 128 + (((ch ^ pxl) & 7) << 4));
                    }
                    // The values are ReLU outputs: non-negative bf16, whose bit patterns order like unsigned 16-bit integers -- the window maximum and its
                    // position are packed 16-bit integer ops on two channels at a time (round 3; the per-channel float scan was 40 % of this epilogue).
                    // Candidates outside the image are zeroed: they can tie but never win, and the FIRST maximum in scan order is
                    //   row = (max of row 1 > max of row 0), column = (right > left) inside the winning row      (strict >: ties go to the earlier one).
                    const unsigned in1 = w + 1 < a.W ? 0xFFFFFFFFu : 0u, in2 = h + 1 < a.H ? 0xFFFFFFFFu : 0u;
                    const unsigned* u0 = reinterpret_cast<const unsigned*>(&v[0]);
                    const unsigned* u1 = reinterpret_cast<const unsigned*>(&v[1]);
                    const unsigned* u2 = reinterpret_cast<const unsigned*>(&v[2]);
                    const unsigned* u3 = reinterpret_cast<const unsigned*>(&v[3]);
                    uint4 best;
                    unsigned* ub = reinterpret_cast<unsigned*>(&best);
                    unsigned code = 0;
                    const u16x2_v one = (u16x2_v)(1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u16x2_v x0 = __builtin_bit_cast(u16x2_v, u0[q]), x1 = __builtin_bit_cast(u16x2_v, u1[q] & in1);
                        const u16x2_v x2 = __builtin_bit_cast(u16x2_v, u2[q] & in2), x3 = __builtin_bit_cast(u16x2_v, u3[q] & in1 & in2);
                        const u16x2_v m01 = __builtin_elementwise_max(x0, x1), m23 = __builtin_elementwise_max(x2, x3);
                        const u16x2_v row = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m23, m01), one);
                        const u16x2_v c01 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(x1, x0), one);
                        const u16x2_v c23 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(x3, x2), one);
                        const u16x2_v rmask = (u16x2_v)(0) - row;                                 // 0xFFFF where row 1 wins
                        const u16x2_v am = ((c23 & rmask) | (c01 & ~rmask)) | (row << 1);         // 2-bit position per channel
                        const unsigned cu = __builtin_bit_cast(unsigned, am);
                        code |= ((cu | (cu >> 14)) & 0xFu) << (4 * q);
                        ub[q] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(m01, m23));
                    }
                    const size_t mo = (size_t)(cn * Hp + (h >> 1)) * Wp + (w >> 1);
                    *reinterpret_cast<uint4*>(a.ypool + (mo * a.ldpool + ch * 8) * 2) = best;
                    a.pidx[mo * 8 + ch] = (unsigned short)code;
                }
            }
        }
        asm volatile("" ::: "memory");
    }
}


// ---------------------------------------------------------------------------------------
// First layer (conv1_1 forward): 3x3 / stride 1 / pad 1, 8 (3 real + 5 zero) input channels -> 64, bias + ReLU.
// k = 72: two MFMA-starved k-slabs for the generic kernel, which then runs at the latency of its per-tile
// prologue.  Here: persistent 4-wave workgroups (3 per CU), 8 x 32-pixel tiles, the 10 x 34 halo patch is ONE
// 16-B chunk per pixel (6 LDS-DMA pieces, double buffered), the whole filter lives in registers as MFMA
// fragments (k-step t = taps 2t | 2t+1 x 8 channels; tap 9 is a zero column), and a fragment read is 32
// consecutive pixels = 512 contiguous bytes.  10 MFMAs per 32 pixels; the kernel is bound by the 128 B / pixel
// it writes, so the epilogue goes through a wave-private LDS image and stores full 128-B rows.
// ---------------------------------------------------------------------------------------
// NI = Cout / 32: 2 = the 64-channel first layer of VGG-16 (SSD300.py:193-198), 1 (round 5) = DarkNet-53's 32-channel one (YOLOv3.py:387) -- on the generic kernel
// that layer ran at 16 TFLOP/s (149 us at 416 x 416 x 8 for 88 MB of output).
template <int NI>
__global__ void __launch_bounds__(256) conv3x3_c8k64_kernel(const GatherArgs a, const int tiles_r, const int tiles_c,
                                                            const int total_tiles, const FastDiv div_tpi, const FastDiv div_tc) {
    constexpr int PW = 34, PPX = 340, NPIECE = 6, PBUF = NPIECE * 1024;
    __shared__ __attribute__((aligned(16))) char smem[2 * PBUF + 4 * 8192];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const int grid = gridDim.x;
    const int slot = blockIdx.x;
    const int my_tiles = slot < total_tiles ? (total_tiles - slot + grid - 1) / grid : 0;
    if (my_tiles == 0) return;

    // filter fragments: channel row i*32 + l31, k-step t -> tap 2t + hi (8 channels = 16 B), tap 9 = zeros
    constexpr int RBK = NI * 64, CPR = NI * 4;              // bytes per pixel row of the output image, 16-byte chunks per row
    uint4 wf[5][NI];
#pragma unroll
    for (int tt = 0; tt < 5; ++tt)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int tap = 2 * tt + hi;
            wf[tt][i] = tap < 9 ? *reinterpret_cast<const uint4*>(a.w + ((size_t)(i * 32 + l31) * a.ldw + tap * 8) * 2) : make_uint4(0, 0, 0, 0);
        }
    float4 bias[NI * 4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias[i * 4 + g] = a.bias ? *reinterpret_cast<const float4*>(a.bias + i * 32 + 8 * g + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int tiles_per_img = tiles_r * tiles_c;
    auto decode = [&](int v, int& n, int& h0, int& w0) __attribute__((always_inline)) {
        n = (int)fdiv((unsigned)v, div_tpi);
        const int rem = v - n * tiles_per_img;
        const int tr = (int)fdiv((unsigned)rem, div_tc);
        h0 = tr * 8; w0 = (rem - tr * tiles_c) * 32;
    };
    // patch pieces: 64 pixels x 16 B; wave w issues pieces w and w + 4 (< 6)
    int p_r[2], p_c[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pp = (wave + 4 * u) * 64 + lane;
        p_r[u] = pp < PPX ? pp / PW : 0x7FFF;
        p_c[u] = pp - (pp / PW) * PW;
    }
    auto issue = [&](int n, int h0, int w0, int buf, bool en) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (wave + 4 * u >= NPIECE) continue;
            const int h = h0 - 1 + p_r[u], w = w0 - 1 + p_c[u];
            const bool ok = en && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const unsigned addr = (unsigned)(((n * a.H + h) * a.W + w) * 16);
            glds16_buf(rx, ok ? addr : 0xFFFFFFF0u, smem_base + (unsigned)(buf * PBUF) + (unsigned)(wave + 4 * u) * 1024u);
        }
    };
    int tn, th0, tw0;
    decode(slot, tn, th0, tw0);
    issue(tn, th0, tw0, 0, true);

    // fragment read offsets: pixel (tile row 2w + j, column l31), k-step t -> tap 2t + hi (t = 4: both halves read
    // tap 8; the filter's zero column discards the upper one)
    unsigned qo[5];
#pragma unroll
    for (int tt = 0; tt < 5; ++tt) {
        const int tap = (2 * tt + hi) < 9 ? 2 * tt + hi : 8;
        const int dr = tap / 3, ds = tap - dr * 3;
        qo[tt] = (unsigned)((((2 * wave + dr) * PW) + l31 + ds) * 16);
    }
    char* sg = smem + 2 * PBUF + wave * 8192;

    for (int it = 0; it < my_tiles; ++it) {
        const int buf = it & 1;
        const bool has_next = it + 1 < my_tiles;
        int nn, nh0, nw0, cn, ch0, cw0;
        decode(has_next ? slot + (it + 1) * grid : slot, nn, nh0, nw0);
        decode(slot + it * grid, cn, ch0, cw0);
        wait_vmcnt<0>();
        block_barrier();            // patch `buf` visible; everybody is done reading patch buf ^ 1
        issue(nn, nh0, nw0, buf ^ 1, has_next);
        const unsigned pb = smem_base + (unsigned)(buf * PBUF);
        f32x16_v acc[NI][2];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        uint4 qf[2][5];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tt = 0; tt < 5; ++tt)
                qf[j][tt] = *reinterpret_cast<const uint4*>(smem + (pb - smem_base) + qo[tt] + j * PW * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tt = 0; tt < 5; ++tt)
#pragma unroll
                for (int i = 0; i < NI; ++i) Mma<bf16_t>::run(wf[tt][i], qf[j][tt], acc[i][j]);
        // ---- epilogue: bias + ReLU -> bf16 -> wave-private LDS image [64 pixels][128 B] -> 16 B per lane, full rows
        const bool pre_relu = a.relu != 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pxl = j * 32 + l31;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * hi;
                    const float4 b = bias[i * 4 + g];
                    float v0 = acc[i][j][4 * g] + b.x, v1 = acc[i][j][4 * g + 1] + b.y;
                    float v2 = acc[i][j][4 * g + 2] + b.z, v3 = acc[i][j][4 * g + 3] + b.w;
                    if (pre_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    uint2 o;
                    o.x = cvt_pk_bf16(v0, v1);
                    o.y = cvt_pk_bf16(v2, v3);
                    *reinterpret_cast<uint2*>(sg + pxl * RBK + ((((cl >> 3) ^ pxl) & (CPR - 1)) << 4) + ((cl & 4) << 1)) = o;
                }
        }
        asm volatile("" ::: "memory");          // wave-private image, in-order LDS: only pins the compiler (TBAA)
#pragma unroll
        for (int r = 0; r < CPR; ++r) {
            const int idx = r * 64 + lane;
            const int pxl = idx / CPR, ch = idx % CPR;
            const int h = ch0 + 2 * wave + (pxl >> 5), w = cw0 + (pxl & 31);
            const uint4 v = *reinterpret_cast<const uint4*>(sg + pxl * RBK + (((ch ^ pxl) & (CPR - 1)) << 4));
            if (h < a.H && w < a.W) {
                const size_t m = (size_t)(cn * a.H + h) * a.W + w;
                *reinterpret_cast<uint4*>(a.y + (m * a.ldy + ch * 8) * 2) = v;
                if (a.ybits) a.ybits[m * CPR + ch] = (unsigned char)pos_bits(v);       // 64 consecutive bytes per wave instruction; the kernel is bound by its 128 B / pixel
            }
        }
        asm volatile("" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------
// wgrad of the 3x3 / stride 1 / pad 1, 64 -> 64 convolution (conv1_2: 2.9 M pixels, dW is only 64 x 576):
//   dW[k][tap][c] += sum_px dy[px][k] * x[px + tap][c]
// The generic wgrad kernels gather x once per tap and waste half of their 128-row tile on K = 64.  Here a
// persistent 4-wave workgroup per CU walks 8 x 32-pixel tiles: the dy tile (256 px x 128 B) and the 10 x 34 halo
// patch of x are LDS-DMA'd ONCE per tile (double buffered, out-of-image pixels zero-filled by the range check) and
// all nine taps read the patch at shifted positions.  Both operands stay [pixel][channel]; MFMA fragments come from
// ds_read_b64_tr_b16; chunk ^ (((pixel >> 1) & 1) << 2) makes the four pixel rows of a transpose read hit distinct
// bank quarters for ANY start pixel.  The 64 x 576 result = 2 x 18 MFMA tiles: wave w keeps output channels
// 32*(w & 1).. x input channels 32*(w >> 1).. of all nine taps (144 accumulator registers) over the whole tile walk,
// 144 MFMAs per tile, one barrier per tile, and adds it to dW with float atomics once at the end.  Bias gradient fused.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wgrad3x3_c64k64_kernel(const WgradArgs a, const int tiles_r, const int tiles_c,
                                                              const int total_tiles, const FastDiv div_tpi, const FastDiv div_tc, const int npairs,
                                                              const int ncb) {
    constexpr int PW = 34, PPX = 340, NPX = 44, NPD = 32;          // patch (43 + 1 pad -> 11 per wave) / dy-tile DMA pieces (8 px x 128 B)
    constexpr int DBUF = NPD * 1024, XBUF = NPX * 1024, STAGE = DBUF + XBUF;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 1, ch_half = wave >> 1;                  // 32 output channels x 32 input channels, all 9 taps
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rdy = make_rsrc(a.dy, a.dy_bytes);
    // Round 4: any C, K that are multiples of 64 -- the workgroup owns ONE (64-channel block of x, 64-channel block of dy) pair of dW, i.e. the conv1_2
    // problem on rows of ldx / lddy elements at a channel offset; consecutive block slots are the pairs of the same tile walk, so the workgroups that
    // fetch the same dy tile / x patch run next to each other on one XCD (xcd_remap) and share it in that L2.
    const int vslot = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = vslot % npairs, slot = vslot / npairs, grid = gridDim.x / npairs;
    const int cblk = pair % ncb, kblk = pair / ncb;
    const int xp = a.ldx * 2, dp = a.lddy * 2;                     // row pitches in bytes
    const int my_tiles = slot < total_tiles ? (total_tiles - slot + grid - 1) / grid : 0;
    if (my_tiles == 0) return;
    const int tiles_per_img = tiles_r * tiles_c;
    auto decode = [&](int v, int& n, int& h0, int& w0) __attribute__((always_inline)) {
        if (a.rev) v = total_tiles - 1 - v;
        n = (int)fdiv((unsigned)v, div_tpi);
        const int rem = v - n * tiles_per_img;
        const int tr = (int)fdiv((unsigned)rem, div_tc);
        h0 = tr * 8; w0 = (rem - tr * tiles_c) * 32;
    };
    // ---- LDS-DMA pieces.  Wave w issues 19 per tile: step s < 8 -> dy tile row s, columns 8w..8w+7; step 8 + t ->
    // patch piece j = w + 4t (patch pixels 8j..8j+7, row-major over the 10 x 34 patch).  Everything that depends
    // on the lane only is hoisted (source offset relative to the tile origin, patch row / column); per piece the
    // loop then spends ~8 VALU on the image-border test and a select -- no branches, so the tile body is ONE
    // scheduling region and the DMA issue can be placed between MFMAs.
    const int sub = lane >> 3;
    const int lc16 = ((lane & 7) ^ (((sub >> 1) & 1) << 2)) * 16;
    const int dy_col = wave * 8 + sub;
    const int dy_rel = dy_col * dp + kblk * 128 + lc16;
    int x_rel[11], x_prc[11];
    static_for<11>([&](auto TT) __attribute__((always_inline)) {
        constexpr int tt = decltype(TT)::value;
        const int px = (wave + 4 * tt) * 8 + sub;
        const int pr = px / PW, pc = px - pr * PW;
        x_rel[tt] = (pr * a.W + pc) * xp + cblk * 128 + lc16;
        x_prc[tt] = (px < PPX ? pr : 0x7FFF) | (pc << 16);
    });
    auto dma_off = [&](int s, int n, int h0, int w0, bool en) __attribute__((always_inline)) -> unsigned {
        if (s < 8) {
            const int base = ((n * a.H + h0 + s) * a.W + w0) * dp;
            const bool ok = en && (h0 + s < a.H) && (w0 + dy_col < a.W);
            const unsigned addr = (unsigned)(base + dy_rel);
            return ok ? addr : 0xFFFFFFF0u;
        } else {
            const int tt = s - 8;
            const int base = ((n * a.H + h0 - 1) * a.W + w0 - 1) * xp;
            const int h = (x_prc[tt] & 0xFFFF) + h0 - 1, w = (x_prc[tt] >> 16) + w0 - 1;
            const bool ok = en && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const unsigned addr = (unsigned)(base + x_rel[tt]);
            return ok ? addr : 0xFFFFFFF0u;
        }
    };
    auto dma_issue = [&](int s, unsigned voff, int buf) __attribute__((always_inline)) {
        const unsigned dst = smem_base + (unsigned)(buf * STAGE);
        if (s < 8) glds16_buf_nc(rdy, voff, dst + (unsigned)(wave + 4 * s) * 1024u);
        else glds16_buf_nc(rx, voff, dst + (unsigned)DBUF + (unsigned)(wave + 4 * (s - 8)) * 1024u);
    };
    int tn, th0, tw0;
    decode(slot, tn, th0, tw0);
    static_for<19>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        dma_issue(s, dma_off(s, tn, th0, tw0, true), 0);
    });

    // transpose-read lane role: group g = lane >> 4 -> channel sub-block 16*(g&1), k half g >> 1 (= hi);
    // lane c = lane & 15 addresses pixel (c >> 2) of a 4-pixel group and the 8-byte quarter (c & 3) of a 32-byte span
    const int g = lane >> 4, c16 = lane & 15;
    const int rr = c16 >> 2;
    const int sub8 = (16 * (g & 1)) * 2 + (c16 & 3) * 8;           // byte offset of this lane's 4 channels inside a 64-B span
    // Per-lane LDS byte offsets, hoisted out of the loop: a fragment read addresses pixel (constant + 8*hi + rr) and
    // the swizzle bit ((pixel >> 1) & 1) only depends on (constant & 3) + rr, so four offsets cover every patch
    // position (one covers the dy tile, whose constants are multiples of 4); the constant * 128 bytes fold into the
    // ds_read offset field -> no address arithmetic inside the tile loop.
    const int lp = 8 * hi + rr;
    auto lane_off = [&](int cb32, int m) __attribute__((always_inline)) {
        const int byte = cb32 * 2 + sub8;
        return lp * 128 + (((byte >> 4) ^ ((((m + rr) >> 1) & 1) << 2)) << 4) + (byte & 8);
    };
    const unsigned poff = (unsigned)lane_off(kb * 32, 0);
    unsigned qoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) qoff[m] = (unsigned)lane_off(ch_half * 32, m);

    f32x16_v acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const bool do_bias = a.dbias != nullptr && ch_half == 0 && cblk == 0;
    float bsum = 0.f;

    for (int it = 0; it < my_tiles; ++it) {
        const int buf = it & 1;
        const bool has_next = it + 1 < my_tiles;
        int nn, nh0, nw0;
        decode(has_next ? slot + (it + 1) * grid : slot, nn, nh0, nw0);
        wait_vmcnt<0>();            // my pieces of tile `it` landed
        block_barrier();            // ... everybody's; everybody is done with tile it-1 (buffer buf ^ 1 is free)
        const unsigned sD = smem_base + (unsigned)(buf * STAGE), sX = sD + DBUF;
        const unsigned pbase = sD + poff;
        unsigned qbase[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) qbase[m] = sX + qoff[m];
        // A patch row rho serves the taps dr = 0..2 of the tile rows rho - dr, so its three (ds) fragments are read
        // once and meet a rolling window of dy-row fragments: 76 fragment reads (152 ds_read_b64_tr_b16) per 144
        // MFMAs.  Step s = half * 10 + rho (half = 16-pixel half of the 32-pixel tile row).  One wave per SIMD: an
        // MFMA only overlaps what this wave issues in its 32-cycle shadow, so the reads of step s + 1, the DMA
        // piece of step s and the bias adds are dealt out BETWEEN the MFMAs by hand (sched_barrier pins the order).
        uint2 plo[4], phi[4], qlo[2][3], qhi[2][3];
        // read #i of the fragment set of step s: 0..1 = dy row (needed first), 2..7 = patch (ds = (i-2) >> 1); odd i = pixels +4
        auto rd = [&](int s, int i) __attribute__((always_inline)) {
            const int half = s / 10, rho = s - half * 10;
            if (i >= 2) {
                const int ds = (i - 2) >> 1;
                const unsigned cpx = (unsigned)(rho * PW + half * 16 + ds);
                const uint2 v = lds_tr16(qbase[cpx & 3u] + (cpx + ((i & 1) ? 4u : 0u)) * 128u);
                if (i & 1) qhi[s & 1][ds] = v; else qlo[s & 1][ds] = v;
            } else if (rho < 8) {
                const unsigned cpx = (unsigned)(rho * 32 + half * 16);
                const uint2 v = lds_tr16(pbase + (cpx + ((i & 1) ? 4u : 0u)) * 128u);
                if (i & 1) phi[rho & 3] = v; else plo[rho & 3] = v;
            }
        };
        static_for<8>([&](auto I) __attribute__((always_inline)) { rd(0, decltype(I)::value); });
        static_for<20>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            constexpr int rho = s % 10;
            const int nm = 3 * ((rho >= 2 ? 3 : rho + 1) - (rho >= 8 ? rho - 7 : 0));      // MFMAs of this step: 3, 6 or 9
            const int dr_lo = rho >= 8 ? rho - 7 : 0, dr_hi = rho >= 2 ? 2 : rho;
            unsigned voff = 0, voff2 = 0;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                const int r = rho - dr;
                if (dr < dr_lo || dr > dr_hi) continue;
#pragma unroll
                for (int ds = 0; ds < 3; ++ds) {
                    const int mi = (dr - dr_lo) * 3 + ds;
                    const uint4 pf = make_uint4(plo[r & 3].x, plo[r & 3].y, phi[r & 3].x, phi[r & 3].y);
                    const uint4 qf = make_uint4(qlo[s & 1][ds].x, qlo[s & 1][ds].y, qhi[s & 1][ds].x, qhi[s & 1][ds].y);
                    Mma<bf16_t>::run(pf, qf, acc[dr * 3 + ds]);
                    if (s < 19) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (i >= (8 * mi + nm - 1) / nm && i < (8 * (mi + 1) + nm - 1) / nm) rd(s + 1, i);
                        // next tile's 19 pieces go out in steps 0..9 (two per step) so that even the last one has half a
                        // tile of MFMAs to cover its memory latency; last tile: zero-fills the idle buffer
                        if (s < 10) {
                            if (mi == 0) voff = dma_off(2 * s, nn, nh0, nw0, has_next);
                            if (mi == 1) dma_issue(2 * s, voff, buf ^ 1);
                            if (2 * s + 1 < 19) {
                                if (mi == 1) voff2 = dma_off(2 * s + 1, nn, nh0, nw0, has_next);
                                if (mi == 2) dma_issue(2 * s + 1, voff2, buf ^ 1);
                            }
                        }
                    }
                    if (dr == 0 && ds == 2) {             // bias gradient, once per dy fragment (every wave: branch-free)
                        const unsigned d[4] = {pf.x, pf.y, pf.z, pf.w};
#pragma unroll
                        for (int h = 0; h < 4; ++h) bsum = dot2_bf16_ones(d[h], bsum);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        });
    }
    // ---- dW (+)= : rows k = kblk*64 + kb*32 + 8*(e>>2) + 4*hi + (e&3), column = tap*C + cblk*64 + ch_half*32 + l31
    // a.ws (deterministic mode, one block pair only): this workgroup's 64 x 576 partial goes to ws[slot] with plain stores and wgrad_reduce_kernel adds
    // the slots in order -- the tile walk of a slot is fixed, so the result is bit-identical from run to run (round 5; the atomics below are not)
    float* const wdst = a.ws ? a.ws + (size_t)slot * a.K * a.RSC : a.dw;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int col = j * a.C + cblk * 64 + ch_half * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kblk * 64 + kb * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
            if (a.ws) wdst[(size_t)k * a.RSC + col] = acc[j][e];
            else atomicAdd(wdst + (size_t)k * a.RSC + col, acc[j][e]);
        }
    }
    if (do_bias) {
        const float tsum = bsum + __shfl_xor(bsum, 32);           // both k halves
        if (hi == 0) {
            if (a.bws) a.bws[(size_t)slot * a.K + kblk * 64 + kb * 32 + l31] = tsum;
            else atomicAdd(a.dbias + kblk * 64 + kb * 32 + l31, tsum);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Filter gradient of the FIRST layer (conv1_1: 3x3 / stride 1 / pad 1, 8 (3 real + 5 zero) input channels -> 64; Conv2DBackpropFilter of
// SSD300.py:193-200): dW is only 64 x 72 and the MFMA work is small (6 MFMAs per 16 pixels), but dy is 128 B per pixel -- 369 MB at batch 32 --
// so the kernel is HBM-bound (415 MB with x; the first-generation kernel it replaces took 165 us = 2.5 TB/s).
// Every WAVE is an independent worker (no block barrier inside the loop): it walks strips of 2 rows x 32 columns, LDS-DMAs the strip's 64 dy pixels
// (8 pieces) and its own 4 x 34-pixel halo patch of x (one 16-byte chunk per pixel, 3 pieces) into a wave-private three-stage ring -- two strips
// in flight behind the one being consumed, counted vmcnt -- and both MFMA operands come out of LDS through ds_read_b64_tr_b16:
//   A = dy^T: 32 output channels x 16 pixels (the layout and swizzle of wgrad3x3_c64k64_kernel's dy tile);
//   B = 32 columns = 4 taps x 8 channels: a transpose read delivers [4 pixels][16 columns] per 16-lane group, and because every lane supplies its
//       own address, the two 8-byte quarters pairs of a group point at TWO DIFFERENT taps' pixels: column n = l31 is (tap 4 cb + (n >> 3), channel n & 7),
//       i.e. dW column 32 cb + n of the [K][9 * 8] filter gradient; the lanes of taps 9 .. 11 (cb = 2) read a zero area.
// 2 x 3 accumulator tiles per wave; at the end the four waves add them in LDS (ds_add_f32) and the workgroup flushes 64 x 72 + 64 float atomics.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wgrad3x3_c8k64_kernel(const WgradArgs a, const int strips_c, const int strips_per_img, const int total_strips,
                                                             const FastDiv div_spi, const FastDiv div_sc) {
    constexpr int PW = 34;
    constexpr int DYB = 8192, XB = 3072, STG = DYB + XB, NST = 3, NPIECE = 11;
    constexpr int ZB = 1280, WAVE_LDS = NST * STG + ZB;
    __shared__ __attribute__((aligned(16))) char smem[4 * WAVE_LDS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem)) + (unsigned)(wave * WAVE_LDS);
    const unsigned zbase = wbase + NST * STG;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rdy = make_rsrc(a.dy, a.dy_bytes);
    // zero area (the taps 9 .. 11 of the third column block)
    for (int i = lane; i < ZB / 16; i += 64) *reinterpret_cast<uint4*>(smem + wave * WAVE_LDS + NST * STG + i * 16) = make_uint4(0u, 0u, 0u, 0u);

    const int nworkers = gridDim.x * 4;
    const int worker = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const int my = worker < total_strips ? (total_strips - worker + nworkers - 1) / nworkers : 0;

    // ---- DMA lane roles.  dy piece p: pixels 8p .. 8p+7 of the strip (row p >> 2, columns 8 (p & 3) + sub), logical chunk = physical ^ swizzle
    const int sub = lane >> 3;
    const int lc16 = ((lane & 7) ^ (((sub >> 1) & 1) << 2)) * 16;
    const int dy_lane = sub * 128 + lc16;
    // x piece q: patch pixels 64 q + lane (row-major over the 4 x 34 patch), one 16-byte chunk each
    int xr[3], xc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int px = 64 * q + lane;
        xr[q] = px < 4 * PW ? px / PW : 0x7FFF;
        xc[q] = px % PW;
    }
    auto issue = [&](int v, int stage, bool en) __attribute__((always_inline)) {
        // strip v -> image n, row pair, column block
        const unsigned vv = (unsigned)(en ? (a.rev ? total_strips - 1 - v : v) : 0);
        const int n = (int)fdiv(vv, div_spi);
        const int rem = (int)vv - n * strips_per_img;
        const int rp = (int)fdiv((unsigned)rem, div_sc);
        const int h0 = rp * 2, w0 = (rem - rp * strips_c) * 32;
        const unsigned dst = wbase + (unsigned)(stage * STG);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int h = h0 + (p >> 2), w = w0 + 8 * (p & 3) + sub;
            const bool ok = en && h < a.H && w < a.W;
            const unsigned off = (unsigned)(((n * a.H + h) * a.W + w0 + 8 * (p & 3)) * 128 + dy_lane);
            glds16_buf(rdy, ok ? off : 0xFFFFFFF0u, dst + (unsigned)p * 1024u);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int h = h0 - 1 + xr[q], w = w0 - 1 + xc[q];
            const bool ok = en && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const unsigned off = (unsigned)(((n * a.H + h) * a.W + w) * 16);
            glds16_buf(rx, ok ? off : 0xFFFFFFF0u, dst + (unsigned)DYB + (unsigned)q * 1024u);
        }
    };

    // ---- fragment lane roles (transpose reads): group g = lane >> 4, rr = pixel of the 4-pixel group, qq = 8-byte quarter
    const int g = lane >> 4, c16 = lane & 15, rr = c16 >> 2, qq = c16 & 3;
    const int lp = 8 * hi + rr;                                  // this lane's pixel inside a 16-pixel k-step
    // A (dy^T), k block kb: channels kb*32 + 16 (g&1) + 4 qq .. : chunk = kb*4 + 2 (g&1) + (qq >> 1), byte 8 (qq & 1); swizzle by ((pixel >> 1) & 1)
    unsigned aoff[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int chunk = kb * 4 + 2 * (g & 1) + (qq >> 1);
        aoff[kb] = (unsigned)(lp * 128 + ((chunk ^ (((lp >> 1) & 1) << 2)) << 4) + 8 * (qq & 1));     // (16 s + 4 does not change (pixel >> 1) & 1)
    }
    // B (x), column block cb: tap = 4 cb + 2 (g&1) + (qq >> 1); patch pixel of strip pixel kp: ((kp >> 5) + dr) * 34 + (kp & 31) + ds
    unsigned boff[3];
    bool bzero[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int tap = 4 * cb + 2 * (g & 1) + (qq >> 1);
        const int dr = tap / 3, ds = tap - dr * 3;
        bzero[cb] = tap >= 9;
        boff[cb] = bzero[cb] ? (unsigned)(lp * 16 + 8 * (qq & 1)) : (unsigned)((dr * PW + ds + lp) * 16 + 8 * (qq & 1));
    }

    f32x16_v acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum[2] = {0.f, 0.f};

    // prologue: strips 0 and 1 (every issue is exactly NPIECE pieces, also past the end: counted vmcnt)
    issue(worker, 0, my > 0);
    issue(worker + nworkers, 1, my > 1);
    int st = 0;
    for (int it = 0; it < my; ++it) {
        const int stn = st >= 1 ? st - 1 : NST - 1;              // (st + 2) % 3: the stage consumed in the previous iteration
        issue(worker + (it + 2) * nworkers, stn, it + 2 < my);
        wait_vmcnt<2 * NPIECE>();                                // strip `it` has landed; two newer strips may stay in flight
        const unsigned sD = wbase + (unsigned)(st * STG), sX = sD + DYB;
        unsigned ab[2], bb[3];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) ab[kb] = sD + aoff[kb];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) bb[cb] = (bzero[cb] ? zbase : sX) + boff[cb];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            // k-step s_: strip pixels 16 s_ .. 16 s_ + 15 = strip row s_ >> 1, columns 16 (s_ & 1) ..
            uint4 pf[2], qf[3];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint2 lo = lds_tr16(ab[kb] + (unsigned)(s_ * 2048)), hi4 = lds_tr16(ab[kb] + (unsigned)(s_ * 2048 + 512));
                pf[kb] = make_uint4(lo.x, lo.y, hi4.x, hi4.y);
            }
            const unsigned xo = (unsigned)(((s_ >> 1) * PW + 16 * (s_ & 1)) * 16);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const uint2 lo = lds_tr16(bb[cb] + xo), hi4 = lds_tr16(bb[cb] + xo + 64u);
                qf[cb] = make_uint4(lo.x, lo.y, hi4.x, hi4.y);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) Mma<bf16_t>::run(pf[kb], qf[cb], acc[kb][cb]);
                const unsigned dd[4] = {pf[kb].x, pf[kb].y, pf[kb].z, pf[kb].w};
#pragma unroll
                for (int h = 0; h < 4; ++h) bsum[kb] = dot2_bf16_ones(dd[h], bsum[kb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every read of this stage has returned before a later issue overwrites it
        st = st == NST - 1 ? 0 : st + 1;
    }
    wait_vmcnt<0>();
    block_barrier();
    if (a.ws) {
        // deterministic mode (round 5): every WAVE stores its own 64 x 72 partial (+ 64 bias sums) with plain stores; wgrad_reduce_kernel adds the
        // workers' partials in worker order.  (The LDS meeting below uses ds_add_f32 from four waves and global float atomics: neither has an order.)
        float* wp = a.ws + (size_t)worker * (64 * 72);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const int col = 32 * cb + l31;
                if (col < 72) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) wp[(kb * 32 + 8 * (e >> 2) + 4 * hi + (e & 3)) * 72 + col] = acc[kb][cb][e];
                }
            }
            const float t = bsum[kb] + __shfl_xor(bsum[kb], 32);
            if (a.bws && hi == 0) a.bws[(size_t)worker * 64 + kb * 32 + l31] = t;
        }
        return;
    }
    // ---- the four waves' partial sums meet in LDS (the ring is dead now), then 64 x 72 + 64 float atomics per workgroup
    float* red = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 64 * 72 + 64; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            const int col = 32 * cb + l31;
            if (col < 72) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = kb * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                    atomicAdd(red + k * 72 + col, acc[kb][cb][e]);
                }
            }
        }
        const float t = bsum[kb] + __shfl_xor(bsum[kb], 32);
        if (hi == 0) atomicAdd(red + 64 * 72 + kb * 32 + l31, t);
    }
    __syncthreads();
    for (int i = tid; i < 64 * 72; i += 256) {
        const float v = red[i];
        if (v != 0.f) atomicAdd(a.dw + i, v);
    }
    if (a.dbias != nullptr && tid < 64) {
        const float v = red[64 * 72 + tid];
        if (v != 0.f) atomicAdd(a.dbias + tid, v);
    }
}

}  // namespace

// per-device f32 scratch of the split-K path, grown on demand.  Calls are stream-ordered by the caller like everything
// else in this library.  A buffer whose address a CAPTURED HIP graph may hold is never freed or moved (the graph would replay kernels that point
// into it): growth then allocates a new, at least twice as large buffer and retires the old one for good.  A buffer that no capture has ever been
// handed (hipStreamIsCapturing at every request) is freed when it is outgrown, after a device synchronize -- round-4 advisory: the doubling
// sequence 64 -> 128 -> 256 -> 512 MB of the first ODTK_F32X3 step used to strand as much again as the arena ends up holding.
struct ConvScratchOwner { void* base = nullptr; size_t bytes = 0; bool captured = false; };
constexpr int SCRATCH_SLOTS = 4;
static ConvScratchOwner g_conv_scratch[16][SCRATCH_SLOTS];
static thread_local int g_scratch_slot = 0;            // odtk_scratch_slot(): one slot per stream the caller launches on concurrently
int set_scratch_slot(int slot) {
    if (slot < 0 || slot >= SCRATCH_SLOTS) return -1;
    g_scratch_slot = slot;
    return 0;
}
int get_scratch_slot() { return g_scratch_slot; }
// (round-5 advisory) The arenas are process-global: the bookkeeping is under a mutex, and once ANY request of this process has come from a capturing stream
// nothing is freed any more (a capture running on another stream or thread would be invalidated by the device synchronize in front of the free) -- an
// outgrown buffer is then retired as before round 5.  Two threads that launch on the same (device, slot) at the same time still share one buffer: one slot
// per concurrently used stream is the caller's contract (odtk_scratch_slot, include/odtk.h).
static std::mutex g_scratch_mutex;
static bool g_scratch_capture_seen = false;
static int scratch_get(ConvScratchOwner (&arena)[16][SCRATCH_SLOTS], size_t bytes, hipStream_t st, void** out) {
    int dev = 0;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    ODTK_REQUIRE(dev >= 0 && dev < 16, "conv: device index %d unsupported", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ConvScratchOwner& o = arena[dev][g_scratch_slot];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (capturing) g_scratch_capture_seen = true;
    if (o.bytes < bytes) {
        ODTK_REQUIRE(!capturing, "conv: library scratch must grow (%zu -> %zu bytes) inside a stream capture: run one eager step first", o.bytes, bytes);
        size_t want = o.bytes ? 2 * o.bytes : ((size_t)64 << 20);
        if (want < bytes) want = bytes;
        if (o.base && !o.captured && !g_scratch_capture_seen) {                  // nobody can still hold the old address once the device is idle
            ODTK_CHECK_HIP(hipDeviceSynchronize());
            ODTK_CHECK_HIP(hipFree(o.base));
            o.base = nullptr; o.bytes = 0;
        }
        void* p = nullptr;
        ODTK_CHECK_HIP(hipMalloc(&p, want));
        o.base = p; o.bytes = want; o.captured = false;
    }
    if (capturing) o.captured = true;
    *out = o.base;
    return ODTK_OK;
}
static int conv_scratch(size_t bytes, hipStream_t st, float** out) { return scratch_get(g_conv_scratch, bytes, st, (void**)out); }

// the x3 engine's split operands (and the column-sum partials of its bias gradient): their own arena, because the kernels they feed take split-K /
// filter-gradient partials from conv_scratch while the operands are still being read
static ConvScratchOwner g_x3_scratch[16][SCRATCH_SLOTS];
int x3_scratch(size_t bytes, hipStream_t st, char** out) { return scratch_get(g_x3_scratch, bytes, st, (void**)out); }
// small per-launch partial sums of the box-side kernels (round 6: RetinaNet's loss sums leave in workgroup order instead of by float atomics)
static ConvScratchOwner g_misc_scratch[16][SCRATCH_SLOTS];
int misc_scratch(size_t bytes, hipStream_t st, char** out) { return scratch_get(g_misc_scratch, bytes, st, (void**)out); }

static int g_num_cu = 0;
static void query_num_cu() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
}

bool gather_v3_supported(const GatherArgs& a, int dtype, int out_dtype) {
    // 32-bit byte offsets in the buffer-addressed LDS-DMA: both operands must stay below 2 GiB
    return dtype == ODTK_BF16 && out_dtype == ODTK_BF16 && (a.idiv == 1 || (a.idiv == 2 && a.dil == 1)) && a.R * a.S <= 32 && a.ldy % 8 == 0 &&
           (a.mask == nullptr || a.ldmask % 8 == 0) && a.ldx % 8 == 0 && a.C % 8 == 0 &&
           (long long)a.N * a.H * a.W * a.ldx * 2 < (1ll << 31) - (1ll << 21) && (long long)a.K * a.ldw * 2 < (1ll << 31) - (1ll << 21) &&
           (long long)a.M * a.ldy * 2 < (1ll << 31) && (a.mask == nullptr || (long long)a.M * a.ldmask * 2 < (1ll << 31));   // buffer-addressed epilogue
}

int launch_gather_v3(GatherArgs& a, hipStream_t st) {
    const int PT = a.K <= 64 ? 64 : 128;
    a.tiles_p = ceil_div(a.K, PT);
    a.tiles_q = ceil_div(a.M, 256);
    if (a.pool_mode) {                                   // the fused-pool variants (the caller checked gather_v6_pool_variant)
        const int pv = gather_v6_pool_variant(a, ODTK_BF16, ODTK_BF16);
        if (pv == 0) return ODTK_ERR_ARG;
        a.pool_rpt = pv == 1 ? 2 : 4;
        a.pool_tpi = ceil_div(a.H, a.pool_rpt);
        a.div_ptpi = make_fastdiv((unsigned)a.pool_tpi);
        a.div_wp = make_fastdiv((unsigned)((a.W + 1) / 2));
        a.tiles_q = a.N * a.pool_tpi;
        a.ksplit = -1;
        const int grid_p = a.tiles_q * a.tiles_p;
        if (pv == 1) hipLaunchKernelGGL((conv_gather_v6_kernel<20, false, 4, 9, 2, 2, 5, true>), dim3(grid_p), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_gather_v6_kernel<15, false, 2, 4, 2, 2, 5, true>), dim3(grid_p), dim3(256), 0, st, a);
        return 0;
    }
    if (g_num_cu == 0) query_num_cu();
    if (gather_v9_wanted(a, g_num_cu)) {
        // Stride-2 input gradients: the parity phases on the small-map kernel (64 x 64 tiles), or on THIS kernel (PT x 256 tiles, eight waves: a third of the LDS-DMA
        // pieces per MFMA) where its tiles cover well over half the CUs -- a phase launch streams dy (each row is read by 1-4 taps, not 9) and this kernel's
        // two-slabs-in-flight ring then runs at memory latency, ~1 us per slab, with 1, 2, 2 and 4 taps' worth of slabs per tile: measured (tools/conv_bench.py,
        // us, small-map | 8-wave): DarkNet-53 at 8 images 52 x 52 (176 tiles) 57 | 43, 104 x 104 (340) 65 | 51, 208 x 208 86 | 74, 416 x 416 251 | 197 -- but
        // 26 x 26 (96 tiles) 59 | 70 and conv8_2 of SSD300 (96 tiles) 35 | 40.  dbg2 bit 13 = always the small-map kernel (A/B).
        const bool phase = a.idiv == 2;
        const int tiles_ph = ceil_div(a.K, PT) * (ceil_div(a.M, 4 * 256) * 4);
        if (!(phase && tiles_ph >= g_num_cu / 2 + 32 && !(a.dbg2 & 8192) && !(a.dbg2 & 128))) return launch_gather_v9(a, st, g_num_cu);
        GatherArgs b = a;                                    // the phase table (tile counts for 64-pixel tiles) is rebuilt for 256-pixel tiles
        b.plan_v9_qt = 256;
        if (int e = launch_gather_v9(b, st, g_num_cu)) return e;       // fills b.v9 / b.tiles_q only (plan_v9_qt != 0: no launch)
        a = b;
        a.tiles_p = ceil_div(a.K, PT);
        a.ksplit = 1;
        const int grid_ph = a.tiles_p * a.tiles_q;
        if (PT == 64) hipLaunchKernelGGL((conv_gather_v3_kernel<64, true, false, true, true, false, false, true>), dim3(grid_ph), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((conv_gather_v3_kernel<128, true, false, true, true, false, false, true>), dim3(grid_ph), dim3(512), 0, st, a);
        return 0;
    }
    const int tiles = a.tiles_p * a.tiles_q;
    const int nk = ceil_div(a.Kdim, 64);
    // split-K: few tiles with a long k loop leave most CUs idle and run at DMA latency; give every tile
    // up to 256 / tiles blocks of >= 4 k-slabs each (dbg bit 13 turns it off for A/B runs)
    int ksplit = 1;
    const int halo = 2 * a.dil * (a.W + 1);               // patch rows beyond the 256 of the tile
    // wide maps (conv2_x, W = 150): single-buffer variant, groups 0..3 / 4..8 of the next chunk ride on taps 3 / 6 (dbg bit 26 = off, A/B)
    // (round 2: rows of 96 .. 143 pixels too -- CenterNet / FCOS 128, YOLOv3 104, YOLOv2 120 -- with their own early-refill group counts G1, G2)
    // (and rows of 160 .. 175 pixels -- conv2_x of the 320-pixel models -- on a 608-row patch)
    // (round 4: rows of 80 .. 95 pixels -- conv3_x of the 320-pixel models, 162 halo rows: two more than the double-buffered patch holds -- on a 448-row patch)
    const bool v6_wide = halo > 160 && halo <= 352 && a.dil * a.W >= 80 && !(a.dbg & (1 << 26));
    const bool v6_wide_hi = v6_wide && a.dil * a.W >= 144 && halo <= 320;   // what the 512-pixel tile variants are instantiated for
    // (round 5: C % 64 != 0 with a zero-filled last chunk where at least 70 % of the chunks' channels are real -- 104 of 128, 152 of 192; dbg2 bit 10 = off)
    const bool c_ok = a.C % 64 == 0 || (a.C > 64 && a.C % 8 == 0 && 10 * a.C >= 7 * 64 * ceil_div(a.C, 64) && !(a.dbg2 & 1024));
    const bool v6_ok = !(a.dbg & 65536) && PT == 128 && c_ok && a.R == 3 && a.S == 3 && a.ostride == 1 && a.idiv == 1 &&
                       a.pad_t == a.dil && a.pad_l == a.dil && a.H == a.Ho && a.W == a.Wo && (halo <= 160 || v6_wide) && a.Kdim == 9 * a.C;
    const int v6_min_tiles = (a.dbg >> 18) & 255;         // A/B (dbg bits 18-25): halo kernel instead of split-K from this many tiles on
    if (tiles <= 128 && nk >= 8 && !(a.dbg & 8192) && !(v6_ok && v6_min_tiles && tiles >= v6_min_tiles)) {
        ksplit = 256 / tiles;                    // (2..6 slabs per part and 512 / tiles were measured: this is the best)
        if (ksplit > nk / 4) ksplit = nk / 4;
        if (ksplit > 32) ksplit = 32;
    }
    // Cout <= 64 on the halo kernel: 64 x 512 tiles (dbg bit 27 = off, A/B)
    if (PT == 64 && ksplit < 2 && !(a.dbg & 65536) && !(a.dbg & (1 << 27)) && a.C % 64 == 0 && a.R == 3 && a.S == 3 && a.ostride == 1 && a.idiv == 1 &&
        a.pad_t == a.dil && a.pad_l == a.dil && a.H == a.Ho && a.W == a.Wo && a.Kdim == 9 * a.C &&
        ((halo <= 160 && a.dil * a.W >= 64) || v6_wide_hi) && ceil_div(a.M, 512) >= 2 * 256) {
        a.ksplit = -1;
        a.tiles_q = ceil_div(a.M, 512);
        if (halo <= 160) hipLaunchKernelGGL((conv_gather_v6_kernel<21, false, 2, 4, 1, 2, 4>), dim3(a.tiles_q), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_gather_v6_kernel<26, false, 4, 9, 1, 2, 4>), dim3(a.tiles_q), dim3(256), 0, st, a);
        return 0;
    }
    // Round 5: 3 x 3 / stride 1 layers with few tiles on the RASTER-RUN HALO kernel with a split over whole 64-channel chunks instead of the 8-wave kernel's split
    // over 64-element slabs (YOLOv3's 13 x 13 / 26 x 26 layers at 8 images, pred2 / pred3 of SSD300): one patch per chunk instead of one gathered slab per tap, four
    // waves of 2 x 4 MFMA tiles -- the halo kernel's 1 000+ TFLOP/s instead of the 8-wave kernel's ~550.  192- or 256-pixel tiles by the same cost model as the
    // unsplit launch; dbg2 bit 8 = off (A/B).
    // Round 5 (late): the same layers where 128 x 128 tiles give at least ~2/3 of a workgroup per CU -- four waves of 64 x 64, a single-buffered 224-row patch
    // (128 + 2 (W + 1) <= 224: W <= 47), 76 KiB of LDS, so TWO workgroups share a CU and cover each other's patch loads / epilogues; whole reduction per
    // workgroup: no f32 partials, no finish launch.  (The round-4 experiment r04t put this shape on the MANY-tile layers of SSD300 and gained nothing in the
    // step; here it replaces a split launch + its finish.)  dbg2 bit 14 (16384) = off (A/B).
    if (ksplit >= 2 && v6_ok && halo <= 96 && a.dil == 1 && a.C % 64 == 0 && !(a.dbg2 & 16384)) {
        const int tq128 = ceil_div(a.M, 128);
        if (tq128 * a.tiles_p >= (2 * g_num_cu) / 3) {
            a.tiles_q = tq128;
            a.ksplit = -1;
            if (a.W >= 32) hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 1, 2, 2, 2, 2>), dim3(tq128 * a.tiles_p), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 0, 0, 2, 2, 2>), dim3(tq128 * a.tiles_p), dim3(256), 0, st, a);
            return 0;
        }
        // ... and 64 (channels) x 128 (pixels) tiles where only those reach that count (DarkNet-53's 26 x 26 input gradients: dx has 256 channels; its 13 x 13
        // forward): four waves of 64 x 32, the same patch and residency, half the filter slab; dbg2 bit 15 (32768) = off (A/B)
        const int tp64 = ceil_div(a.K, 64);
        if (tq128 * tp64 >= (2 * g_num_cu) / 3 && !(a.dbg2 & 32768)) {
            a.tiles_q = tq128;
            a.tiles_p = tp64;
            a.ksplit = -1;
            if (a.W >= 32) hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 1, 2, 1, 2, 1>), dim3(tq128 * tp64), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 0, 0, 1, 2, 1>), dim3(tq128 * tp64), dim3(256), 0, st, a);
            return 0;
        }
    }
    if (ksplit >= 2 && v6_ok && halo <= 160 && ceil_div(a.C, 64) >= 2 && !(a.dbg2 & 256)) {
        const int ncs = ceil_div(a.C, 64);
        int best_qt = 0, best_split = 1;
        long long best_cost = 0;
        for (int qt = 256; qt >= 192; qt -= 64) {
            const int t = ceil_div(a.M, qt) * a.tiles_p;
            int sp = g_num_cu / t;
            if (sp > ncs) sp = ncs;
            if (sp < 1) sp = 1;
            const long long cost = (long long)ceil_div(t * sp, g_num_cu) * (ceil_div(ncs, sp) * (qt + 8) + 48);     // rounds x (chunks x pixels + prologue / epilogue)
            if (best_qt == 0 || cost < best_cost) { best_qt = qt; best_split = sp; best_cost = cost; }
        }
        if (best_split >= 2) {
            float* ws = nullptr;
            if (int e = conv_scratch((size_t)best_split * a.M * a.ldy * sizeof(float), st, &ws)) return e;
            a.ws = ws; a.cs_split = best_split;
            a.tiles_q = ceil_div(a.M, best_qt);
            const int grid = a.tiles_p * a.tiles_q * best_split;
            if (best_qt == 192) hipLaunchKernelGGL((conv_gather_v6_kernel<11, true, 0, 0, 2, 2, 3, false, true>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v6_kernel<13, true, 0, 0, 2, 2, 4, false, true>), dim3(grid), dim3(256), 0, st, a);
            a.ksplit = best_split;                        // what splitk_finish_kernel sums (same [part][M][ldy] layout)
            hipLaunchKernelGGL(splitk_finish_kernel, dim3(ceil_div(a.M * (a.ldy / 8), 256)), dim3(256), 0, st, a);
            a.ksplit = -6;                                // odtk_conv_last_kernel: "conv_gather_v6_kernel+splitk"
            return 0;
        }
    }
    if (ksplit >= 2) {
        float* ws = nullptr;
        if (int e = conv_scratch((size_t)ksplit * a.M * a.ldy * sizeof(float), st, &ws)) return e;
        a.ksplit = ksplit; a.ws = ws;
        const int grid = tiles * ksplit;
        if (PT == 64) hipLaunchKernelGGL((conv_gather_v3_kernel<64, true, false, true, false, true>), dim3(grid), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((conv_gather_v3_kernel<128, true, false, true, false, true>), dim3(grid), dim3(512), 0, st, a);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(ceil_div(a.M * (a.ldy / 8), 256)), dim3(256), 0, st, a);
        return 0;
    }
    a.ksplit = 1;
    const int grid = tiles;
    // raster-run halo kernel: 3x3 (dilated), stride 1, SAME, patch of 256 + 2 * dil * (W + 1) rows <= 416 (dbg bit 16 = off, A/B)
    if (v6_ok) {
        if (g_num_cu == 0) query_num_cu();
        a.ksplit = -1;                                   // tells the dispatcher which kernel ran (odtk_conv_last_kernel)
        // 128 x 512 tiles of four 128 x 128 wave tiles (0.5 instead of 0.75 fragment reads per MFMA) where the map is large enough to fill
        // the chip twice over with them: conv3_x (W = 75) and conv2_x (W = 150); dbg bit 27 = off (A/B)
        const int tq512 = ceil_div(a.M, 512);
        // (measured, same box: conv3_2 fwd 198 -> 180 us, dgrad 215 -> 199, conv3_1 fwd 116 -> 106, conv2_1 fwd 164 -> 157, conv2_2 fwd 228 -> 224; the
        // wide variant with a masked / accumulating epilogue -- conv2_2 dgrad -- is 2 % SLOWER: 64 more operand registers -> not used there)
        const bool post512 = a.accumulate || a.mask;
        if (!(a.dbg & (1 << 27)) && tq512 * a.tiles_p >= 2 * g_num_cu && a.dil * a.W >= 64 && (halo <= 160 || (!post512 && v6_wide_hi))) {
            a.tiles_q = tq512;
            if (halo <= 160) hipLaunchKernelGGL((conv_gather_v6_kernel<21, false, 2, 4, 1, 4, 4>), dim3(tq512 * a.tiles_p), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v6_kernel<26, false, 4, 9, 1, 4, 4>), dim3(tq512 * a.tiles_p), dim3(256), 0, st, a);
            return 0;
        }
        // 128 x 192 tiles where 256-pixel tiles leave a quarter of the CUs idle and 192-pixel tiles still fit one round (conv5_x: 184 -> 244
        // workgroups of 3/4 the work, 56 -> 50 us); dbg bit 28 = off (A/B)
        // (round 4: whenever 192-pixel tiles need less time by rounds x (pixels + ~32 of prologue / epilogue per tile) -- conv4_1's input gradient: 362 tiles of
        //  256 pixels = 2 rounds at 71 % -> 482 of 192 = 2 rounds of 3/4 the work; conv6 forward: 368 -> 488)
        const int tq192 = ceil_div(a.M, 192);
        const long long cost256 = (long long)ceil_div(tiles, g_num_cu) * (256 + 32), cost192 = (long long)ceil_div(tq192 * a.tiles_p, g_num_cu) * (192 + 32);
        if (!(a.dbg & (1 << 28)) && halo <= 160 && cost192 < cost256) {
            a.tiles_q = tq192;
            hipLaunchKernelGGL((conv_gather_v6_kernel<11, true, 0, 0, 2, 2, 3>), dim3(tq192 * a.tiles_p), dim3(256), 0, st, a);
            return 0;
        }
        if (halo <= 160) hipLaunchKernelGGL((conv_gather_v6_kernel<13, true, 0, 0>), dim3(grid), dim3(256), 0, st, a);
        else if (halo > 320) hipLaunchKernelGGL((conv_gather_v6_kernel<19, false, 5, 10>), dim3(grid), dim3(256), 0, st, a);
        else if (a.dil * a.W >= 144) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 4, 9>), dim3(grid), dim3(256), 0, st, a);
        else if (a.dil * a.W >= 128) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 4, 8>), dim3(grid), dim3(256), 0, st, a);
        else if (a.dil * a.W >= 112) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 3, 7>), dim3(grid), dim3(256), 0, st, a);
        else if (a.dil * a.W >= 96) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 3, 6>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_gather_v6_kernel<14, false, 2, 5>), dim3(grid), dim3(256), 0, st, a);
        return 0;
    }
    const bool db = (a.dbg & 32) == 0;      // fragment double buffering (default on; dbg bit 5 turns it off)
    const bool early = (a.dbg & 128) != 0;  // "landed one slab early" protocol (dbg bit 7, A/B)
    const bool buf = (a.dbg & 256) == 0;    // buffer-addressed DMA (default on; dbg bit 8 = 64-bit global addressing, A/B)
    const bool c64 = a.C % 64 == 0 && a.Kdim % 64 == 0 && (a.dbg & 4096) == 0;   // wave-uniform tap walk (dbg bit 12 = per-lane walk, A/B)
    const bool ilv = (a.dbg & 16384) != 0;  // interleaved slab body (dbg bit 14, A/B)
#define ODTK_V3(PT_) \
    do { \
        if (buf && c64 && ilv) hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, true, false, true, true, false, true>), dim3(grid), dim3(512), 0, st, a); \
        else if (buf && c64) hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, true, false, true, true>), dim3(grid), dim3(512), 0, st, a); \
        else if (buf) hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, true, false, true>), dim3(grid), dim3(512), 0, st, a); \
        else if (early) hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, true, true, false>), dim3(grid), dim3(512), 0, st, a); \
        else if (db) hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, true, false, false>), dim3(grid), dim3(512), 0, st, a); \
        else hipLaunchKernelGGL((conv_gather_v3_kernel<PT_, false, false, false>), dim3(grid), dim3(512), 0, st, a); \
    } while (0)
    if (PT == 64) ODTK_V3(64); else ODTK_V3(128);
#undef ODTK_V3
    return 0;
}

// Raster-run halo kernel with the 2x2 pool in its epilogue (odtk_conv2d_fwd_pool2x2, round 4): 3x3 / stride 1 / SAME forward layers on C % 64 == 0 whose map
// width lets whole row pairs fill a 320-pixel tile -- 1 = rows of 144..159 pixels (two rows per tile: conv2_2 + pool2 at 150 x 150), 2 = rows of 64..79 pixels
// (four rows: conv3_3 + pool3 at 75 x 75; odd sizes pool with clipped windows).  0 = not covered.
int gather_v6_pool_variant(const GatherArgs& a, int dtype, int out_dtype) {
    if (!gather_v3_supported(a, dtype, out_dtype)) return 0;
    if (!(a.C % 64 == 0 && a.R == 3 && a.S == 3 && a.ostride == 1 && a.idiv == 1 && a.dil == 1 && a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo &&
          a.Kdim == 9 * a.C && a.K % 8 == 0 && a.K > 64 && !a.accumulate && !a.mask && a.H >= 2)) return 0;
    if (a.dbg & 65536) return 0;
    if (a.W >= 144 && a.W <= 159) return 1;
    if (a.W >= 64 && a.W <= 79) return 2;
    return 0;
}

bool gather_c64_supported(const GatherArgs& a, int dtype, int out_dtype) {
    return dtype == ODTK_BF16 && out_dtype == ODTK_BF16 && a.C == 64 && a.ldx == 64 && a.K == 64 && a.R == 3 && a.S == 3 &&
           a.dil == 1 && a.ostride == 1 && a.idiv == 1 && a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo &&
           !a.accumulate && a.ldy % 8 == 0 && (a.mask == nullptr || a.ldmask % 8 == 0) && a.ldw == 576 &&
           (long long)a.N * a.H * a.W * 64 * 2 < (1ll << 31);
}

int launch_gather_c64(GatherArgs& a, hipStream_t st) {
    if (g_num_cu == 0) query_num_cu();
    if ((unsigned)a.dbg >> 31) a.rev = 0;                    // debug bit 31: every kernel walks its tiles upwards (A/B)
    const int tr = ceil_div(a.H, 8), tc = ceil_div(a.W, 32);
    const int tiles = a.N * tr * tc;
    const int grid = tiles < g_num_cu ? tiles : g_num_cu;
    const FastDiv d0 = make_fastdiv((unsigned)(tr * tc)), d1 = make_fastdiv((unsigned)tc);
    const int abl = (a.dbg & 1) | ((a.dbg & 1024) ? 2 : 0) | ((a.dbg & 64) ? 4 : 0);
#define ODTK_C64(A) case A: hipLaunchKernelGGL(conv3x3_c64k64_kernel<A>, dim3(grid), dim3(256), 0, st, a, tr, tc, tiles, d0, d1); break;
    switch (abl) {
        ODTK_C64(0) ODTK_C64(1) ODTK_C64(2) ODTK_C64(4)
        default: return 1;
    }
#undef ODTK_C64
    return 0;
}

bool gather_c8_supported(const GatherArgs& a, int dtype, int out_dtype) {
    return dtype == ODTK_BF16 && out_dtype == ODTK_BF16 && a.C == 8 && a.ldx == 8 && (a.K == 64 || (a.K == 32 && !a.ybits)) && a.ldy == a.K && a.R == 3 && a.S == 3 &&
           a.dil == 1 && a.ostride == 1 && a.idiv == 1 && a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo &&
           !a.accumulate && a.mask == nullptr && a.ldw == 72 && (long long)a.N * a.H * a.W * 16 < (1ll << 31);
}

int launch_gather_c8(GatherArgs& a, hipStream_t st) {
    if (g_num_cu == 0) query_num_cu();
    const int tr = ceil_div(a.H, 8), tc = ceil_div(a.W, 32);
    const int tiles = a.N * tr * tc;
    const int grid = tiles < 3 * g_num_cu ? tiles : 3 * g_num_cu;
    if (a.K == 32) hipLaunchKernelGGL(conv3x3_c8k64_kernel<1>, dim3(grid), dim3(256), 0, st, a, tr, tc, tiles, make_fastdiv((unsigned)(tr * tc)), make_fastdiv((unsigned)tc));
    else hipLaunchKernelGGL(conv3x3_c8k64_kernel<2>, dim3(grid), dim3(256), 0, st, a, tr, tc, tiles, make_fastdiv((unsigned)(tr * tc)), make_fastdiv((unsigned)tc));
    return 0;
}

static bool g_wgrad_deterministic = true;       // round 6: the default (odtk_debug_set key 5 = 0 switches to float atomics)
void set_wgrad_deterministic(bool on) { g_wgrad_deterministic = on; }
bool get_wgrad_deterministic() { return g_wgrad_deterministic; }


bool wgrad_c64_supported(const WgradArgs& a, int dtype) {
    if (!(dtype == ODTK_BF16 && a.C % 64 == 0 && a.K % 64 == 0 && a.C >= 64 && a.K >= 64 && a.ldx >= a.C && a.ldx % 8 == 0 && a.lddy >= a.K && a.lddy % 8 == 0 &&
          a.R == 3 && a.S == 3 && a.dil == 1 && a.stride == 1 && a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo && a.RSC == 9 * a.C &&
          (long long)a.N * a.H * a.W * a.ldx * 2 < (1ll << 31) && (long long)a.P * a.lddy * 2 < (1ll << 31)))
        return false;
    if (a.C == 64 && a.K == 64) return true;                    // conv1_2: the kernel's home
    // Round 4: one (64 x 64) block pair of dW per workgroup for the wide, large maps whose 8 x 32-pixel tiles waste little -- conv2_1 / conv2_2 at 150 x 150
    // (5 x 19 tiles per image: 94 %): 161 -> 1xx us and 273 -> 2xx us against the 8-wave gather kernel (one x slab per tap there, one halo patch per tile here).
    // Narrow maps (W = 75: 78 %, W = 38: 59 %) stay on the generic kernels; dbg bit 17 = off (A/B).
    if (a.dbg & (1 << 17)) return false;      // (deterministic mode, round 6: one partial per workgroup in its block of a dW-shaped slot buffer + the reduction launch, like 64 -> 64)
    const int npairs = (a.C / 64) * (a.K / 64);
    if (a.dbg & (1 << 15)) return npairs <= 16;                 // tests: the block-pair path on small / ragged problems too
    const double eff = (double)a.W / (32.0 * ceil_div(a.W, 32)) * (double)a.H / (8.0 * ceil_div(a.H, 8));
    const long long tiles = (long long)a.N * ceil_div(a.H, 8) * ceil_div(a.W, 32);
    return npairs <= 8 && eff >= 0.9 && tiles * npairs >= 16 * 256;
}

int launch_wgrad_c64(WgradArgs& a, hipStream_t st) {
    if (g_num_cu == 0) query_num_cu();
    a.rev = ((unsigned)a.dbg >> 31) ? 0 : 1;
    const int tr = ceil_div(a.H, 8), tc = ceil_div(a.W, 32);
    const int tiles = a.N * tr * tc;
    const int ncb = a.C / 64, npairs = ncb * (a.K / 64);
    int per_pair = g_num_cu / npairs;                           // workgroups per block pair (one workgroup per CU in all)
    if (per_pair > tiles) per_pair = tiles;
    if (per_pair < 1) per_pair = 1;
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 2);
    a.dy_bytes = (unsigned)((size_t)a.P * a.lddy * 2);
    // deterministic mode: one partial per workgroup + the fixed-order reduction launch.  Workgroup `slot` of block pair (kblk, cblk) stores its 64 x 576 partial into
    // ITS block of the dW-shaped buffer ws[slot]; the per_pair workgroups of every pair fill the per_pair buffers completely, so the reduction launch sums
    // ws[0 .. per_pair) in slot order into dW like pixel splits (round 6: conv2_x left this kernel in deterministic mode before -- 200 us of the mode's 5 %)
    if (int e = wgrad_split_scratch(a, per_pair, a.dbias ? per_pair : 0, st)) return e;
    hipLaunchKernelGGL(wgrad3x3_c64k64_kernel, dim3(per_pair * npairs), dim3(256), 0, st, a, tr, tc, tiles, make_fastdiv((unsigned)(tr * tc)),
                       make_fastdiv((unsigned)tc), npairs, ncb);
    wgrad_split_reduce(a, st);
    return 0;
}

bool wgrad_c8_supported(const WgradArgs& a, int dtype) {
    return dtype == ODTK_BF16 && a.C == 8 && a.ldx == 8 && a.K == 64 && a.lddy == 64 && a.R == 3 && a.S == 3 && a.dil == 1 && a.stride == 1 &&
           a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo && a.RSC == 72 && (long long)a.N * a.H * a.W * 128 < (1ll << 32) - 65536;
}

int launch_wgrad_c8(WgradArgs& a, hipStream_t st) {
    if (g_num_cu == 0) query_num_cu();
    a.rev = ((unsigned)a.dbg >> 31) ? 0 : 1;
    const int sc = ceil_div(a.W, 32), sr = ceil_div(a.H, 2);
    const int spi = sc * sr, total = a.N * spi;
    int grid = ceil_div(total, 4 * 8);                         // >= 8 strips per wave before a second workgroup per CU would pay its flush
    if (grid > g_num_cu) grid = g_num_cu;
    if (grid < 1) grid = 1;
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 2);
    a.dy_bytes = (unsigned)((size_t)a.P * a.lddy * 2);
    if (int e = wgrad_split_scratch(a, grid * 4, a.dbias ? grid * 4 : 0, st)) return e;          // deterministic mode: one partial per wave
    hipLaunchKernelGGL(wgrad3x3_c8k64_kernel, dim3(grid), dim3(256), 0, st, a, sc, spi, total, make_fastdiv((unsigned)spi), make_fastdiv((unsigned)sc));
    wgrad_split_reduce(a, st);
    return 0;
}

// ---------------------------------------------------------------------------------------
// Narrow f32 filter gradient (round 4): the 7..56-channel layers of RetinaNet.py:27's widths (and every other narrow f32 layer) on
// conv_wgrad_kernel<float> ran at 2-45 TFLOP/s -- each of its 32-pixel iterations is two barriers around register-staged, transposing
// loads.  v_mfma_f32_32x32x2_f32 takes ONE value per lane and operand (A: dy[pixel hi][channel l31], B: x[pixel hi][channel l31]), so both
// operands can stay in their natural [pixel][channel] rows: an 8 x 32-pixel tile of dy and its (8 + R - 1) x (32 + S - 1) halo patch of x go HBM -> LDS by
// LDS-DMA (two stages; out-of-image and pad-column chunks are zero-filled by the buffer range check), every tap reads the patch at a shifted row, all
// fragment reads are 4-byte words of consecutive lanes (conflict-free).  Persistent workgroups walk the tiles with the accumulators live -- taps x KI x CI
// tiles of 32 x 32 per wave, a wave takes a quarter of each tile's pixels -- and flush once: waves summed through LDS, then float atomics into dW
// (and the bias gradient, summed from the A fragments, into dbias).  MFMA-bound for 3x3 (28 -> 28 at 200 x 200 x 16: 248 -> ~90 us), HBM-bound for 1x1.
// Stride 1, dilation 1, SAME padding, R = S in {1, 3}, K <= 32 KI, C <= 32 CI.
// ---------------------------------------------------------------------------------------
namespace {
template <bool R3, int KI, int CI>
__global__ void __launch_bounds__(256) wgrad_f32_narrow_kernel(const WgradArgs a, int tiles_h, int tiles_w, int total_tiles) {
    constexpr int TH = R3 ? 8 : 4, TW = 32, TP = TH * TW;                 // tile: TH rows of 32 pixels
    constexpr int PH = TH + (R3 ? 2 : 0), PW = TW + (R3 ? 2 : 0), PP = PH * PW;
    constexpr int NT = R3 ? 9 : 1;
    constexpr int RBK = 128 * KI, RBC = 128 * CI;                           // LDS row bytes of a dy / x pixel
    constexpr int RPK = 8 / KI, RPC = 8 / CI;                               // pixel rows per 1-KiB DMA piece
    constexpr int NPK = TP / RPK, NPC = (PP + RPC - 1) / RPC;               // pieces per tile
    constexpr int XOFF = TP * RBK;                                          // x patch behind the dy tile
    constexpr int STAGE = XOFF + NPC * 1024;
    constexpr int NQK = (NPK + 3) / 4, NQC = (NPC + 3) / 4;                 // pieces per wave
    static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the LDS");
    static_assert(NT * KI * CI * 16 * 256 * 4 / 4 <= 2 * STAGE, "the flush image must fit");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rdy = make_rsrc(a.dy, a.dy_bytes);

    // per lane and piece: position inside the tile / patch and the byte offset relative to the tile's first pixel
    int kph[NQK], kpw[NQK], xph[NQC], xpw[NQC];
    unsigned koff[NQK], xoff[NQC];
    bool kon[NQK], xon[NQC];
#pragma unroll
    for (int i = 0; i < NQK; ++i) {
        const int q = wave + 4 * i;
        const int row = q * RPK + lane / (8 * KI), ch = lane % (8 * KI);
        kph[i] = row / TW; kpw[i] = row % TW;
        kon[i] = q < NPK && ch * 4 < a.lddy;
        koff[i] = (unsigned)((kph[i] * a.Wo + kpw[i]) * a.lddy * 4 + ch * 16);
    }
#pragma unroll
    for (int i = 0; i < NQC; ++i) {
        const int q = wave + 4 * i;
        const int row = q * RPC + lane / (8 * CI), ch = lane % (8 * CI);
        xph[i] = row / PW; xpw[i] = row % PW;
        xon[i] = q < NPC && row < PP && ch * 4 < a.ldx;
        xoff[i] = (unsigned)((xph[i] * a.W + xpw[i]) * a.ldx * 4 + ch * 16);
    }
    const int tpi = tiles_h * tiles_w;
    auto issue = [&](int tile, int stage) __attribute__((always_inline)) {
        const int n = tile / tpi, r = tile - n * tpi;
        const int th = r / tiles_w, tw = r - th * tiles_w;
        const int h0 = th * TH, w0 = tw * TW;
        const unsigned sb = smem_base + (unsigned)stage * STAGE;
        const unsigned kbase = (unsigned)(((n * a.Ho + h0) * a.Wo + w0) * a.lddy * 4);
        const int xh0 = h0 - a.pad_t, xw0 = w0 - a.pad_l;
        const int xbase = ((n * a.H + xh0) * a.W + xw0) * a.ldx * 4;            // may be negative: only used with in-image pixels
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            if ((int)wave_u + 4 * i >= NPK) continue;       // (wave-uniform: a piece past the tile would land in the other stage)
            const bool ok = kon[i] && h0 + kph[i] < a.Ho && w0 + kpw[i] < a.Wo;
            glds16_buf(rdy, ok ? kbase + koff[i] : 0xFFFFFFF0u, sb + (wave_u + 4u * (unsigned)i) * 1024u);
        }
#pragma unroll
        for (int i = 0; i < NQC; ++i) {
            if ((int)wave_u + 4 * i >= NPC) continue;
            const bool ok = xon[i] && (unsigned)(xh0 + xph[i]) < (unsigned)a.H && (unsigned)(xw0 + xpw[i]) < (unsigned)a.W;
            glds16_buf(rx, ok ? (unsigned)(xbase + (int)xoff[i]) : 0xFFFFFFF0u, sb + XOFF + (wave_u + 4u * (unsigned)i) * 1024u);
        }
    };

    f32x16_v acc[NT][KI][CI];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < KI; ++i)
#pragma unroll
            for (int j = 0; j < CI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;
    float bsum[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) bsum[i] = 0.f;

    int stage = 0;
    int tile = blockIdx.x;
    if (tile < total_tiles) issue(tile, 0);
    for (; tile < total_tiles; tile += gridDim.x) {
        wait_vmcnt<0>();
        block_barrier();                                   // this tile's operands landed for every wave; the other stage is free
        if (tile + (int)gridDim.x < total_tiles) issue(tile + gridDim.x, stage ^ 1);
        const char* sK = smem + stage * STAGE;
        const char* sX = sK + XOFF;
        const int pbase = wave * (TP / 4);
        // fragments of step t + 1 are read while the MFMAs of step t issue (two register sets: without it every step exposed one LDS round trip in front of
        // its first MFMA -- 154 us instead of ~110 for the 28 -> 28 layer at 200 x 200)
        float av[2][KI], bv[2][NT][CI];
        auto rd = [&](int t, float (&A)[KI], float (&B)[NT][CI]) __attribute__((always_inline)) {
            const int p = pbase + 2 * t + hi;
            const int h = p >> 5, w = p & 31;
#pragma unroll
            for (int i = 0; i < KI; ++i) A[i] = *reinterpret_cast<const float*>(sK + p * RBK + (i * 32 + l31) * 4);
            const char* xb = sX + (R3 ? h * PW + w : p) * RBC + l31 * 4;
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int dr = tp / 3, ds = tp - dr * 3;
#pragma unroll
                for (int j = 0; j < CI; ++j) B[tp][j] = *reinterpret_cast<const float*>(xb + (R3 ? dr * PW + ds : 0) * RBC + j * 128);
            }
        };
        rd(0, av[0], bv[0]);
#pragma unroll 2
        for (int t = 0; t < TP / 8; ++t) {
            const int cur = t & 1;
            if (t + 1 < TP / 8) rd(t + 1, av[cur ^ 1], bv[cur ^ 1]);
#pragma unroll
            for (int i = 0; i < KI; ++i) bsum[i] += av[cur][i];
#pragma unroll
            for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                for (int j = 0; j < CI; ++j)
#pragma unroll
                    for (int i = 0; i < KI; ++i) acc[tp][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][tp][j], acc[tp][i][j], 0, 0, 0);
        }
        stage ^= 1;
    }
    wait_vmcnt<0>();
    __syncthreads();                                       // every wave is past its last fragment read: LDS becomes the flush image
    // flush: the four waves' accumulators summed in LDS (wave after wave, a thread owns its slots), then float atomics by all threads
    float* img = reinterpret_cast<float*>(smem);            // [NT * KI * CI * 16][64]
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < KI; ++i)
#pragma unroll
                    for (int j = 0; j < CI; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float* slot = img + (((t * KI + i) * CI + j) * 16 + e) * 64 + lane;
                            *slot = wv == 0 ? acc[t][i][j][e] : *slot + acc[t][i][j][e];
                        }
        }
        __syncthreads();
    }
    // deterministic mode (round 6): this workgroup's whole K x RSC partial goes to ws[workgroup] with plain stores (zeros included) and the reduction launch adds
    // the workgroups in order; the tile walk of a workgroup is fixed, so the result is bit-identical from run to run
    float* const wdst = a.ws ? a.ws + (size_t)blockIdx.x * a.K * a.RSC : a.dw;
    for (int idx = tid; idx < NT * KI * CI * 16 * 64; idx += 256) {
        const int ln = idx & 63, e = (idx >> 6) & 15, tij = idx >> 10;
        const int j = tij % CI, i = (tij / CI) % KI, t = tij / (CI * KI);
        const int k = i * 32 + 8 * (e >> 2) + 4 * (ln >> 5) + (e & 3), c = j * 32 + (ln & 31);
        const float v = img[idx];
        if (k < a.K && c < a.C) {
            if (a.ws) wdst[(size_t)k * a.RSC + t * a.C + c] = v;
            else if (v != 0.f) atomicAdd(wdst + (size_t)k * a.RSC + t * a.C + c, v);
        }
    }
    if (a.dbias != nullptr) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const float v = bsum[i] + __shfl_xor(bsum[i], 32);
            if (hi == 0) img[(wave * KI + i) * 32 + l31] = v;
        }
        __syncthreads();
        if (tid < 32 * KI && tid < a.K) {
            const int i = tid >> 5, l = tid & 31;
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) v += img[(wv * KI + i) * 32 + l];
            if (a.bws) a.bws[(size_t)blockIdx.x * a.K + tid] = v;
            else if (v != 0.f) atomicAdd(a.dbias + tid, v);
        }
    }
}
}  // namespace

bool wgrad_f32_narrow_supported(const WgradArgs& a, int dtype) {
    if (dtype != ODTK_F32 || (a.dbg2 & 32)) return false;      // (deterministic mode, round 6: one partial per workgroup + the reduction launch)
    const bool r3 = a.R == 3 && a.S == 3 && a.pad_t == 1 && a.pad_l == 1, r1 = a.R == 1 && a.S == 1 && a.pad_t == 0 && a.pad_l == 0;
    if (!(r3 || r1) || a.stride != 1 || a.dil != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (r3 ? !(a.K <= 32 && a.C <= 32) : !(a.K <= 64 && a.C <= 64)) return false;
    return a.ldx % 4 == 0 && a.lddy % 4 == 0 && (long long)a.N * a.H * a.W * a.ldx * 4 < (1ll << 31) && (long long)a.P * a.lddy * 4 < (1ll << 31) &&
           (long long)a.P >= 4096;                            // (small maps: the split-pixel kernel fills the chip better than a handful of 256-pixel tiles)
}

int launch_wgrad_f32_narrow(WgradArgs& a, hipStream_t st) {
    if (g_num_cu == 0) query_num_cu();
    const bool r3 = a.R == 3;
    const int TH = r3 ? 8 : 4;
    const int tiles_h = ceil_div(a.H, TH), tiles_w = ceil_div(a.W, 32);
    const int total = a.N * tiles_h * tiles_w;
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 4);
    a.dy_bytes = (unsigned)((size_t)a.P * a.lddy * 4);
    const int KI = a.K <= 32 ? 1 : 2, CI = a.C <= 32 ? 1 : 2;
    const int per_cu = r3 ? 1 : (KI * CI == 1 ? 2 : 1);
    const int grid = total < g_num_cu * per_cu ? total : g_num_cu * per_cu;
    if (int e = wgrad_split_scratch(a, grid, a.dbias ? grid : 0, st)) return e;
#define ODTK_WN(R3_, KI_, CI_) hipLaunchKernelGGL((wgrad_f32_narrow_kernel<R3_, KI_, CI_>), dim3(grid), dim3(256), 0, st, a, tiles_h, tiles_w, total)
    if (r3) ODTK_WN(true, 1, 1);
    else if (KI == 1 && CI == 1) ODTK_WN(false, 1, 1);
    else if (KI == 1) ODTK_WN(false, 1, 2);
    else if (CI == 1) ODTK_WN(false, 2, 1);
    else ODTK_WN(false, 2, 2);
#undef ODTK_WN
    wgrad_split_reduce(a, st);
    return 0;
}

bool wgrad_v3_supported(const WgradArgs& a, int dtype) {
    // (K = 64 with fewer than 256 columns -- conv1_1's 72 -- stays on conv_wgrad_dma_kernel: 185 us there, 208 us here, measured in round 2)
    return dtype == ODTK_BF16 && (a.K > 64 || (a.K == 64 && a.RSC >= 256)) && a.lddy % 8 == 0 && a.ldx % 8 == 0 && a.C % 8 == 0 &&
           (long long)a.P * a.lddy * 2 < (1ll << 31) && (long long)a.N * a.H * a.W * a.ldx * 2 < (1ll << 31);
}

// ---------------------------------------------------------------------------------------
// "v8" filter gradient (round 2): FOUR waves, one per SIMD, every wave a 128 (k) x 128 (columns) block of 4 x 4 accumulator
// tiles -- the wave shape of the 128 x 512 halo kernel: 16 transpose reads + 8 | 10 LDS-DMA pieces per 32 MFMAs, 768 | 832 LDS
// bytes per MFMA against 1408 of conv_wgrad_v3_kernel (these kernels run at MFMA time + LDS time).
// WPN = 2: 256 (k) x 256 (columns) tile, waves 2 x 2 (Cout a multiple of 256: conv3_x .. conv7);
// WPN = 1: 128 (k) x 512 (columns), waves 1 x 4 (Cout <= 128: conv2_2, the 100-channel head).
// 32 pixels per k-slab, three stages, every slab issues the same number of pieces (past the end: out-of-range offsets = zero
// fill into a stage nobody reads) so the counted vmcnt is a constant; transpose reads of the second 16 pixels and the DMA
// pieces are pinned between the MFMAs.  The bias gradient (column sums of dy) is dealt over the column tiles of a k range:
// tile tq sums the slabs kt = tq (mod tiles_q), so no workgroup is the slow one of its round.
// ---------------------------------------------------------------------------------------
template <int WPN>
__global__ void __launch_bounds__(256) conv_wgrad_v8_kernel(const WgradArgs a) {
    constexpr int PI = 4, QI = 4;
    constexpr int NPS = WPN, NQS = 4 / WPN;          // 128-channel dy sub-slabs, 128-column x sub-slabs
    constexpr int NSUB = NPS + NQS;
    constexpr int PKE = 32;                          // pixels per k-slab
    constexpr int OPB = PKE * 256;                   // bytes per sub-slab (8 KiB)
    constexpr int STAGE = NSUB * OPB;
    constexpr int NST = 3;
    constexpr int NDMA = 2 * NSUB;                   // pieces per wave and slab (8 | 10)
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WPN, wq = wave / WPN;
    const int ntiles = a.tiles_p * a.tiles_q;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vb / ntiles, tile = vb - split * ntiles;
    const int tq = tile / a.tiles_p, tp = tile - tq * a.tiles_p;
    const int p0 = tp * (128 * NPS), q0 = tq * (128 * NQS);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rdy = make_rsrc(a.dy, a.dy_bytes);

    // DMA lane role: pixel row dr of the 4-pixel piece, logical 16-B chunk dch (source-side swizzle)
    const int dr = lane >> 4;
    const int dch = (lane & 15) ^ (dr << 2);
    const int pch = p0 + dch * 8;
    // x sub-slab s of this lane: tap offset as a packed (row, column) pair of 16-bit adds, byte offset the tap adds to the pixel's
    // address; an out-of-range column chunk gets a tap offset that fails the bounds check of every pixel
    unsigned pkd[NQS];
    int qd[NQS];
#pragma unroll
    for (int s = 0; s < NQS; ++s) {
        const int j0 = q0 + s * 128 + dch * 8;
        pkd[s] = 0x80008000u; qd[s] = 0;
        if (j0 < a.RSC) {
            const int rs = j0 / a.C;
            const int qc = j0 - rs * a.C;
            const int qr = rs / a.S, qs = rs - qr * a.S;
            const int dh = qr * a.dil - a.pad_t, dw = qs * a.dil - a.pad_l;
            pkd[s] = ((unsigned)(dh & 0xffff) << 16) | (unsigned)(dw & 0xffff);
            qd[s] = ((dh * a.W + dw) * a.ldx + qc) * 2;
        }
    }
    const unsigned lim = ((unsigned)(a.H - 1) << 16) | (unsigned)(a.W - 1);
    const int HoWo = a.Ho * a.Wo;
    const int iters_total = (a.P + PKE - 1) / PKE;
    const int it0 = split * a.iters_per_split;
    int it1 = it0 + a.iters_per_split;
    if (it1 > iters_total) it1 = iters_total;
    if (it0 >= it1) return;
    const int nk = it1 - it0;

    // The two pixels of this lane per slab (pieces wave and wave + 4) WALK: one slab on = 32 pixels = dn images + dho rows + dwo
    // columns, carried with compares and adds -- no per-slab division or 32-bit multiply (quarter rate, and with one wave per
    // SIMD every VALU cycle outside an MFMA shadow is lost).  Past the end of dy the buffer range check returns zeros, and what
    // a slab past this block's pixel range fetches lands in a stage nobody reads, so the pieces need no pixel bound at all.
    const int l2 = a.ldx * 2;
    const int dn = PKE / HoWo, r1 = PKE - dn * HoWo, dho = r1 / a.Wo, dwo = r1 - dho * a.Wo;
    const int DHS = dho * a.stride, DWS = dwo * a.stride, WoS = a.Wo * a.stride, HoS = a.Ho * a.stride;
    const unsigned DX = (unsigned)((dn * a.H * a.W + DHS * a.W + DWS) * l2);
    const unsigned CW = (unsigned)((a.stride * a.W - WoS) * l2), CH = (unsigned)((a.H * a.W - HoS * a.W) * l2);
    const unsigned DY = (unsigned)(PKE * a.lddy * 2);
    int hi0[2], wi0[2];
    unsigned xo[2], dyo[2], pk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = it0 * PKE + (wave + 4 * h) * 4 + dr;
        const unsigned n = fdiv((unsigned)p, a.div_howo);
        const unsigned rem = (unsigned)p - n * (unsigned)HoWo;
        const unsigned ho = fdiv(rem, a.div_wo);
        const unsigned wo = rem - ho * (unsigned)a.Wo;
        hi0[h] = (int)ho * a.stride;
        wi0[h] = (int)wo * a.stride;
        xo[h] = (unsigned)(((int)n * a.H + hi0[h]) * a.W + wi0[h]) * (unsigned)l2;
        dyo[h] = (unsigned)((p * a.lddy + pch) * 2);
        pk[h] = ((unsigned)hi0[h] << 16) | (unsigned)wi0[h];
    }
    auto advance = [&](int h) __attribute__((always_inline)) {
        wi0[h] += DWS; hi0[h] += DHS; xo[h] += DX; dyo[h] += DY;
        const bool cw = wi0[h] >= WoS;
        wi0[h] -= cw ? WoS : 0; hi0[h] += cw ? a.stride : 0; xo[h] += cw ? CW : 0u;
        const bool chh = hi0[h] >= HoS;
        hi0[h] -= chh ? HoS : 0; xo[h] += chh ? CH : 0u;
        pk[h] = ((unsigned)hi0[h] << 16) | (unsigned)wi0[h];
    };
    typedef unsigned short u16x2_v __attribute__((ext_vector_type(2)));
    // piece q of the current slab: q < 2 NPS -> dy sub-slab q / 2, else x sub-slab (q - 2 NPS) / 2; pixel half q & 1
    auto piece = [&](int q, int stage) __attribute__((always_inline)) {
        const int h = q & 1, sub = q >> 1;
        const unsigned dst = smem_base + (unsigned)(stage * STAGE + sub * OPB) + (unsigned)(wave + 4 * h) * 1024u;
        if (sub < NPS) {
            glds16_buf_nc(rdy, pch + sub * 128 < a.lddy ? dyo[h] + (unsigned)(sub * 256) : 0xFFFFFFF0u, dst);
        } else {
            const int s = sub - NPS;
            const u16x2_v t = __builtin_bit_cast(u16x2_v, pk[h]) + __builtin_bit_cast(u16x2_v, pkd[s]);
            const u16x2_v m = __builtin_elementwise_min(t, __builtin_bit_cast(u16x2_v, lim));
            const bool ok = __builtin_bit_cast(unsigned, m) == __builtin_bit_cast(unsigned, t);
            glds16_buf_nc(rx, ok ? xo[h] + (unsigned)qd[s] : 0xFFFFFFF0u, dst);
        }
    };

    // transpose-read lane role (see conv_wgrad_dma_kernel): group g = lane>>4, c = lane&15
    const int g = lane >> 4, c = lane & 15;
    const int rr = c >> 2;
    unsigned fo[PI + QI];                            // 4 dy fragments (k rows), 4 x fragments (columns)
#pragma unroll
    for (int i = 0; i < PI + QI; ++i) {
        const int ch = ((i & 3) * 32 + 16 * (g & 1)) / 8 + ((c & 3) >> 1);
        const int sub = i < PI ? wp : NPS + wq;
        fo[i] = smem_base + (unsigned)(sub * OPB + (2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8);
    }

    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum[PI] = {0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int q = 0; q < NDMA; ++q) piece(q, 0);
    advance(0); advance(1);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) piece(q, 1);
    int st_c = 0, st_n = 2;
    int bias_cnt = tq, bias_w = 0;                   // slabs until this tile's next bias turn; the column wave that takes it
    for (int kt = 0; kt < nk; ++kt) {
        if (a.dbg & 4) wait_vmcnt<0>(); else wait_vmcnt<NDMA>();
        block_barrier();
        unsigned fb[PI + QI];                        // fragment addresses in this stage: the reads below only add immediates
        const unsigned so = (unsigned)st_c * STAGE;
#pragma unroll
        for (int i = 0; i < PI + QI; ++i) fb[i] = fo[i] + so;
        uint2 fr[2][PI + QI][2];
        auto rd = [&](int ks, int t) __attribute__((always_inline)) {      // read t of k-step ks: fragments in the order P0 Q0 P1 P2 P3 Q1 Q2 Q3
            constexpr int ORD[8] = {0, 4, 1, 2, 3, 5, 6, 7};
            const int f = ORD[t >> 1], half = t & 1;
            fr[ks][f][half] = lds_tr16(fb[f] + (unsigned)(ks * 4096 + half * 1024));
        };
#pragma unroll
        for (int t = 0; t < 16; ++t) rd(0, t);
        const bool bias_turn = a.dbias != nullptr && bias_cnt == 0 && wq == bias_w;      // wave-uniform
        static_for<2>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int j = 0; j < QI; ++j)
#pragma unroll
                for (int i = 0; i < PI; ++i) {
                    const int mi = j * PI + i, slot = ks * 16 + mi;
                    const uint4 pf = make_uint4(fr[ks][i][0].x, fr[ks][i][0].y, fr[ks][i][1].x, fr[ks][i][1].y);
                    const uint4 qf = make_uint4(fr[ks][PI + j][0].x, fr[ks][PI + j][0].y, fr[ks][PI + j][1].x, fr[ks][PI + j][1].y);
                    Mma<bf16_t>::run(pf, qf, acc[i][j]);
                    if (ks == 0) rd(1, mi);
                    // the walk to slab kt + 2 in the shadow of the first MFMAs, then its pieces from slot 4 on
                    if (slot == 1) advance(0);
                    if (slot == 2) advance(1);
#pragma unroll
                    for (int q = 0; q < NDMA; ++q)
                        if (4 + q * 28 / NDMA == slot && !(a.dbg & 4)) piece(q, st_n);
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (bias_turn) {
#pragma unroll
                for (int i = 0; i < PI; ++i)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        bsum[i] = dot2_bf16_ones(fr[ks][i][half].x, bsum[i]);
                        bsum[i] = dot2_bf16_ones(fr[ks][i][half].y, bsum[i]);
                    }
            }
        });
        if (bias_cnt == 0) { bias_cnt = a.tiles_q; bias_w = bias_w + 1 == 4 / WPN ? 0 : bias_w + 1; }
        --bias_cnt;
        st_c = st_c == 2 ? 0 : st_c + 1;
        st_n = st_n == 2 ? 0 : st_n + 1;
    }
    wait_vmcnt<0>();

    const int l31 = lane & 31, hi = lane >> 5;
    if (a.ws) {
        block_barrier();                                   // every wave is past its last fragment read
        if (!(a.dbg & 16))
            store_partial_tile<PI, QI>(acc, reinterpret_cast<float*>(smem) + wave * (32 * QI * 32), a.ws + (size_t)split * a.K * a.RSC,
                                       p0 + wp * 128, q0 + wq * 128, a.K, a.RSC, lane);
    } else {
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const int col = q0 + wq * 128 + j * 32 + l31;
            if (col >= a.RSC) continue;
#pragma unroll
            for (int i = 0; i < PI; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = p0 + wp * 128 + i * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                    if (k < a.K && !(a.dbg & 16)) atomicAdd(a.dw + (size_t)k * a.RSC + col, acc[i][j][e]);
                }
            }
        }
    }
    if (a.dbias != nullptr) {
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const float t = bsum[i] + __shfl_xor(bsum[i], 32);       // both pixel halves of the k row
            const int k = p0 + wp * 128 + i * 32 + l31;
            if (hi == 0 && k < a.K) {
                // one bias slot per (pixel split, column tile, column wave): every wave writes its sum, zero if it had no turn
                if (a.bws) a.bws[((size_t)(split * a.tiles_q + tq) * (4 / WPN) + wq) * a.K + k] = t;
                else if (t != 0.f) atomicAdd(a.dbias + k, t);
            }
        }
    }
}

// Deterministic split-reduce of the filter gradient, OPT-IN (odtk_debug_set key 5 = 1; config key 'deterministic_wgrad' of the
// model classes): with more than one pixel split the blocks store their partial tiles (full 16-byte rows through a wave-private
// LDS patch) to the per-(device, slot) scratch and wgrad_reduce_kernel adds them in split order -- bit-identical from run to
// run.  The default stays float atomics into dw: measured on the SSD300 step at batch 32, same box, the split-reduce costs
// 2.5-2.7 % (3 210 | 3 186 against 3 292 | 3 284 images/s): it moves splits x |dw| bytes twice (~700 MB per step) where the
// atomics, ~40 us per 256 x 256-tile launch as they are, move them once.
int wgrad_split_scratch(WgradArgs& a, int splits, int bias_slots, hipStream_t st) {
    a.ws = nullptr; a.bws = nullptr; a.nsplit = splits; a.nbslot = bias_slots;
    if (splits < 2 || !g_wgrad_deterministic) return 0;
    float* base = nullptr;
    const size_t wbytes = (size_t)splits * a.K * a.RSC * sizeof(float);
    if (int e = conv_scratch(wbytes + (size_t)bias_slots * a.K * sizeof(float), st, &base)) return e;
    a.ws = base;
    a.bws = bias_slots ? base + (size_t)splits * a.K * a.RSC : nullptr;
    return 0;
}
void wgrad_split_reduce(const WgradArgs& a, hipStream_t st) {
    if (!a.ws) return;
    const long long n4 = (long long)a.K * a.RSC / 4;
    if (a.nsplit >= 64) {                                  // many partials of a small gradient: 16 lanes per output
        const int wb = (int)((n4 + 15) / 16), bb = a.bws ? ceil_div(a.K, 16) : 0;
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(wb + bb), dim3(256), 0, st, a.ws, a.nsplit, n4, a.dw, a.bws, a.nbslot, a.K, a.dbias, wb);
        return;
    }
    const int wblocks = (int)((n4 + 255) / 256), bblocks = a.bws ? ceil_div(a.K, 16) : 0;      // (bias: 16 outputs per workgroup, 16 lanes each)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + bblocks), dim3(256), 0, st, a.ws, a.nsplit, n4, a.dw, a.bws, a.nbslot, a.K,
                       a.dbias, wblocks);
}

// Returns false (nothing launched) when the pixel range per block would be short: every block ends with 64 K float atomics (~40 us
// measured, independent of the layer), which only >= ~100 slabs of 32 pixels amortise better than the 32 K of the 128 x 256 kernel.
// Measured at batch 32, same box, v8 | v3: conv3_2 223 | 241 us, conv4_2 220 | 252, conv6 134 | 150 (taken); conv4_1 138 | 133, conv3_1
// 150 | 138, conv5_x 97 | 79, conv7 73 | 49 (left on v3); the 128 x 512 variant (WPN = 1) lost on conv2_2 (299 | 272) and the heads.
bool launch_wgrad_v8(WgradArgs& a, hipStream_t st) {
    const int tiles_p = ceil_div(a.K, 256), tiles_q = ceil_div(a.RSC, 256);
    const int tiles = tiles_p * tiles_q;
    const int iters_total = ceil_div(a.P, 32);
    if (g_num_cu == 0) query_num_cu();
    int best_s = 1;
    double best_t = 1e30;
    const int smax = iters_total < 8 ? 1 : iters_total / 8;
    for (int s = 1; s <= smax && s <= 1024; ++s) {      // rounds x (slabs x ~0.85 us + prologue and 64 K float atomics per block)
        const int ips = ceil_div(iters_total, s);
        const int sp = ceil_div(iters_total, ips);
        const double rounds = (double)ceil_div(tiles * sp, g_num_cu);
        const double tt = rounds * (ips * 0.85 + 45.0);
        if (tt < best_t - 1e-9) { best_t = tt; best_s = sp; }
    }
    const int ips = ceil_div(iters_total, best_s);
    if (ips < 110 && !(a.dbg & (1 << 30))) return false;
    a.tiles_p = tiles_p; a.tiles_q = tiles_q;
    a.iters_per_split = ips;
    const int splits = ceil_div(iters_total, a.iters_per_split);
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 2);
    a.dy_bytes = (unsigned)((size_t)a.P * a.lddy * 2);
    if (wgrad_split_scratch(a, splits, a.dbias ? splits * tiles_q * 2 : 0, st)) return false;
    hipLaunchKernelGGL(conv_wgrad_v8_kernel<2>, dim3(tiles * splits), dim3(256), 0, st, a);
    wgrad_split_reduce(a, st);
    a.which = 8;
    return true;
}

int launch_wgrad_v3(WgradArgs& a, hipStream_t st) {
    // four-wave kernel with 128 x 128 wave tiles: Cout a multiple of 256 (no tile waste) and long pixel ranges per block
    // (dbg bit 29 = off, bit 30 = also with short ranges: A/B)
    if (!(a.dbg & (1 << 29)) && a.K % 256 == 0 && a.RSC % 256 == 0 && launch_wgrad_v8(a, st)) return 0;
    a.tiles_p = ceil_div(a.K, 128);
    a.tiles_q = ceil_div(a.RSC, 256);
    const int tiles = a.tiles_p * a.tiles_q;
    const int iters_total = ceil_div(a.P, 64);
    // one 512-thread block per CU.  Pick the pixel split count by a small cost model: rounds of blocks x
    // (k-slabs per block x ~1.0 us + ~16 us of prologue and 32 K float atomics per block: fitted on conv5_x / conv6,
    // 101 -> 81 us and 186 -> 153 us against the earlier 6 us)
    if (g_num_cu == 0) query_num_cu();
    int best_s = 1;
    double best_t = 1e30;
    const int smax = iters_total < 4 ? 1 : iters_total / 4;
    for (int s = 1; s <= smax && s <= 1024; ++s) {
        const int ips = ceil_div(iters_total, s);
        const int sp = ceil_div(iters_total, ips);
        const double rounds = (double)ceil_div(tiles * sp, g_num_cu);
        const double tt = rounds * (ips * 1.0 + 16.0);
        if (tt < best_t - 1e-9) { best_t = tt; best_s = sp; }
    }
    a.iters_per_split = ceil_div(iters_total, best_s);
    const int splits = ceil_div(iters_total, a.iters_per_split);
    a.x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.ldx * 2);
    a.dy_bytes = (unsigned)((size_t)a.P * a.lddy * 2);
    if (int e = wgrad_split_scratch(a, splits, a.dbias ? splits : 0, st)) return e;
    hipLaunchKernelGGL(conv_wgrad_v3_kernel, dim3(tiles * splits), dim3(512), 0, st, a);
    wgrad_split_reduce(a, st);
    a.which = 3;
    return 0;
}

// ---------------------------------------------------------------------------------------
// "x3" (round 4; the arithmetic behind ODTK_F32X3 descriptors): f32 convolutions on the bf16 MFMA kernels by operand splitting.
//     a = a_hi + a_lo with a_hi = bf16(a), a_lo = bf16(a - a_hi);    a * b = a_hi b_hi + a_hi b_lo + a_lo b_hi + O(2^-17 |a b|),   accumulated in f32 by the MFMA.
// The three partial products are ONE implicit GEMM with a three times longer reduction over [hi | hi | lo] x [hi | lo | hi].  The gathered operand (forward: the
// activation, input gradient: dy) is STORED as two parts [hi | lo] (pitch 2 ldc + 64) and the kernels read its first part twice (GatherArgs::x3c); filters are stored
// with all three parts per tap.  For the filter gradient, whose reduction runs over pixels, x is stacked [hi ; lo ; hi] and dy [hi ; hi ; lo] along the IMAGE axis
// (3 N images) and the existing filter-gradient kernels run unchanged; their f32 result is dW.
// The f32 MFMA of gfx950 runs at 1/16 of the bf16 rate (157 TFLOP/s; the identity-free conv + norm stacks of RetinaNet, RefineDet320, PFPNetR, LH_RCNN do not survive
// bf16 OPERANDS); three bf16 products are 5.3x that.  Outputs leave as f32 rows straight from the accumulators (x3_store4: the raster-run halo kernel's
// F32OUT instantiations and the 8-wave kernel's single-part launch) or, with few tiles, through split-K partial tiles and splitk_finish_f32_kernel.
// ---------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void split_hi_lo(float v, bf16_t& hi, bf16_t& lo) {
    hi = f32_to_bf16(v);
    lo = f32_to_bf16(v - __uint_as_float((unsigned)hi << 16));
}
// eight consecutive channels of one f32 row (zero past C) as two packed bf16x8: the high halves and the low halves
__device__ __forceinline__ void split8(const float* __restrict__ row, int c0, int C, bool vec, uint4& vh, uint4& vl) {
    float v[8];
    if (vec && c0 + 8 <= C) {                              // 16-byte aligned: pitch and c0 are multiples of 4
        const float4 a = *reinterpret_cast<const float4*>(row + c0), b = *reinterpret_cast<const float4*>(row + c0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? row[c0 + e] : 0.f;
    }
    unsigned short h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split_hi_lo(v[e], h[e], l[e]);
    vh = make_uint4(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16, h[4] | (unsigned)h[5] << 16, h[6] | (unsigned)h[7] << 16);
    vl = make_uint4(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16, l[4] | (unsigned)l[5] << 16, l[6] | (unsigned)l[7] << 16);
}
// dst [M][ldrow >= nparts * ldc] (nparts = 2 | 3; parts hi/lo by `pattern` bit i = part i is the low half) from src [M][lds] f32, C valid channels; pad columns of every part zeroed
// One split job: `rows` = false -> dst rows [part 0 | part 1 | ...] of ldc channels each at pitch ldrow (split3_chan), true -> dst [3 M][ld] row blocks (split3_rows).
// Round 6: the two operands of a pass (pixels + filter; x + dy of a filter gradient) are split by ONE launch (split3_pair_kernel: workgroups [0, j0.blocks) run job 0,
// the rest job 1) -- the filter split is ~3 us of work behind a launch boundary of its own, ~220 times per RetinaNet / YOLOv3 step.
struct SplitJob {
    const float* src; long long M; int C, lds; bf16_t* dst; int ld, pattern, nparts, ldrow; FastDiv dc; int rows, blocks, zero_lo;
};
__device__ __forceinline__ void split3_job(const SplitJob& j, int bid, int nblk) {
    const int cpr = j.ld >> 3;                             // 8-channel chunks per part row
    const long long total = j.M * cpr;
    const bool vec = (j.lds & 3) == 0 && ((uintptr_t)j.src & 15) == 0;
    for (long long i = (long long)bid * 256 + threadIdx.x; i < total; i += (long long)nblk * 256) {
        const long long m = total < (1ll << 31) ? (long long)fdiv((unsigned)i, j.dc) : i / cpr;
        const int c0 = (int)(i - m * cpr) * 8;
        uint4 vh, vl;
        split8(j.src + m * j.lds, c0, j.C, vec, vh, vl);
        if (j.zero_lo) vl = make_uint4(0u, 0u, 0u, 0u);        // experiment switch: see set_x3_zero_lo
        if (j.rows) {
#pragma unroll
            for (int part = 0; part < 3; ++part) *reinterpret_cast<uint4*>(j.dst + ((long long)part * j.M + m) * j.ld + c0) = ((j.pattern >> part) & 1) ? vl : vh;
        } else {
            bf16_t* row = j.dst + m * j.ldrow + c0;
#pragma unroll
            for (int part = 0; part < 3; ++part)
                if (part < j.nparts) *reinterpret_cast<uint4*>(row + part * j.ld) = ((j.pattern >> part) & 1) ? vl : vh;
        }
    }
}
__global__ void __launch_bounds__(256) split3_chan_kernel(const SplitJob j) { split3_job(j, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) split3_rows_kernel(const SplitJob j) { split3_job(j, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) split3_pair_kernel(const SplitJob j0, const SplitJob j1) {
    if ((int)blockIdx.x < j0.blocks) split3_job(j0, blockIdx.x, j0.blocks);
    else split3_job(j1, (int)blockIdx.x - j0.blocks, j1.blocks);
}
// sum of the f32 partial tiles + bias (+ ReLU, ReLU mask, accumulate: the semantics of x3_store4) -> f32 y; pad columns (>= K) zeroed
__global__ void __launch_bounds__(256) splitk_finish_f32_kernel(const float* __restrict__ ws, int ksplit, long long M, int K, int ldy, const float* __restrict__ bias, int relu,
                                                                const float* __restrict__ mask, int ldmask, int accumulate, float* y) {
    const int cpr = ldy >> 2;
    const long long total = M * cpr;
    const long long stride = M * ldy;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / cpr;
        const int c0 = (int)(i - m * cpr) * 4;
        float4 v = *reinterpret_cast<const float4*>(ws + m * ldy + c0);
        for (int p = 1; p < ksplit; ++p) {
            const float4 u = *reinterpret_cast<const float4*>(ws + p * stride + m * ldy + c0);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
        float pv[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (accumulate) { const float4 t = *reinterpret_cast<const float4*>(y + m * ldy + c0); pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w; }
        if (mask) { const float4 t = *reinterpret_cast<const float4*>(mask + m * ldmask + c0); mk[0] = t.x; mk[1] = t.y; mk[2] = t.z; mk[3] = t.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c0 + e >= K) { o[e] = 0.f; continue; }
            if (bias) o[e] += bias[c0 + e];
            if (relu) o[e] = fmaxf(o[e], 0.f);
            o[e] += pv[e];
            if (!(mk[e] > 0.f)) o[e] = 0.f;
        }
        *reinterpret_cast<float4*>(y + m * ldy + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
static int grid_1d(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
}  // namespace

// EXPERIMENT (round 6, odtk_debug_set key 6 bit 17): with the low halves zeroed the x3 engine computes hi x hi only -- f32 tensors, ONE bf16 product per f32 product:
// the numerics a cheaper mixed engine would have, at the x3 engine's cost; used to put those numerics through the bf16 gate before building that engine.
static bool g_x3_zero_lo = false;
void set_x3_zero_lo(bool on) { g_x3_zero_lo = on; }
static SplitJob split_job(const float* src, long long M, int C, int lds, void* dst, int ld, int pattern, int nparts, int ldrow, int rows) {
    SplitJob j;
    j.src = src; j.M = M; j.C = C; j.lds = lds; j.dst = (bf16_t*)dst; j.ld = ld; j.pattern = pattern; j.nparts = nparts; j.ldrow = ldrow;
    j.dc = make_fastdiv((unsigned)(ld >> 3)); j.rows = rows; j.blocks = grid_1d(M * (ld >> 3)); j.zero_lo = g_x3_zero_lo ? 1 : 0;
    return j;
}
void launch_split3_chan(const float* src, long long M, int C, int lds, void* dst, int ldc, int pattern, int nparts, int ldrow, hipStream_t st) {
    const SplitJob j = split_job(src, M, C, lds, dst, ldc, pattern, nparts, ldrow, 0);
    hipLaunchKernelGGL(split3_chan_kernel, dim3(j.blocks), dim3(256), 0, st, j);
}
void launch_split3_rows(const float* src, long long M, int C, int lds, void* dst, int ldd, int pattern, hipStream_t st) {
    const SplitJob j = split_job(src, M, C, lds, dst, ldd, pattern, 3, 0, 1);
    hipLaunchKernelGGL(split3_rows_kernel, dim3(j.blocks), dim3(256), 0, st, j);
}
// two channel-layout splits (pixels + filter of a forward / input-gradient pass) in one launch
void launch_split3_chan2(const float* src0, long long M0, int C0, int lds0, void* dst0, int ldc0, int pattern0, int nparts0, int ldrow0,
                         const float* src1, long long M1, int C1, int lds1, void* dst1, int ldc1, int pattern1, int nparts1, int ldrow1, hipStream_t st) {
    const SplitJob j0 = split_job(src0, M0, C0, lds0, dst0, ldc0, pattern0, nparts0, ldrow0, 0), j1 = split_job(src1, M1, C1, lds1, dst1, ldc1, pattern1, nparts1, ldrow1, 0);
    hipLaunchKernelGGL(split3_pair_kernel, dim3(j0.blocks + j1.blocks), dim3(256), 0, st, j0, j1);
}
// two row-block splits (x and dy of a filter gradient) in one launch
void launch_split3_rows2(const float* src0, long long M0, int C0, int lds0, void* dst0, int ldd0, int pattern0,
                         const float* src1, long long M1, int C1, int lds1, void* dst1, int ldd1, int pattern1, hipStream_t st) {
    const SplitJob j0 = split_job(src0, M0, C0, lds0, dst0, ldd0, pattern0, 3, 0, 1), j1 = split_job(src1, M1, C1, lds1, dst1, ldd1, pattern1, 3, 0, 1);
    hipLaunchKernelGGL(split3_pair_kernel, dim3(j0.blocks + j1.blocks), dim3(256), 0, st, j0, j1);
}
// how many f32 partial tiles the x3 gather of `a` writes (1 = straight into the output)
int gather_x3_ksplit(const GatherArgs& a) {
    if (g_num_cu == 0) query_num_cu();
    const int PT = a.K <= 64 ? 64 : 128;
    const int tiles = ceil_div(a.K, PT) * ceil_div(a.M, 256);
    const int nk = ceil_div(a.Kdim, 64);
    int ks = 1;
    if (tiles <= g_num_cu / 2 && nk >= 8) {
        ks = g_num_cu / tiles;
        if (ks > nk / 4) ks = nk / 4;
        if (ks > 32) ks = 32;
        if (ks < 1) ks = 1;
    }
    return ks;
}
// bf16 gather (operands already split: a.x, a.w, a.C = 3 x the logical channels) -> f32 out [M][ldy] (+ bias, ReLU); `partials` holds ksplit f32 tiles when ksplit > 1
int launch_gather_x3(GatherArgs& a, float* out, float* partials, const float* bias, int relu, const float* mask, int ldmask, int accumulate, hipStream_t st) {
    const int PT = a.K <= 64 ? 64 : 128;
    a.tiles_p = ceil_div(a.K, PT);
    a.tiles_q = ceil_div(a.M, 256);
    a.ksplit = gather_x3_ksplit(a);
    a.ws = a.ksplit > 1 ? partials : out;
    a.bias = a.ksplit > 1 ? nullptr : bias;
    a.relu = a.ksplit > 1 ? 0 : relu;
    a.mask = a.ksplit > 1 ? nullptr : (const char*)mask;
    a.ldmask = ldmask;
    a.accumulate = a.ksplit > 1 ? 0 : accumulate;
    // 3x3 / stride 1 / SAME over whole 64-channel chunks (the heads' and the pyramid's 256-channel layers: 768 split channels): the raster-run halo kernel
    const int halo = 2 * a.dil * (a.W + 1);
    // Round 6: stride-2 input gradients on whole 64-channel chunks as four PARITY PHASES in one launch of the 8-wave kernel (what round 5 gave the bf16 engine: a phase
    // walks only the taps of its row / column parity -- 9 tap-slabs per four pixels instead of 36, none of them multiplying zeros), whole reduction per workgroup,
    // f32 rows from the registers.  DarkNet-53's five down-sampling layers; dbg2 bit 13 = off (A/B)
    if (a.idiv == 2 && a.dil == 1 && a.ostride == 1 && a.C % 64 == 0 && a.x3c % 64 == 0 && a.Kdim == a.R * a.S * a.C && a.R * a.S <= 32 && !(a.dbg & 65536) && !(a.dbg2 & 8192)) {
        GatherArgs b = a;
        b.plan_v9_qt = 256;
        if (int e = launch_gather_v9(b, st, g_num_cu)) return e;       // fills b.v9 / b.tiles_q only (plan_v9_qt != 0: no launch)
        if (ceil_div(b.K, PT) * b.tiles_q >= g_num_cu / 2) {                        // (few tiles: the split-K form below keeps more CUs busy)
            a = b;
            a.tiles_p = ceil_div(a.K, PT);
            a.ksplit = 1; a.ws = out; a.bias = bias; a.relu = relu; a.mask = (const char*)mask; a.accumulate = accumulate;
            const int grid_ph = a.tiles_p * a.tiles_q;
            if (PT == 64) hipLaunchKernelGGL((conv_gather_v3_kernel<64, true, false, true, true, true, false, true>), dim3(grid_ph), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v3_kernel<128, true, false, true, true, true, false, true>), dim3(grid_ph), dim3(512), 0, st, a);
            return 0;
        }
    }
    // Round 6: few-tile 3 x 3 layers (DarkNet-53's 13 x 13 / 26 x 26 maps at 8 images -- YOLOv3 trains on this engine by default now) on the 128 x 128 / 64 x 128
    // tiles of the halo kernel, two workgroups per CU, whole reduction per workgroup, f32 rows straight from the registers: what round 5 (r05s) gave the bf16
    // engine instead of a split launch + its finish launch (here: the 8-wave kernel's split-K partials + splitk_finish_f32_kernel).  dbg2 bit 14 = off (A/B).
    if (a.ksplit > 1 && PT == 128 && !(a.dbg & 65536) && !(a.dbg2 & 16384) && halo <= 96 && a.dil == 1 && a.C % 64 == 0 && a.R == 3 && a.S == 3 && a.ostride == 1 &&
        a.idiv == 1 && a.pad_t == 1 && a.pad_l == 1 && a.H == a.Ho && a.W == a.Wo && a.Kdim == 9 * a.C) {
        const int tq128 = ceil_div(a.M, 128), tp64 = ceil_div(a.K, 64);
        const bool t128 = tq128 * a.tiles_p >= (2 * g_num_cu) / 3, t64 = !t128 && tq128 * tp64 >= (2 * g_num_cu) / 3 && !(a.dbg2 & 32768);
        if (t128 || t64) {
            a.ksplit = -1; a.ws = out; a.bias = bias; a.relu = relu; a.mask = (const char*)mask; a.accumulate = accumulate;
            a.tiles_q = tq128;
            if (t128) {
                if (a.W >= 32) hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 1, 2, 2, 2, 2, false, true>), dim3(tq128 * a.tiles_p), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 0, 0, 2, 2, 2, false, true>), dim3(tq128 * a.tiles_p), dim3(256), 0, st, a);
            } else {
                a.tiles_p = tp64;
                if (a.W >= 32) hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 1, 2, 1, 2, 1, false, true>), dim3(tq128 * tp64), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((conv_gather_v6_kernel<7, false, 0, 0, 1, 2, 1, false, true>), dim3(tq128 * tp64), dim3(256), 0, st, a);
            }
            return 0;
        }
    }
    // Round 6: what is left with few tiles -- the 1 x 1 layers of the small maps above all -- on the small-map kernel (conv_v9.hip: 64 x 64 tiles, whole reduction per
    // workgroup) with f32 output, instead of split-K partial tiles + splitk_finish_f32_kernel
    if (a.ksplit > 1 && !(a.dbg & 65536)) {
        GatherArgs b = a;
        b.ksplit = 1; b.ws = out; b.bias = bias; b.relu = relu; b.mask = (const char*)mask; b.accumulate = accumulate;
        if (launch_gather_v9_x3(b, st, g_num_cu)) { a = b; return 0; }
    }
    if (a.ksplit == 1 && PT == 128 && !(a.dbg & 65536) && a.C % 64 == 0 && a.R == 3 && a.S == 3 && a.ostride == 1 && a.idiv == 1 && a.pad_t == a.dil &&
        a.pad_l == a.dil && a.H == a.Ho && a.W == a.Wo && a.Kdim == 9 * a.C) {
        const int tiles = a.tiles_p * a.tiles_q;
        const int tq512 = ceil_div(a.M, 512), tq192 = ceil_div(a.M, 192);
        const long long cost256 = (long long)ceil_div(tiles, g_num_cu) * (256 + 32), cost192 = (long long)ceil_div(tq192 * a.tiles_p, g_num_cu) * (192 + 32);
        a.ksplit = -1;
        if (halo <= 160 && a.dil * a.W >= 64 && tq512 * a.tiles_p >= 2 * g_num_cu) {
            a.tiles_q = tq512;
            hipLaunchKernelGGL((conv_gather_v6_kernel<21, false, 2, 4, 1, 4, 4, false, true>), dim3(tq512 * a.tiles_p), dim3(256), 0, st, a);
            return 0;
        }
        if (halo <= 160 && cost192 < cost256) {
            a.tiles_q = tq192;
            hipLaunchKernelGGL((conv_gather_v6_kernel<11, true, 0, 0, 2, 2, 3, false, true>), dim3(tq192 * a.tiles_p), dim3(256), 0, st, a);
            return 0;
        }
        if (halo <= 160) { hipLaunchKernelGGL((conv_gather_v6_kernel<13, true, 0, 0, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a); return 0; }
        if (halo <= 352 && a.dil * a.W >= 80) {             // rows of 80 .. 175 pixels: single patch buffer, early refill (the group counts of launch_gather_v3)
            const int w = a.dil * a.W;
            if (halo > 320) hipLaunchKernelGGL((conv_gather_v6_kernel<19, false, 5, 10, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            else if (w >= 144) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 4, 9, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            else if (w >= 128) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 4, 8, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            else if (w >= 112) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 3, 7, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            else if (w >= 96) hipLaunchKernelGGL((conv_gather_v6_kernel<18, false, 3, 6, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_v6_kernel<14, false, 2, 5, 2, 2, 4, false, true>), dim3(tiles), dim3(256), 0, st, a);
            return 0;
        }
        a.ksplit = 1;                                     // other row lengths: the 8-wave kernel below
    }
    const int grid = a.tiles_p * a.tiles_q * a.ksplit;
    if (PT == 64) hipLaunchKernelGGL((conv_gather_v3_kernel<64, true, false, true, false, true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_gather_v3_kernel<128, true, false, true, false, true>), dim3(grid), dim3(512), 0, st, a);
    if (a.ksplit > 1)
        hipLaunchKernelGGL(splitk_finish_f32_kernel, dim3(grid_1d((long long)a.M * (a.ldy >> 2))), dim3(256), 0, st, a.ws, a.ksplit, (long long)a.M, a.K, a.ldy, bias, relu,
                           mask, ldmask, accumulate, out);
    return 0;
}

}  // namespace cv
}  // namespace odtk
