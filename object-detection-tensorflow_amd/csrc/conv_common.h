// Shared device helpers of the implicit-GEMM convolution kernels (gfx950 only):
// MFMA wrappers, the swizzled 128-byte-row LDS slab layout, LDS-DMA, argument structs.
#pragma once
#include "common.h"

namespace odtk {
namespace cv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(16))) float f32x16_v;
typedef __attribute__((ext_vector_type(4))) short v4i16_v;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) -- for bodies whose index must fold
// (register-array indices, instruction-offset immediates) even where the loop unroller gives up
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct FastDiv {
    unsigned mul, shift;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FastDiv f) {   // x < 2^31
    return (__umulhi(x, f.mul) + x) >> f.shift;
}

__host__ __device__ __forceinline__ int swz(int row) {
    return (((row >> 1) ^ (row >> 5)) & 1) | (((row >> 3) & 1) << 1) | (((row >> 4) & 1) << 2);
}
// variant without the row-bit-5 term: identical conflict behaviour for the 32-row fragment reads,
// and constant over rows r, r+32, r+64, r+96 (what the LDS-DMA loader needs)
__host__ __device__ __forceinline__ int swz_g(int row) {
    return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 4) & 1) << 2);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KCH = 8;
    static __device__ __forceinline__ void run(const uint4& p, const uint4& q, f32x16_v& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, p),
                                                      __builtin_bit_cast(bf16x8_v, q), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KCH = 4;
    static __device__ __forceinline__ void run(const uint4& p, const uint4& q, f32x16_v& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(p.x), __uint_as_float(q.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(p.y), __uint_as_float(q.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(p.z), __uint_as_float(q.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(p.w), __uint_as_float(q.w), acc, 0, 0, 0);
    }
};

// One staged k-slab: 4 sub-steps of two 16-B slots (lanes 0-31 slot 2ks, lanes 32-63 slot 2ks+1).
template <typename T, int PI, int QI, bool SWZ_G = false>
__device__ __forceinline__ void mma_slab(const char* sP, const char* sQ, int prow0, int qrow0,
                                         int lane, f32x16_v (&acc)[PI][QI]) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int slot = ks * 2 + hi;
        uint4 pf[PI], qf[QI];
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int row = prow0 + i * 32 + l31;
            pf[i] = *reinterpret_cast<const uint4*>(sP + row * 128 + ((slot ^ (SWZ_G ? swz_g(row) : swz(row))) << 4));
        }
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const int row = qrow0 + j * 32 + l31;
            qf[j] = *reinterpret_cast<const uint4*>(sQ + row * 128 + ((slot ^ (SWZ_G ? swz_g(row) : swz(row))) << 4));
        }
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int j = 0; j < QI; ++j) Mma<T>::run(pf[i], qf[j], acc[i][j]);
    }
}

// Same slab product with the fragments of sub-step ks+1 in flight while the MFMAs of sub-step ks
// issue (two register sets): hides LDS latency that grows when LDS-DMA writes share the LDS.
template <typename T, int PI, int QI>
__device__ __forceinline__ void mma_slab_db(const char* sP, const char* sQ, int prow0, int qrow0,
                                            int lane, f32x16_v (&acc)[PI][QI]) {
    const int l31 = lane & 31, hi = lane >> 5;
    uint4 pf[2][PI], qf[2][QI];
    auto load = [&](int ks, uint4 (&p)[PI], uint4 (&q)[QI]) __attribute__((always_inline)) {
        const int slot = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int row = prow0 + i * 32 + l31;
            p[i] = *reinterpret_cast<const uint4*>(sP + row * 128 + ((slot ^ swz_g(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const int row = qrow0 + j * 32 + l31;
            q[j] = *reinterpret_cast<const uint4*>(sQ + row * 128 + ((slot ^ swz_g(row)) << 4));
        }
    };
    load(0, pf[0], qf[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) load(ks + 1, pf[(ks + 1) & 1], qf[(ks + 1) & 1]);
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int j = 0; j < QI; ++j) Mma<T>::run(pf[ks & 1][i], qf[ks & 1][j], acc[i][j]);
    }
}

// XCD-aware bijective remap of the linear block id (block b runs on XCD b % 8): every XCD gets
// a contiguous range of tiles so neighbouring tiles share their operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct GatherArgs {
    const char* x;     // activations gathered along k  [N][H][W][ldx]
    const char* w;     // filter rows [K][R*S*C]
    const float* bias; // [K] or null
    const char* mask;  // relu source (same geometry as y) or null
    char* y;
    int N, H, W, C, ldx;
    int Ho, Wo, K, ldy, ldmask;
    int R, S, ostride, dil, pad_t, pad_l, idiv;
    int M, Kdim, ldw;
    int relu, accumulate;
    int tiles_p, tiles_q;
    FastDiv div_howo, div_wo;
    int dbg;           // perf experiments only (odtk_debug_set key 2): bit0 skip pixel-operand DMA, bit1 skip filter DMA after slab 0
    int dbg2;          // dispatch A/B switches that leave results intact (odtk_debug_set key 6, include/odtk.h): bit 2 = ODTK_F32X3 descriptors on the exact f32 kernels,
                       // bit 3 = on the split path wherever it is supported (also below the size policy: tests), bit 4 = no 32-row f32 filter tile
    int x3c;                 // x3 engine: channels per split part of the PIXEL operand, stored [hi | lo]; reduction channel c reads channel c - x3c when c >= x3c (the
                             // virtual layout [hi | hi | lo] against filters [hi | lo | hi]); 0 = plain operand
    unsigned x_bytes, w_bytes;   // extents of x and w for the buffer-addressed DMA (range check = zero fill)
    int ksplit;        // split-K: blocks per tile (1 = off) and their f32 partial tiles [ksplit][M][ldy]
    float* ws;
    // fused 2x2 / stride-2 max pooling of the (bias + ReLU) output in the epilogue (odtk_conv2d_fwd_pool2x2; conv3x3_c64k64_kernel only):
    // pool_mode 0 = off, 1 = y AND the pooled map, 2 = the pooled map only (y is never written).  ypool [N][ceil(H/2)][ceil(W/2)][ldpool],
    // pidx = the recorded first arg-max (2 bits per channel, one uint16 per 16-byte chunk: the format of odtk_maxpool2x2_fwd_idx)
    char* ypool;
    unsigned short* pidx;
    int ldpool, pool_mode;
    // ... and in the raster-run halo kernel (round 4: conv2_2 + pool2, conv3_3 + pool3): a tile is `pool_rpt` (even) whole image rows of one image, so that both rows
    // of every pooling window are in the tile's LDS image; pool_tpi = tiles per image (the last one may hold fewer rows)
    int pool_rpt, pool_tpi;
    FastDiv div_ptpi, div_wp;
    // Tile order of the persistent conv1_x kernels (round 3): 1 = walk the tiles from the LAST to the first.  The maps of conv1_x are 369 MB, more than the
    // 256 MB of memory-side cache: a kernel that starts where its producer STOPPED finds the producer's last ~2/3 still cached.  Forward: conv1_1 walks up,
    // conv1_2 walks down; backward: pool1's gradient is written upwards, conv1_2's filter gradient walks down, its input gradient up, conv1_1's filter gradient down.
    int rev;
    // ReLU mask as sign bits (round 4; the 3(8) -> 64 first-layer kernel writes them, the 64 -> 64 halo kernel's input-gradient pass reads them): one byte per
    // (pixel, 16-byte chunk), bit e = channel 8 * chunk + e of the activation is > 0.  conv1_2's dgrad read the 369-MB activation only for these signs.
    unsigned char* ybits;            // written beside y (forward), [M][ldy / 8]
    const unsigned char* mask_bits;  // read instead of `mask` (input gradient), [M][ldmask / 8]
    // Round 5: split-K of the raster-run halo kernel over whole 64-channel CHUNKS (its F32OUT instantiations only): `cs_split` consecutive blocks share a tile, block
    // `part` reduces chunks [part * ncs / cs_split, (part + 1) * ncs / cs_split) and stores its f32 partial tile to ws[part]; splitk_finish_kernel sums them.  0 | 1 = off.
    int cs_split;
    // Round 5: the small-map gather kernel (conv_v9.hip).  Stride-2 input gradients run as FOUR parity phases in one launch: phase (ph, pw) owns the output pixels
    // (2 i + ph, 2 j + pw), whose taps all have one row / column parity -- only those taps' k-slabs are walked (1, 2, 2 and 4 of the 9 taps of a 3 x 3 filter instead of 9
    // slabs of which 5 .. 8 multiply zeros).  v9_phases = 0: plain launch; 4: the table below is valid.
    int v9_phases;
    int plan_v9_qt;     // != 0: launch_gather_v9 only fills the phase table, for pixel tiles of this size (the 8-wave kernel's phase launch)
    struct V9Phase {
        int tile0;          // first q-tile of this phase in the launch's q-tile order
        int Hq, Wq, Mq;     // the phase's pixel grid (rows ph, ph + 2, ... of Ho; columns pw, pw + 2, ... of Wo) and N * Hq * Wq
        int r0, s0;         // parity of the taps that reach this phase
        int nk;             // k-slabs of the phase: taps x (C / 64)
        FastDiv d_hw, d_w;
    } v9[4];
};

struct WgradArgs {
    const char* x;   // fwd input [N][H][W][ldx]
    const char* dy;  // [P][lddy]
    float* dw;       // [K][RSC]
    float* dbias;    // [K] or null: += column sums of dy (fused bias gradient)
    int N, H, W, C, ldx;
    int Ho, Wo, K, lddy;
    int R, S, stride, dil, pad_t, pad_l;
    int P, RSC;
    int tiles_p, tiles_q, iters_per_split;
    FastDiv div_howo, div_wo;
    int dbg;         // perf experiments only (odtk_debug_set key 2): bit0/1 zero-page DMA sources, bit2 no DMA after slab 0, bit4 no atomics
    int dbg2;        // dispatch A/B switches that leave results intact (key 6): bit 5 = narrow f32 layers on the legacy filter-gradient kernel
    unsigned x_bytes, dy_bytes;   // extents for the buffer-addressed DMA (8-wave kernel)
    // deterministic split-reduce (opt-in, odtk_debug_set key 5; 8-wave / four-wave kernels with > 1 pixel split): every block STORES its partial tile to
    // ws[split][K][RSC] (and its bias sums to bws[slot][K]); wgrad_reduce_kernel adds the splits in fixed order into dw / dbias.
    // null = float atomics straight into dw (the default; one split always)
    float* ws;
    float* bws;
    int nsplit, nbslot;
    int rev;         // tile order of the persistent conv1_x filter-gradient kernels: see GatherArgs::rev
    int which;       // set by launch_wgrad_v3: the kernel generation that ran (3 | 8), for odtk_conv_last_kernel
};

// 16 bytes of zeros that padded / out-of-range LDS-DMA lanes fetch instead of branching
static __device__ uint4 g_zero_page[4] = {};

// One LDS-DMA piece: 64 lanes x 16 B land at lds_addr + lane*16 (lds_addr wave-uniform, in an SGPR).
// Inline asm on purpose: hipcc would otherwise wait vmcnt(0) before the next ds_read of the OTHER
// stage (it cannot tell the stages apart) and serialise the pipeline; completion is waited for by
// explicit s_waitcnt vmcnt(N) in front of the slab barrier.  M0 is saved/restored around it.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
// Buffer-addressed LDS-DMA piece: 32-bit per-lane byte offset into a raw buffer; lanes whose offset is
// >= num_records (e.g. 0xFFFFFFF0 for padded / out-of-range taps) fetch ZEROS by the hardware range check,
// so no zero page, no 64-bit address arithmetic, no M0 save/restore (nothing else in these kernels uses M0).
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :
                 : "v"(voff), "s"(lds_addr), "s"(rsrc)
                 : "memory");
}
// same, without the compiler-level memory clobber: for kernels that pin the instruction order themselves
// (sched_barrier) and fence LDS visibility with explicit s_waitcnt vmcnt + s_barrier
__device__ __forceinline__ void glds16_buf_nc(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_addr), "s"(rsrc));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ uint2 lds_tr16(unsigned lds_byte_addr) {
    const v4i16_v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) v4i16_v*)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(uint2, v);
}

// four consecutive f32 output channels c .. c+3 of pixel row m of the x3 engine's single-part launches: bias, ReLU, the producer's ReLU mask (f32 rows:
// keep where mask > 0) and accumulation into what the row holds, then one 16-byte store.  (c < ldy; channels >= K hold zeros: their filter rows are masked)
__device__ __forceinline__ void x3_store4(const GatherArgs& a, int m, int c, float (&o)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (a.bias && c + e < a.K) o[e] += a.bias[c + e];
        if (a.relu) o[e] = fmaxf(o[e], 0.f);
    }
    float* dst = a.ws + (size_t)m * a.ldy + c;
    if (a.accumulate) {
        const float4 pv = *reinterpret_cast<const float4*>(dst);
        o[0] += pv.x; o[1] += pv.y; o[2] += pv.z; o[3] += pv.w;
    }
    if (a.mask) {                                           // (input-gradient semantics of the bf16 epilogue: the mask gates the SUM)
        const float4 mk = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.mask) + (size_t)m * a.ldmask + c);
        o[0] = mk.x > 0.f ? o[0] : 0.f; o[1] = mk.y > 0.f ? o[1] : 0.f; o[2] = mk.z > 0.f ? o[2] : 0.f; o[3] = mk.w > 0.f ? o[3] : 0.f;
    }
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
}

// acc + lo(v) + hi(v) of a packed bf16 pair in ONE VALU op (v_dot2c_f32_bf16 against (1, 1))
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16_ones(unsigned v, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, v), __builtin_bit_cast(bf16x2_v, 0x3f803f80u), acc, false);
}

// ---- bf16 pack helpers -----------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// hardware f32 -> packed bf16 (round-to-nearest-even, NaN quieted): one VALU op per two values
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// fused ReLU backward on packed bf16: keep u's halves where the matching half of m is > 0
// the same from the sign bits of the pair (bit 0 = low half, bit 1 = high half)
__device__ __forceinline__ unsigned keep_where_bits(unsigned u, unsigned b2) {
    const unsigned lo = (0u - (b2 & 1u)) & 0xffffu, hi = (0u - ((b2 >> 1) & 1u)) & 0xffff0000u;
    return u & (lo | hi);
}
// sign bits of a 16-byte chunk of bf16 values: bit e = element e > 0
__device__ __forceinline__ unsigned pos_bits(const uint4& v) {
    const unsigned* u = reinterpret_cast<const unsigned*>(&v);
    unsigned b = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        b |= ((int)(u[h] << 16) > 0 ? 1u : 0u) << (2 * h);
        b |= ((int)(u[h] & 0xffff0000u) > 0 ? 1u : 0u) << (2 * h + 1);
    }
    return b;
}
__device__ __forceinline__ unsigned keep_where_pos(unsigned u, unsigned m) {
    const unsigned lo = ((int)(m << 16) > 0) ? (u & 0xffffu) : 0u;
    const unsigned hi = ((int)(m & 0xffff0000u) > 0) ? (u & 0xffff0000u) : 0u;
    return lo | hi;
}
// epilogue post-ops on one 16-byte chunk (8 bf16) read back from the LDS image.  ReLU without
// accumulate was already applied in f32 before the image was written.
__device__ __forceinline__ void post_chunk(uint4& v, bool accumulate, bool relu, const uint4& old, bool has_mask,
                                           const uint4& mk) {
    unsigned* u = reinterpret_cast<unsigned*>(&v);
    if (accumulate) {
        const unsigned* uo = reinterpret_cast<const unsigned*>(&old);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float lo = bf16_lo(u[h]) + bf16_lo(uo[h]), hi = bf16_hi(u[h]) + bf16_hi(uo[h]);
            if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            u[h] = cvt_pk_bf16(lo, hi);
        }
    }
    if (has_mask) {
        const unsigned* um = reinterpret_cast<const unsigned*>(&mk);
#pragma unroll
        for (int h = 0; h < 4; ++h) u[h] = keep_where_pos(u[h], um[h]);
    }
}

// 8-wave / 3-stage LDS ring kernels (conv_v3.hip)
bool gather_v3_supported(const GatherArgs& a, int dtype, int out_dtype);
int launch_gather_v3(GatherArgs& a, hipStream_t st);
bool gather_c64_supported(const GatherArgs& a, int dtype, int out_dtype);   // resident-filter 64->64 3x3 kernel (with the fused pool)
int gather_v6_pool_variant(const GatherArgs& a, int dtype, int out_dtype);   // raster-run halo kernel with the fused 2x2 pool: 0 = not covered
int launch_gather_c64(GatherArgs& a, hipStream_t st);
bool gather_c8_supported(const GatherArgs& a, int dtype, int out_dtype);    // first-layer (3 -> 64) 3x3 kernel
int launch_gather_c8(GatherArgs& a, hipStream_t st);
bool wgrad_c64_supported(const WgradArgs& a, int dtype);                      // halo-patch 64->64 3x3 wgrad
bool wgrad_f32_narrow_supported(const WgradArgs& a, int dtype);               // f32, K / C <= 32 (3x3) | 64 (1x1): pixel-major LDS-DMA tiles, 32x32x2 MFMA
int launch_wgrad_f32_narrow(WgradArgs& a, hipStream_t st);
int launch_wgrad_c64(WgradArgs& a, hipStream_t st);
bool wgrad_c8_supported(const WgradArgs& a, int dtype);                       // first layer (3(8) -> 64) filter gradient: wave-private strips
int launch_wgrad_c8(WgradArgs& a, hipStream_t st);
bool wgrad_v3_supported(const WgradArgs& a, int dtype);
int launch_wgrad_v3(WgradArgs& a, hipStream_t st);
// x3: f32 convolutions on the bf16 MFMA kernels by operand splitting (conv_v3.hip)
void launch_split3_chan(const float* src, long long M, int C, int lds, void* dst, int ldc, int pattern, int nparts, int ldrow, hipStream_t st);
void launch_split3_rows(const float* src, long long M, int C, int lds, void* dst, int ldd, int pattern, hipStream_t st);
void launch_split3_chan2(const float* src0, long long M0, int C0, int lds0, void* dst0, int ldc0, int pattern0, int nparts0, int ldrow0,
                         const float* src1, long long M1, int C1, int lds1, void* dst1, int ldc1, int pattern1, int nparts1, int ldrow1, hipStream_t st);      // both in ONE launch
void launch_split3_rows2(const float* src0, long long M0, int C0, int lds0, void* dst0, int ldd0, int pattern0,
                         const float* src1, long long M1, int C1, int lds1, void* dst1, int ldd1, int pattern1, hipStream_t st);
int gather_x3_ksplit(const GatherArgs& a);
int launch_gather_x3(GatherArgs& a, float* out, float* partials, const float* bias, int relu, const float* mask, int ldmask, int accumulate, hipStream_t st);
// small-map gather kernel (conv_v9.hip): 64 x 64 tiles, four waves, deep LDS-DMA ring, no split-K; parity phases for stride-2 input gradients
bool gather_v9_wanted(const GatherArgs& a, int num_cu);
int launch_gather_v9(GatherArgs& a, hipStream_t st, int num_cu);
bool launch_gather_v9_x3(GatherArgs& a, hipStream_t st, int num_cu);      // conv_v9.hip: the small-map kernel with f32 output on the x3 engine's split operands (true = launched)
void set_x3_zero_lo(bool on);   // EXPERIMENT (odtk_debug_set key 6 bit 17): the x3 engine's splits write zeros for the low halves = the numerics of ONE bf16 product per f32 product
int misc_scratch(size_t bytes, hipStream_t st, char** out); // small partial-sum buffers of the box-side kernels, same per-(device, slot) arena rules
int x3_scratch(size_t bytes, hipStream_t st, char** out);   // the engine's own arena (per device and scratch slot), grown on demand; never moved once a captured graph holds it
// deterministic filter-gradient flush (odtk_debug_set key 5), shared by every filter-gradient kernel since round 6: wgrad_split_scratch points a.ws / a.bws at
// `splits` dW-shaped partial buffers (+ `bias_slots` x K bias partials) of the per-(device, slot) scratch when the mode is on and splits > 1 (null otherwise: the
// kernel adds into dW itself); wgrad_split_reduce launches the fixed-order sum of the partials into dW / dbias (nothing when a.ws is null)
int wgrad_split_scratch(WgradArgs& a, int splits, int bias_slots, hipStream_t st);
void wgrad_split_reduce(const WgradArgs& a, hipStream_t st);
bool get_wgrad_deterministic();        // (elementwise.hip: the scalar gamma gradient of the L2 norm follows the same switch)
void set_wgrad_deterministic(bool on);  // odtk_debug_set key 5: deterministic split-reduce instead of float atomics
int get_scratch_slot();                  // the calling thread's slot (0..3)
int set_scratch_slot(int slot);          // split-K partial buffers are per (device, slot); 0 on success

}  // namespace cv
}  // namespace odtk
