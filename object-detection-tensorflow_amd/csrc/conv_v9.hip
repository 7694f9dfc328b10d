// Small-map implicit-GEMM gather (round 5; gfx950 only, bf16 operands): forward convolutions and input gradients whose whole launch is a few GFLOP --
// the 10 x 10 ... 3 x 3 extras and heads of SSD300 (SSD300.py:304-313, :85-90), the 1 x 1 layers of DarkNet-53 at 13 x 13 / 26 x 26 and 8 images
// (YOLOv3.py:387-417), FCOS P5-P7 (FCOS.py:350-364).  On the 8-wave kernel (conv_v3.hip) these ran as 128 x 256 tiles with split-K: 4 .. 32 workgroups of 512 threads
// and 147 KiB of LDS, f32 partial tiles through HBM and a second launch to sum them -- 17 + 5 us for a layer of 0.02 GFLOP, every one of them on the step's critical chain.
// What bounds such a layer is latency, not arithmetic, so this kernel trades arithmetic intensity for parallelism and depth:
//   * tile = 64 channels x 64 pixels, four wave64 as 2 x 2, one 32 x 32 x 16 MFMA per wave and k sub-step: an 800-pixel layer is 13 x K / 64 workgroups with the
//     WHOLE reduction in each -- no split-K, no partial tiles, no second launch, results independent of the launch geometry;
//   * a k-slab is 64 k-elements (16 KiB for both operands: four LDS-DMA pieces per wave) in the swizzled 128-byte-row layout of the other kernels; NST stages,
//     NST - 1 slabs in flight behind a counted s_waitcnt vmcnt -- the k loop runs at the LDS-DMA rate of the CU, not at one memory round trip per slab;
//     64 KiB of LDS (NST = 4) and < 64 registers: two or three workgroups share a CU and fill each other's prologue / epilogue;
//   * past the last slab the ring keeps issuing out-of-range pieces (zero fill by the buffer range check) so that every wait count is a compile-time constant;
//   * stride-2 input gradients (idiv == 2) run as four PARITY PHASES in one launch: the output pixels (2 i + ph, 2 j + pw) are reached by the taps of one row and one
//     column parity only, so a phase walks 1, 2, 2 or 4 of the nine taps of a 3 x 3 filter -- 9 tap-slabs per four pixels instead of 36, and none of them multiplies
//     zeros (the 8-wave kernel masked the mismatching taps: conv8_2's input gradient did four times its useful MFMA work).
// Epilogue: the semantics of epilogue_bf16 (bias, ReLU, ReLU mask, accumulate; pad channels zero) on a 64 x 64 image in LDS, 16 bytes per lane.
#include "conv_common.h"

namespace odtk {
namespace cv {
namespace {

template <int N>
__device__ __forceinline__ void v9_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void v9_barrier() {
    asm volatile("s_barrier" ::: "memory");
}

typedef short s16x2v9 __attribute__((ext_vector_type(2)));

// C64: C % 64 == 0 and Kdim % 64 == 0 -- a k-slab never straddles a tap, the tap walk is wave-uniform.  PHASE (implies C64): parity phases of a stride-2 input gradient.
// F32OUT (round 6, implies C64, no PHASE): the x3 engine's form -- the pixel operand stores [hi | lo] per row and is read as [hi | hi | lo] (GatherArgs::x3c) against
// filters stored [hi | lo | hi]; the f32 accumulators leave as f32 rows straight from the registers (x3_store4: bias, ReLU, ReLU mask, accumulate).  What the small maps'
// 1 x 1 layers of DarkNet-53 run on since YOLOv3 trains on f32x3 by default (they took the 8-wave kernel's split-K partials + splitk_finish_f32_kernel before).
template <int NST, bool C64, bool PHASE, bool F32OUT = false>
__global__ void __launch_bounds__(256) conv_gather_v9_kernel(const GatherArgs a) {
    static_assert(!F32OUT || (C64 && !PHASE), "F32OUT: whole 64-channel chunks, no parity phases");
    constexpr int PT = 64, QT = 64;
    constexpr int STAGE = (PT + QT) * 128;             // 16 KiB: filter rows, then pixel rows
    constexpr int NDMA = 4;                            // LDS-DMA pieces per wave and slab
    constexpr int D = NST - 1;                         // slabs in flight
    static_assert(NST >= 2 && NST * STAGE >= QT * PT * 2, "ring holds the output image");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave & 1, wq = wave >> 1;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    int tq = lin / a.tiles_p;
    const int tp = lin - tq * a.tiles_p;
    // ---- phase of this tile (stride-2 input gradient): its pixel grid, tap parity and slab count
    int ph = 0, pw = 0, Hq = a.Ho, Wq = a.Wo, Mq = a.M, r_first = 0, s_first = 0, nk = (a.Kdim + 63) >> 6;
    FastDiv d_hw = a.div_howo, d_w = a.div_wo;
    if (PHASE) {
        const int pi = (tq >= a.v9[1].tile0) + (tq >= a.v9[2].tile0) + (tq >= a.v9[3].tile0);
        const GatherArgs::V9Phase& P = a.v9[pi];
        ph = pi >> 1; pw = pi & 1;
        tq -= P.tile0;
        Hq = P.Hq; Wq = P.Wq; Mq = P.Mq; r_first = P.r0; s_first = P.s0; nk = P.nk;
        d_hw = P.d_hw; d_w = P.d_w;
    }
    const int p0 = tp * PT, q0 = tq * QT;
    const int rstep = PHASE ? 2 : 1;

    const int r0 = tid >> 3;                           // DMA rows r0, r0 + 32 of both operands
    const int cc = (tid & 7) ^ swz_g(r0);              // logical 16-byte chunk this lane fetches (the swizzle is the same for rows r and r + 32)
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rw = make_rsrc(a.w, a.w_bytes);

    // ---- per pixel row: byte offset of tap (0, 0) channel 0 and the bit mask of in-range taps (the arithmetic of conv_gather_v3_kernel)
    unsigned qoff32[2], qmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = q0 + r0 + 32 * i;
        qoff32[i] = 0; qmask[i] = 0;
        if (ml < Mq) {
            const int n = (int)fdiv((unsigned)ml, d_hw), rem = ml - n * (Hq * Wq);
            const int hi_ = (int)fdiv((unsigned)rem, d_w), wi_ = rem - hi_ * Wq;
            const int ho = PHASE ? 2 * hi_ + ph : hi_, wo = PHASE ? 2 * wi_ + pw : wi_;
            const int hb = ho * a.ostride - a.pad_t, wb = wo * a.ostride - a.pad_l;
            // stride-2 input gradient: tap r reads dy row (hb + r) / 2 = (hb >> 1) + ((r + 1) >> 1) for every parity-matching r
            const int hq = a.idiv == 2 ? hb >> 1 : hb, wq2 = a.idiv == 2 ? wb >> 1 : wb;
            qoff32[i] = (unsigned)(((n * a.H + hq) * a.W + wq2) * a.ldx * 2);      // may wrap below 0: only used with in-range taps
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
    unsigned poff32[2];
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = p0 + r0 + 32 * i;
        pok[i] = row < a.K;
        poff32[i] = (unsigned)(row * a.ldw * 2);
    }
    if (C64) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { qoff32[i] += (unsigned)(cc * 16); poff32[i] += (unsigned)(cc * 16); }
    }
    // ---- tap walk.  C64: wave-uniform (cur_r, cur_s, cur_c), one tap per C / 64 slabs; else per lane (kr, ks, kc) of this lane's chunk
    int cur_r = r_first, cur_s = s_first, cur_c = 0, issued = 0;
    int klin = cc * 8, kc, ks, kr;
    {
        const int rs = klin / a.C;
        kc = klin - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const unsigned sP = smem_base + (unsigned)stage * STAGE + wave_u * 1024u;
        const unsigned sQ = sP + PT * 128;
        if (C64) {
            const bool live = issued < nk;
            const int sr = a.idiv == 2 ? (cur_r + 1) >> 1 : cur_r * a.dil, ss = a.idiv == 2 ? (cur_s + 1) >> 1 : cur_s * a.dil;
            const int xc = (F32OUT && a.x3c && cur_c >= a.x3c) ? cur_c - a.x3c : cur_c;       // x3: the third part re-reads the first part's channels
            const unsigned toff32 = (unsigned)((sr * a.W + ss) * a.ldx * 2 + xc * 2);
            const unsigned tapbit = live ? 1u << (cur_r * a.S + cur_s) : 0u;
            const unsigned woff = (unsigned)(((cur_r * a.S + cur_s) * a.C + cur_c) * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16_buf(rx, (qmask[i] & tapbit) ? qoff32[i] + toff32 : 0xFFFFFFF0u, sQ + i * 4096u);
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16_buf(rw, (live && pok[i]) ? poff32[i] + woff : 0xFFFFFFF0u, sP + i * 4096u);
            ++issued;
            cur_c += 64;
            if (cur_c >= a.C) {
                cur_c = 0;
                cur_s += rstep;
                if (cur_s >= a.S) { cur_s = s_first; cur_r += rstep; }
            }
            return;
        }
        const bool kv = kr < a.R;                      // (false past the end of the reduction: zero fill)
        const int tap = kr * a.S + ks;
        const int tr = a.idiv == 2 ? (kr + 1) >> 1 : kr * a.dil, ts = a.idiv == 2 ? (ks + 1) >> 1 : ks * a.dil;
        const unsigned toff32 = (unsigned)((tr * a.W + ts) * a.ldx * 2 + kc * 2);
        const unsigned tapbit = kv ? (1u << tap) : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_buf(rx, (qmask[i] & tapbit) ? qoff32[i] + toff32 : 0xFFFFFFF0u, sQ + i * 4096u);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_buf(rw, (kv && pok[i]) ? poff32[i] + (unsigned)(klin * 2) : 0xFFFFFFF0u, sP + i * 4096u);
        klin += 64;
        kc += 64;
        while (kc >= a.C) {
            kc -= a.C;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };

    f32x16_v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // ---- k loop: slabs kt .. kt + D - 1 are in flight at the top of iteration kt
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    const int l31 = lane & 31, hi = lane >> 5;
    const int prow = wp * 32 + l31, qrow = wq * 32 + l31;
    const unsigned pfo = (unsigned)(prow * 128), qfo = (unsigned)(PT * 128 + qrow * 128);
    const unsigned psw = (unsigned)swz_g(prow), qsw = (unsigned)swz_g(qrow);
    int st_c = 0, st_n = D;                            // stage consumed now / stage refilled now (NST = D + 1: the stage of slab kt - 1)
    for (int kt = 0; kt < nk; ++kt) {
        v9_wait_vmcnt<(D - 1) * NDMA>();               // this wave's pieces of slab kt landed
        v9_barrier();                                  // ... everybody's did, and everybody is past its reads of slab kt - 1
        issue(st_n);
        const char* sS = smem + st_c * STAGE;
        uint4 pf[4], qf[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const unsigned slot = (unsigned)(k4 * 2 + hi);
            pf[k4] = *reinterpret_cast<const uint4*>(sS + pfo + ((slot ^ psw) << 4));
            qf[k4] = *reinterpret_cast<const uint4*>(sS + qfo + ((slot ^ qsw) << 4));
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) Mma<bf16_t>::run(pf[k4], qf[k4], acc);
        st_c = st_c == NST - 1 ? 0 : st_c + 1;
        st_n = st_n == NST - 1 ? 0 : st_n + 1;
    }
    v9_wait_vmcnt<0>();                                // the trailing zero-fill pieces must land before the image overwrites the ring
    if constexpr (F32OUT) {
        const int m = q0 + qrow;
        if (m < Mq) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = p0 + wp * 32 + 8 * g + 4 * hi;
                if (c >= a.ldy) continue;
                float o[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                x3_store4(a, m, c, o);
            }
        }
        return;
    }
    v9_barrier();

    // ---- epilogue: acc (+ bias) -> bf16 (ReLU) -> swizzled [pixel][channel] image -> 16 bytes per lane (+ accumulate, ReLU mask) -> y
    {
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = p0 + wp * 32 + 8 * g + 4 * hi;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) {
                if (c < a.K) b.x = a.bias[c];
                if (c + 1 < a.K) b.y = a.bias[c + 1];
                if (c + 2 < a.K) b.z = a.bias[c + 2];
                if (c + 3 < a.K) b.w = a.bias[c + 3];
            }
            bv[g] = b;
        }
        const s16x2v9 floor16 = (a.relu && !a.accumulate) ? (s16x2v9)(0) : (s16x2v9)(-32768);
        char* rowp = smem + qrow * 128;
        const int qq = qrow & 7;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wp * 32 + 8 * g + 4 * hi;             // channel inside the tile
            uint2 o;
            o.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v9, cvt_pk_bf16(acc[4 * g] + bv[g].x, acc[4 * g + 1] + bv[g].y)), floor16));
            o.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v9, cvt_pk_bf16(acc[4 * g + 2] + bv[g].z, acc[4 * g + 3] + bv[g].w)), floor16));
            *reinterpret_cast<uint2*>(rowp + ((((cl >> 3) ^ qq) & 7) << 4) + ((cl & 4) << 1)) = o;
        }
    }
    __syncthreads();
    {
        const int qt = tid >> 3, ch = tid & 7;
        const int c0 = p0 + ch * 8;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = qt + 32 * it;
            const int ml = q0 + row;
            if (ml >= Mq || c0 >= a.ldy) continue;
            long long m = ml;
            if (PHASE) {
                const int n = (int)fdiv((unsigned)ml, d_hw), rem = ml - n * (Hq * Wq);
                const int hi_ = (int)fdiv((unsigned)rem, d_w), wi_ = rem - hi_ * Wq;
                m = ((long long)n * a.Ho + 2 * hi_ + ph) * a.Wo + 2 * wi_ + pw;
            }
            uint4 v = *reinterpret_cast<const uint4*>(smem + row * 128 + (((ch ^ row) & 7) << 4));
            char* yp = a.y + (m * a.ldy + c0) * 2;
            if (a.accumulate || a.mask) {
                uint4 old = make_uint4(0u, 0u, 0u, 0u), mk = make_uint4(0u, 0u, 0u, 0u);
                if (a.accumulate) old = *reinterpret_cast<const uint4*>(yp);
                if (a.mask) mk = *reinterpret_cast<const uint4*>(a.mask + (m * a.ldmask + c0) * 2);
                post_chunk(v, a.accumulate != 0, a.relu != 0, old, a.mask != nullptr, mk);
            }
            *reinterpret_cast<uint4*>(yp) = v;
        }
    }
}

}  // namespace

// Policy.  The kernel re-reads the pixel operand once per 64-channel tile column and the filter once per 64-pixel tile row through L2 -- fine while the launch is
// small, a loss against the 128 x 256 tiles beyond that: taken below ~8 GFLOP (measured per layer, profiles/r05*), and for every stride-2 input gradient on whole
// 64-channel chunks (the parity phases do a quarter of the 8-wave kernel's MFMA work at any size).  dbg2 (odtk_debug_set key 6): bit 6 = off, bit 7 = wherever supported.
bool gather_v9_wanted(const GatherArgs& a, int num_cu) {
    if (a.dbg2 & 64) return false;
    if (a.pool_mode || a.ybits || a.mask_bits || a.x3c || a.ws) return false;
    if (a.R * a.S > 32 || a.K < 1) return false;
    const bool phase = a.idiv == 2 && a.dil == 1 && a.ostride == 1 && a.C % 64 == 0 && a.Kdim == a.R * a.S * a.C;
    if (a.idiv == 2 && !phase) return false;
    if (a.dbg2 & 128) return true;
    if (phase) return true;
    // ... where the 8-wave kernel would split K: at most num_cu / 2 of its 128 x 256 tiles
    const int tiles256 = ceil_div(a.K, a.K <= 64 ? 64 : 128) * ceil_div(a.M, 256);
    const double flops = 2.0 * a.M * a.K * a.Kdim;
    return tiles256 <= num_cu / 2 && flops < 8.0e9;
}

// the x3 engine's launch (a.x / a.w = the split operands, a.C = 3 x the logical channels, a.ws = the f32 output): true = launched
bool launch_gather_v9_x3(GatherArgs& a, hipStream_t st, int num_cu) {
    if ((a.dbg2 & 64) || a.pool_mode || a.ybits || a.mask_bits || a.idiv != 1 || a.R * a.S > 32 || a.K < 1) return false;
    if (!(a.C % 64 == 0 && a.x3c % 64 == 0 && a.Kdim == a.R * a.S * a.C)) return false;
    const int tiles256 = ceil_div(a.K, a.K <= 64 ? 64 : 128) * ceil_div(a.M, 256);
    const double flops = 2.0 * a.M * a.K * a.Kdim;
    // (long reductions stay on the split-K form: RetinaNet's 3 x 3 / 256-channel heads on P5-P7 -- 108 slabs per tile -- measured 0.5 % of the step SLOWER here,
    //  DarkNet-53's 1 x 1 layers -- 12-48 slabs -- 2.5 % faster: gpurun r6n)
    if (!(a.dbg2 & 128) && !(tiles256 <= num_cu / 2 && flops < 8.0e9 && a.Kdim <= 64 * 64)) return false;
    a.tiles_p = ceil_div(a.K, 64);
    a.tiles_q = ceil_div(a.M, 64);
    a.v9_phases = 0;
    a.ksplit = -9;
    const int grid = a.tiles_p * a.tiles_q;
    if (grid <= 0) return true;
    if (grid <= num_cu && !(a.dbg2 & 512)) hipLaunchKernelGGL((conv_gather_v9_kernel<8, true, false, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_gather_v9_kernel<4, true, false, true>), dim3(grid), dim3(256), 0, st, a);
    return true;
}

int launch_gather_v9(GatherArgs& a, hipStream_t st, int num_cu) {
    a.tiles_p = ceil_div(a.K, 64);
    const bool phase = a.idiv == 2;
    if (phase) {
        int t0 = 0;
        const int ncs = a.C / 64;
        for (int pi = 0; pi < 4; ++pi) {
            const int ph = pi >> 1, pw = pi & 1;
            GatherArgs::V9Phase& P = a.v9[pi];
            P.Hq = a.Ho > ph ? (a.Ho - ph + 1) / 2 : 0;
            P.Wq = a.Wo > pw ? (a.Wo - pw + 1) / 2 : 0;
            P.Mq = a.N * P.Hq * P.Wq;
            // tap r reaches output row h when (h - pad_t + r) is even
            P.r0 = (ph + a.pad_t) & 1;
            P.s0 = (pw + a.pad_l) & 1;
            const int nr = a.R > P.r0 ? (a.R - P.r0 + 1) / 2 : 0, ns = a.S > P.s0 ? (a.S - P.s0 + 1) / 2 : 0;
            P.nk = nr * ns * ncs;
            P.tile0 = t0;
            P.d_hw = make_fastdiv((unsigned)(P.Hq * P.Wq > 0 ? P.Hq * P.Wq : 1));
            P.d_w = make_fastdiv((unsigned)(P.Wq > 0 ? P.Wq : 1));
            t0 += ceil_div(P.Mq, a.plan_v9_qt ? a.plan_v9_qt : 64);
        }
        a.v9_phases = 4;
        a.tiles_q = t0;
        if (a.plan_v9_qt) return 0;                          // the 8-wave kernel runs the phases on its own tiles (launch_gather_v3): the table only
    } else {
        a.v9_phases = 0;
        a.tiles_q = ceil_div(a.M, 64);
    }
    a.ksplit = -9;                                       // tells the dispatcher which kernel ran (odtk_conv_last_kernel)
    const int grid = a.tiles_p * a.tiles_q;
    if (grid <= 0) return 0;
    const bool c64 = a.C % 64 == 0 && a.Kdim % 64 == 0 && a.Kdim == a.R * a.S * a.C;
    // Ring depth.  A workgroup's k loop runs at (bytes in flight) / (LDS-DMA latency, ~1.1 us): three slabs = 48 KiB in flight measured 41 GB/s per workgroup
    // (pred4 forward, 36 slabs: 18 us).  With at most one workgroup per CU anyway, the ring takes 128 KiB (seven slabs in flight); with more, 64 KiB so that two
    // workgroups share a CU.  dbg2 bit 9 = always the four-stage ring (A/B).
    const bool deep = grid <= num_cu && !(a.dbg2 & 512);
#define ODTK_V9(NST_) \
    do { \
        if (phase) hipLaunchKernelGGL((conv_gather_v9_kernel<NST_, true, true>), dim3(grid), dim3(256), 0, st, a); \
        else if (c64) hipLaunchKernelGGL((conv_gather_v9_kernel<NST_, true, false>), dim3(grid), dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((conv_gather_v9_kernel<NST_, false, false>), dim3(grid), dim3(256), 0, st, a); \
    } while (0)
    if (deep) ODTK_V9(8); else ODTK_V9(4);
#undef ODTK_V9
    return 0;
}

}  // namespace cv
}  // namespace odtk
