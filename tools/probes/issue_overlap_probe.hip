// Probe (gfx950): which instructions of ONE wave overlap with that wave's own MFMA execution?
// One wave per SIMD (256 threads / CU).  Loop body = 4 x { v_mfma_f32_32x32x16_bf16 (independent accumulators) ; K fillers }.
// Filler kinds: 0 v_add_f32 (independent), 1 ds_read_b64_tr_b16, 2 ds_read_b128, 3 buffer_load_dwordx4 ... lds (1 KiB piece),
//               4 s_add_u32, 5 v_cndmask+v_add pair (address-style VALU)
// Prints ns per MFMA; the MFMA-only row is the floor.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K, int NW, bool STREAM>
__global__ void __launch_bounds__(NW * 64) probe(const char* src, unsigned nbytes, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned la = lbase + (wave & 3) * 8192 + lane * 16;
    const unsigned ldma = __builtin_amdgcn_readfirstlane(lbase + 32768 + (wave & 3) * 4096);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, nbytes, 0x00020000);
    unsigned voff = (blockIdx.x * 4 + (wave & 3)) * 4096 + lane * 16;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = 0x3f80; b[e] = 0x3f80; }
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    uint2 r2[8]; uint4 r4[8];
    unsigned sacc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[k & 7]));
                if (KIND == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r2[k & 7]) : "v"(la), "n"((k & 7) * 1024));
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[k & 7]) : "v"(la), "n"((k & 7) * 1024));
                if (KIND == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(ldma), "s"(rs));
                if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if ((KIND == 6 && j == 0) || (KIND == 7 && (j & 1) == 0) || (KIND == 8 && j == 0 && (it & 1) == 0))
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(ldma), "s"(rs));
                if (KIND == 5) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(voff) : "v"(lane) : "vcc");
            }
        }
        if (KIND == 1 || KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (KIND >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); if (STREAM) { voff += 256u * 4u * 4096u; if (voff >= nbytes - 4096u) voff -= (nbytes - 4096u) & ~0xFFFu; } }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + (float)sacc;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (KIND == 1) for (int k = 0; k < 8; ++k) s += (float)r2[k].x;
    if (KIND == 2) for (int k = 0; k < 8; ++k) s += (float)r4[k].x;
    out[blockIdx.x * NW * 64 + threadIdx.x] = s + (float)voff;
}
template <int KIND, int K, int NW = 4, bool STREAM = false>
void run(const char* src, unsigned nbytes, float* out, const char* label) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, K, NW, STREAM>), dim3(grid), dim3(NW * 64), 0, 0, src, nbytes, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double per = best * 1e6 / (iters * 4.0) / (NW / 4);
    printf("%-28s K=%d waves/CU %2d : %7.3f ms  %6.1f ns per MFMA slot (per SIMD)   (%s)\n", label, K, NW, best, per, hipGetErrorString(hipGetLastError()));
}
int main() {
    char* src; float* out; const unsigned nbytes = 1u << 30;
    hipMalloc(&src, nbytes); hipMemset(src, 0, nbytes); hipMalloc(&out, 256 * 1024 * 4);
    run<0, 0>(src, nbytes, out, "MFMA only");
    run<0, 0, 8>(src, nbytes, out, "MFMA only");
    run<0, 2>(src, nbytes, out, "v_add_f32"); run<0, 4>(src, nbytes, out, "v_add_f32"); run<0, 6>(src, nbytes, out, "v_add_f32"); run<0, 8>(src, nbytes, out, "v_add_f32");
    run<4, 4>(src, nbytes, out, "s_add_u32"); run<4, 8>(src, nbytes, out, "s_add_u32");
    run<5, 1>(src, nbytes, out, "cmp+cndmask+add"); run<5, 2>(src, nbytes, out, "cmp+cndmask+add");
    run<1, 1>(src, nbytes, out, "ds_read_b64_tr_b16"); run<1, 2>(src, nbytes, out, "ds_read_b64_tr_b16"); run<1, 4>(src, nbytes, out, "ds_read_b64_tr_b16");
    run<2, 1>(src, nbytes, out, "ds_read_b128"); run<2, 2>(src, nbytes, out, "ds_read_b128");
    run<3, 1>(src, nbytes, out, "buffer_load..lds 1KiB"); run<3, 2>(src, nbytes, out, "buffer_load..lds 1KiB");
    run<1, 1, 8>(src, nbytes, out, "ds_read_b64_tr_b16"); run<1, 2, 8>(src, nbytes, out, "ds_read_b64_tr_b16");
    run<3, 1, 8>(src, nbytes, out, "buffer_load..lds 1KiB");
    run<6, 1, 4>(src, nbytes, out, "lds-dma 1 per 4 MFMA (hot)"); run<6, 1, 8>(src, nbytes, out, "lds-dma 1 per 4 MFMA (hot)");
    run<7, 1, 4>(src, nbytes, out, "lds-dma 1 per 2 MFMA (hot)"); run<7, 1, 8>(src, nbytes, out, "lds-dma 1 per 2 MFMA (hot)");
    run<8, 1, 8>(src, nbytes, out, "lds-dma 1 per 8 MFMA (hot)");
    run<6, 1, 4, true>(src, nbytes, out, "lds-dma 1 per 4 MFMA (stream)"); run<6, 1, 8, true>(src, nbytes, out, "lds-dma 1 per 4 MFMA (stream)");
    run<7, 1, 8, true>(src, nbytes, out, "lds-dma 1 per 2 MFMA (stream)");
    run<8, 1, 8, true>(src, nbytes, out, "lds-dma 1 per 8 MFMA (stream)");
    return 0;
}
