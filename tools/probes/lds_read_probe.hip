// LDS read-rate probe (gfx950): how fast do ds_read_b64_tr_b16 / ds_read_b64 / ds_read_b128 stream with the lane address
// patterns the conv kernels use?  4 waves per workgroup, one workgroup per CU, 8 independent reads per iteration.
// Prints bytes / ns / CU (128 B/clk at 2.4 GHz = 307 B/ns).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) rd(const int* lane_off, int iters, unsigned* out) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + lane_off[threadIdx.x & 63] + ((threadIdx.x >> 6) & 3) * 8192;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            uint2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[k]) : "v"(a), "n"(k * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y;
        } else if (MODE == 1) {
            uint2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"(a), "n"(k * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y;
        } else {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(a), "n"(k * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].w;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <int MODE, int NW = 4>
void run(const int* d_off, unsigned* out, const char* label, int bytes_per_lane) {
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<MODE, NW>), dim3(grid), dim3(NW * 64), 0, 0, d_off, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)iters * 8 * NW * 64 * bytes_per_lane;
    printf("%-44s waves %2d %8.3f ms  %7.1f B/ns/CU  %6.2f ns per wave-read per CU\n", label, NW, best, bytes / (best * 1e6), best * 1e6 / (iters * 8.0 * NW));
}
int main() {
    int h[64]; int* d; unsigned* out;
    hipMalloc(&d, sizeof(h)); hipMalloc(&out, 1024);
    auto up = [&]() { hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); };
    // A: lane-linear 8 B
    for (int l = 0; l < 64; ++l) h[l] = l * 8; up();
    run<0>(d, out, "tr_b64  lane-linear 8 B", 8);
    run<0, 8>(d, out, "tr_b64  lane-linear 8 B", 8);
    run<0, 16>(d, out, "tr_b64  lane-linear 8 B", 8);
    run<1>(d, out, "b64     lane-linear 8 B", 8);
    run<1, 8>(d, out, "b64     lane-linear 8 B", 8);
    run<1, 16>(d, out, "b64     lane-linear 8 B", 8);
    for (int l = 0; l < 64; ++l) h[l] = l * 16; up();
    run<2>(d, out, "b128    lane-linear 16 B", 16);
    run<2, 8>(d, out, "b128    lane-linear 16 B", 16);
    run<2, 16>(d, out, "b128    lane-linear 16 B", 16);
    // B: wgrad halo kernel pattern: pixel rows of 128 B, lane -> pixel 8*hi + rr, 32-channel block, swizzled chunk
    for (int m = 0; m < 4; ++m) {
        for (int l = 0; l < 64; ++l) {
            const int hi = l >> 5, g = l >> 4, c16 = l & 15, rr = c16 >> 2;
            const int sub8 = (16 * (g & 1)) * 2 + (c16 & 3) * 8;
            const int byte = sub8, px = m + 8 * hi + rr;
            h[l] = px * 128 + (((byte >> 4) ^ (((px >> 1) & 1) << 2)) << 4) + (byte & 8);
        }
        up();
        char lab[64]; snprintf(lab, 64, "tr_b64  halo pattern, start pixel %d", m);
        run<0>(d, out, lab, 8);
    }
    // C: same without swizzle
    for (int l = 0; l < 64; ++l) {
        const int hi = l >> 5, g = l >> 4, c16 = l & 15, rr = c16 >> 2;
        h[l] = (8 * hi + rr) * 128 + (16 * (g & 1)) * 2 + (c16 & 3) * 8;
    }
    up();
    run<0>(d, out, "tr_b64  halo pattern, no swizzle", 8);
    // D: wgrad v3 pattern (256-B pixel rows, 4-pixel pieces of 1 KiB)
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, c = l & 15, rr = c >> 2;
        const int ch = (16 * (g & 1)) / 8 + ((c & 3) >> 1);
        h[l] = (2 * (g >> 1)) * 1024 + (rr * 16 + (ch ^ (rr << 2))) * 16 + (c & 1) * 8;
    }
    up();
    run<0>(d, out, "tr_b64  wgrad-v3 pattern", 8);
    // E: b128 conv fragment pattern: lane l31 -> row (128 B), hi -> 16-B slot, swizzle (row & 7)
    for (int l = 0; l < 64; ++l) { const int r = l & 31, hi = l >> 5; h[l] = r * 128 + ((hi ^ (r & 7)) << 4); }
    up();
    run<2>(d, out, "b128    fragment pattern (row swizzle)", 16);
    return 0;
}
