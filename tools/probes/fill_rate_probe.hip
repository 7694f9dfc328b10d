// Probe: per-CU operand fill rate from L2 into LDS.
//   V0  LDS-DMA (global_load_lds_dwordx4), NW waves, each keeps DEPTH pieces in flight
//   V1  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)
//   V2  global_load_dwordx4 -> VGPR only (no LDS write)
// Source: a WS-byte window per block re-read many times (L2 resident), 128-B rows like the conv
// slabs (8 lanes x 16 B per row, rows 'stride' bytes apart).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I object-detection-tensorflow_amd/csrc tools/probes/fill_rate_probe.hip -o tools/probes/bin/fill_rate_probe
#include "conv_common.h"
#include <vector>
using namespace odtk;
using namespace odtk::cv;

template <int V, int NW, int DEPTH>
__global__ void __launch_bounds__(NW * 64) fill(const char* src, size_t win, int stride, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const char* base = src + (size_t)blockIdx.x * win;
    // piece = 8 rows x 128 B; lane -> (row = lane>>3, chunk = lane&7)
    const size_t lane_off = (size_t)(lane >> 3) * stride + (lane & 7) * 16;
    const size_t rows_in_win = win / stride;
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t row = wave * 8;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                glds16(base + row * stride + lane_off, smem_base + ((wave * DEPTH + d) & 63) * 1024u);
                row += NW * 8;
                if (row + 8 > rows_in_win) row = wave * 8;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                v[d] = *reinterpret_cast<const uint4*>(base + row * stride + lane_off);
                row += NW * 8;
                if (row + 8 > rows_in_win) row = wave * 8;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (V == 1) *reinterpret_cast<uint4*>(smem + ((wave * DEPTH + d) & 63) * 1024 + lane * 16) = v[d];
                else { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
            }
        }
    }
    __syncthreads();
    const uint4 r = *reinterpret_cast<const uint4*>(smem + tid * 16);
    out[blockIdx.x * NW * 64 + tid] = __uint_as_float(r.x ^ acc.x ^ acc.y ^ acc.z ^ acc.w);
}

template <int V, int NW, int DEPTH>
void run(const char* src, size_t win, int stride, float* out, const char* label) {
    const int iters = 4096 / DEPTH, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<V, NW, DEPTH>), dim3(grid), dim3(NW * 64), 0, 0, src, win, stride, iters, out);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fill<V, NW, DEPTH>), dim3(grid), dim3(NW * 64), 0, 0, src, win, stride, iters, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes_per_cu = (double)iters * DEPTH * NW * 1024.0;
    printf("%-34s V%d waves %2d depth %2d win %6zu KB stride %5d : %7.3f ms  %7.1f GB/s/CU  %6.1f TB/s chip  (%s)\n", label, V, NW,
           DEPTH, win >> 10, stride, best, bytes_per_cu / (best * 1e-3) / 1e9, bytes_per_cu * grid / (best * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const size_t total = 256ull * (1 << 20);
    char* src; float* out;
    (void)hipMalloc(&src, total);
    (void)hipMemset(src, 1, total);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    for (size_t win : {size_t(64) << 10, size_t(1) << 20}) {
        for (int stride : {128, 1024}) {
            run<0, 4, 4>(src, win, stride, out, "LDS-DMA 4 waves x4");
            run<0, 8, 6>(src, win, stride, out, "LDS-DMA 8 waves x6");
            run<0, 8, 12>(src, win, stride, out, "LDS-DMA 8 waves x12");
            run<0, 16, 8>(src, win, stride, out, "LDS-DMA 16 waves x8");
            run<1, 8, 6>(src, win, stride, out, "reg-staged 8 waves x6");
            run<1, 16, 8>(src, win, stride, out, "reg-staged 16 waves x8");
            run<2, 8, 6>(src, win, stride, out, "load-only 8 waves x6");
            run<2, 16, 8>(src, win, stride, out, "load-only 16 waves x8");
        }
    }
    return 0;
}
