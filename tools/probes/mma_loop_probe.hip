// Probe: what bounds the slab loop (ds_read_b128 fragments -> 32x32x16 bf16 MFMA -> barrier)
// of the conv kernels, without any global traffic?  Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I object-detection-tensorflow_amd/csrc \
//         tools/probes/mma_loop_probe.hip -o tools/probes/bin/mma_loop_probe
#include "conv_common.h"
#include <vector>
#include <cstdlib>

using namespace odtk;
using namespace odtk::cv;

// V: 0 baseline (mma_slab), 1 no barrier, 2 no ds_read in the loop, 3 fragment double buffering,
//    4 baseline + setprio, 5 double buffering + setprio, 6 double buffering, no barrier
template <int V, int PI, int QI, int WP, int WQ>
__global__ void __launch_bounds__(WP* WQ * 64) probe(const uint4* src, float* out, int nslab) {
    constexpr int PT = WP * PI * 32, QT = WQ * QI * 32;
    constexpr int STAGE = (PT + QT) * 128;
    constexpr int NTHR = WP * WQ * 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wq = wave / WP;
    for (int i = tid; i < 2 * STAGE / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = src[i & 4095];
    __syncthreads();
    f32x16_v acc[PI][QI];
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    const int prow0 = wp * PI * 32, qrow0 = wq * QI * 32;
    uint4 pf0[PI], qf0[QI], pf1[PI], qf1[QI];
    auto load = [&](const char* sP, const char* sQ, int ks, uint4 (&pf)[PI], uint4 (&qf)[QI]) {
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
    auto mm = [&](uint4 (&pf)[PI], uint4 (&qf)[QI]) {
        if (V == 4 || V == 5) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int j = 0; j < QI; ++j) Mma<bf16_t>::run(pf[i], qf[j], acc[i][j]);
        if (V == 4 || V == 5) __builtin_amdgcn_s_setprio(0);
    };
    if (V == 2) load(smem, smem + PT * 128, 0, pf0, qf0);
    for (int kt = 0; kt < nslab; ++kt) {
        if (V != 1 && V != 6) asm volatile("s_barrier" ::: "memory");
        const char* sP = smem + (kt & 1) * STAGE;
        const char* sQ = sP + PT * 128;
        if (V == 0 || V == 1 || V == 4) {
            mma_slab<bf16_t, PI, QI, true>(sP, sQ, prow0, qrow0, lane, acc);
        } else if (V == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mm(pf0, qf0);
                asm volatile("" : "+v"(pf0[0].x));
            }
        } else {
            load(sP, sQ, 0, pf0, qf0);
            load(sP, sQ, 1, pf1, qf1);
            mm(pf0, qf0);
            load(sP, sQ, 2, pf0, qf0);
            mm(pf1, qf1);
            load(sP, sQ, 3, pf1, qf1);
            mm(pf0, qf0);
            mm(pf1, qf1);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < QI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * NTHR + tid] = s;
}

template <int V, int PI, int QI, int WP, int WQ>
void run(const uint4* src, float* out, int blocks_per_cu, const char* label) {
    constexpr int PT = WP * PI * 32, QT = WQ * QI * 32;
    const int nslab = 4096, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<V, PI, QI, WP, WQ>), dim3(grid), dim3(WP * WQ * 64), 0, 0, src, out, nslab);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<V, PI, QI, WP, WQ>), dim3(grid), dim3(WP * WQ * 64), 0, 0, src, out, nslab);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * PT * QT * 64.0 * nslab * grid;
    printf("%-46s V%d tile %3dx%3d waves %dx%d wavetile %3dx%3d blocks/CU %d : %8.3f ms %8.1f TF  (%s)\n", label, V, PT, QT,
           WP, WQ, PI * 32, QI * 32, blocks_per_cu, best, fl / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff)) | (unsigned short)((rand() & 1) << 15);   // ~ +-[0.5, 2)
    uint4* src; float* out;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    run<0, 2, 2, 2, 4>(src, out, 1, "baseline 8 waves 128x256");
    run<1, 2, 2, 2, 4>(src, out, 1, "no barrier");
    run<2, 2, 2, 2, 4>(src, out, 1, "no ds_read");
    run<3, 2, 2, 2, 4>(src, out, 1, "frag double buffer");
    run<4, 2, 2, 2, 4>(src, out, 1, "baseline + setprio");
    run<5, 2, 2, 2, 4>(src, out, 1, "frag double buffer + setprio");
    run<6, 2, 2, 2, 4>(src, out, 1, "frag double buffer, no barrier");
    run<0, 2, 2, 2, 2>(src, out, 1, "4 waves 128x128, 1 block/CU");
    run<0, 2, 2, 2, 2>(src, out, 2, "4 waves 128x128, 2 blocks/CU");
    run<3, 2, 2, 2, 2>(src, out, 2, "4 waves 128x128, 2 blocks/CU, dbuf");
    run<0, 4, 2, 2, 4>(src, out, 1, "8 waves 256x256 (wave 128x64)");
    run<3, 4, 2, 2, 4>(src, out, 1, "8 waves 256x256 (wave 128x64) dbuf");
    run<2, 4, 2, 2, 4>(src, out, 1, "8 waves 256x256 no ds_read");
    run<0, 2, 4, 2, 2>(src, out, 1, "4 waves 128x256 (wave 64x128)");
    run<3, 2, 4, 2, 2>(src, out, 1, "4 waves 128x256 (wave 64x128) dbuf");
    run<3, 4, 2, 1, 4>(src, out, 1, "4 waves 128x256 (wave 128x64) dbuf");
    run<3, 4, 4, 1, 4>(src, out, 1, "4 waves 128x512 (wave 128x128) dbuf");
    run<3, 4, 4, 2, 2>(src, out, 1, "4 waves 256x256 (wave 128x128) dbuf");
    run<0, 4, 4, 2, 2>(src, out, 1, "4 waves 256x256 (wave 128x128)");
    return 0;
}
