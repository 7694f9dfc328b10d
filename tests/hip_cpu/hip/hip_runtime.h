// TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> that lets the NON-MFMA kernels of libodtk (csrc/lhrcnn.hip, csrc/augment.hip) be compiled
// with g++ and executed on the CPU from the same source, so that the CPU tier (`pytest -m "not gpu"`) checks the kernel source against the oracles without a
// GPU (tests/test_hip_cpu.py).  Execution model: a launch runs its workgroups one after the other; the threads of a workgroup are ucontext fibers on one OS
// thread, switched only at __syncthreads / __shfl_xor / __ballot (64-lane waves, lane = threadIdx.x % 64), which is exactly the lock-step a wavefront
// guarantees for those calls.  Not modelled: anything that depends on real concurrency between workgroups (none of these kernels does), LDS banking, timing.
// float atomics are plain adds (one OS thread).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <ucontext.h>
#include <algorithm>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "no error (CPU emulation)"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static

using std::max;
using std::min;
inline float __fmul_rn(float a, float b) { return a * b; }      // (built with -ffp-contract=off: no fused multiply-add, as in the device build)
inline float __fadd_rn(float a, float b) { return a + b; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }

namespace hipcpu {

struct Fiber {
    ucontext_t uc;
    dim3 thread;
    bool done = false;
    std::vector<char> stack;
};
struct Wave {
    unsigned long long slot[2][64];
    int count = 0, gen = 0;
};
struct Block {
    dim3 idx, dim, grid;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int arrived = 0, gen = 0, alive = 0;
    ucontext_t sched;
    Fiber* cur = nullptr;
    std::function<void()> body;
};
inline Block*& current() { static Block* b = nullptr; return b; }
inline void yield() { Block* b = current(); swapcontext(&b->cur->uc, &b->sched); }
inline void trampoline() {
    Block* b = current();
    b->body();
    b->cur->done = true;
    b->alive--;
    swapcontext(&b->cur->uc, &b->sched);
}
inline void run_block(Block& b) {
    const unsigned n = b.dim.x * b.dim.y * b.dim.z;
    b.fibers.resize(n);
    b.waves.assign((n + 63) / 64, Wave());
    b.arrived = 0; b.gen = 0; b.alive = (int)n;
    for (unsigned t = 0; t < n; ++t) {
        Fiber& f = b.fibers[t];
        f.done = false;
        f.thread = dim3(t % b.dim.x, (t / b.dim.x) % b.dim.y, t / (b.dim.x * b.dim.y));
        if (f.stack.empty()) f.stack.resize(256 * 1024);
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack.data();
        f.uc.uc_stack.ss_size = f.stack.size();
        f.uc.uc_link = &b.sched;
        makecontext(&f.uc, (void (*)())trampoline, 0);
    }
    current() = &b;
    while (b.alive > 0) {
        for (unsigned t = 0; t < n; ++t) {
            if (b.fibers[t].done) continue;
            b.cur = &b.fibers[t];
            swapcontext(&b.sched, &b.cur->uc);
        }
    }
    current() = nullptr;
}
template <typename F>
inline void launch(dim3 grid, dim3 block, F&& f) {
    static Block b;                                   // fibers (and their stacks) are reused from launch to launch
    b.dim = block; b.grid = grid;
    b.body = std::function<void()>(f);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                b.idx = dim3(x, y, z);
                run_block(b);
            }
}
inline unsigned flat_tid() { Block* b = current(); const dim3& t = b->cur->thread; return t.x + b->dim.x * (t.y + b->dim.y * t.z); }
// wave-wide exchange: every lane deposits 8 bytes, waits until all 64 lanes of its wave have, then reads; double-buffered by generation so that a lane which
// runs ahead into the NEXT exchange cannot overwrite what a slower lane still has to read
inline const unsigned long long* exchange(unsigned long long mine) {
    Block* b = current();
    const unsigned tid = flat_tid();
    Wave& w = b->waves[tid / 64];
    const int gen = w.gen;
    w.slot[gen & 1][tid % 64] = mine;
    if (++w.count == 64) { w.count = 0; w.gen++; }
    else while (w.gen == gen) yield();
    return w.slot[gen & 1];
}
}  // namespace hipcpu

#define blockIdx (hipcpu::current()->idx)
#define blockDim (hipcpu::current()->dim)
#define gridDim (hipcpu::current()->grid)
#define threadIdx (hipcpu::current()->cur->thread)

inline void __syncthreads() {
    hipcpu::Block* b = hipcpu::current();
    const int gen = b->gen;
    const int n = (int)(b->dim.x * b->dim.y * b->dim.z);
    if (++b->arrived == n) { b->arrived = 0; b->gen++; }
    else while (b->gen == gen) hipcpu::yield();
}
template <typename T>
inline T __shfl_xor(T v, int lane_mask) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const unsigned long long* all = hipcpu::exchange(bits);
    T out;
    memcpy(&out, &all[(hipcpu::flat_tid() % 64) ^ (unsigned)lane_mask], sizeof(T));
    return out;
}
inline unsigned long long __ballot(int pred) {
    const unsigned long long* all = hipcpu::exchange(pred ? 1ull : 0ull);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (all[l] & 1ull) << l;
    return m;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipcpu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
