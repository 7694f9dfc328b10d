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
#include <stdlib.h>
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
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "no error (CPU emulation)"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 2; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
template <typename F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }

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
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { const unsigned o = *p; if (v > o) *p = v; return o; }
inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o | v; return o; }
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
// dynamic shared memory (the 4th launch argument): the backend rewrites `extern __shared__ T name[];` into `T* name = (T*)hipcpu::dynamic_smem();`
inline std::vector<unsigned long long>& dyn_buf() { static std::vector<unsigned long long> v; return v; }
inline long long& launch_counter() { static long long n = 0; return n; }      // kernel launches so far (tests assert launch counts with it)
inline void* dynamic_smem() { return dyn_buf().data(); }
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
        if (f.stack.empty()) f.stack.resize(192 * 1024);      // (512 threads x 192 KiB = 96 MiB at most, shared by all launches)
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
inline Block& the_block() { static Block b; return b; }
template <typename F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F&& f) {
    Block& b = the_block();                           // ONE set of fibers (and stacks) for every launch site: launches never overlap
    launch_counter()++;
    b.dim = block; b.grid = grid;
    if (dyn_buf().size() * 8 < shmem + 16) dyn_buf().resize(shmem / 8 + 2);
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

inline void __threadfence() {}          // one OS thread, workgroups one after the other: every earlier store is visible
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

template <typename T>
inline T __shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const unsigned long long* all = hipcpu::exchange(bits);
    T out;
    memcpy(&out, &all[(unsigned)src_lane & 63u], sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_down(T v, unsigned delta) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const unsigned long long* all = hipcpu::exchange(bits);
    const unsigned lane = hipcpu::flat_tid() % 64, src = lane + delta;
    T out;
    memcpy(&out, &all[src < 64 ? src : lane], sizeof(T));
    return out;
}
inline int __any(int pred) { return __ballot(pred) != 0ull; }
inline int __all(int pred) { return __ballot(pred) == ~0ull; }
// compile-only stand-ins for csrc/boxes.hip (its NMS kernels are NOT run under the emulation: they rely on the implicit lock-step of a wave between fences)
inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
template <typename T, typename V> inline T hipcpu_fetch_add(T* p, V v) { const T o = *p; *p = o + (T)v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) hipcpu_fetch_add((p), (v))

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipcpu::launch((grid), (block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); })
