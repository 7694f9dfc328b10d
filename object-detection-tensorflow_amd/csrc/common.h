// Shared host/device helpers for libodtk (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/odtk.h"

namespace odtk {

void set_error(const char* fmt, ...);
void set_nms_legacy(bool on);   // boxes.hip: odtk_debug_set key 3
void set_gn_small_rows(int rows);   // elementwise.hip: odtk_debug_set key 7
void set_bn_small_rows(int rows); // elementwise.hip: odtk_debug_set key 4 (single-workgroup batch-norm up to this many rows; 0 = off)

#define ODTK_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            odtk::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                            hipGetErrorString(_e));                                  \
            return ODTK_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

#define ODTK_REQUIRE(cond, ...)                                                      \
    do {                                                                             \
        if (!(cond)) {                                                               \
            odtk::set_error(__VA_ARGS__);                                            \
            return ODTK_ERR_ARG;                                                     \
        }                                                                            \
    } while (0)

#define ODTK_LAUNCH_CHECK() ODTK_CHECK_HIP(hipGetLastError())

typedef unsigned short bf16_t;   // raw bfloat16 bits

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
// round-to-nearest-even; NaN kept quiet
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct elem;
template <> struct elem<bf16_t> {
    __host__ __device__ static float load(bf16_t v) { return bf16_to_f32(v); }
    __host__ __device__ static bf16_t store(float f) { return f32_to_bf16(f); }
};
template <> struct elem<float> {
    __host__ __device__ static float load(float v) { return v; }
    __host__ __device__ static float store(float f) { return f; }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t dtype_size(int dt) { return dt == ODTK_F32 ? 4 : 2; }

// Stream-ordered zero fill by a kernel of this library.  hipMemsetAsync is NOT used inside entry points: captured
// into a HIP graph (ROCm 7.2) its memset node was observed to replay with a garbage fill value (the SSD300 step's
// d(pred) came back as 0x48600000 / 0x70200000 patterns, depending on the host heap layout), api.hip.
int zero_async(void* p, size_t bytes, hipStream_t st);

}  // namespace odtk
