// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds 16-bit ids; each lane supplies an
// address; prints which LDS element ids each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const int* addr_in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + addr_in[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 4; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 8;                       // lane-linear 8-byte pieces
            if (pat == 1) h_addr[l] = 0;                           // all the same address
            if (pat == 2) h_addr[l] = (l & 15) * 8 + (l >> 4) * 1024;   // groups 1 KB apart
            if (pat == 3) h_addr[l] = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;  // 4 rows x 256 B pitch
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %4d ->", l, h_addr[l]);
            for (int j = 0; j < 4; ++j) printf(" %4d", h_out[l * 4 + j]);
            printf("\n");
        }
    }
    return 0;
}
