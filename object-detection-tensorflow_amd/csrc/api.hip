// libodtk: error reporting and device queries (host side of the C-ABI).
#include "common.h"
#include <stdarg.h>

namespace odtk {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4* __restrict__ p16, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
}  // namespace

int zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return ODTK_OK;
    unsigned char* b = (unsigned char*)p;
    size_t head = (16 - ((size_t)(uintptr_t)b & 15)) & 15;            // bytes up to the first 16-byte boundary
    if (head > bytes) head = bytes;
    if (head) hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (uint4*)nullptr, (size_t)0, b, (int)head);
    const size_t n16 = (bytes - head) / 16, ntail = (bytes - head) % 16;
    if (n16 || ntail) {
        size_t blocks = (n16 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks == 0) blocks = 1;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)(b + head), n16, b + head + n16 * 16, (int)ntail);
    }
    ODTK_LAUNCH_CHECK();
    return ODTK_OK;
}
}  // namespace odtk

extern "C" const char* odtk_last_error(void) { return odtk::g_err; }
extern "C" int odtk_zero(void* p, long long bytes, void* stream) {
    if (p == nullptr || bytes < 0) { odtk::set_error("zero: bad argument"); return ODTK_ERR_ARG; }
    return odtk::zero_async(p, (size_t)bytes, (hipStream_t)stream);
}
extern "C" int odtk_version(void) { return 100; }
// Host-side CRC32C (Castagnoli, reflected 0x82f63b78), slice-by-8: the checksum of TensorFlow's checkpoint blocks and
// tensors (tf_checkpoint.py reads / writes hundreds of MB of weights; SSD300.py:31, :490-504).  No device work.
extern "C" unsigned int odtk_crc32c(const void* data, long long n, unsigned int crc) {
    static unsigned int tab[8][256];
    static bool ready = false;
    if (!ready) {
        for (unsigned int i = 0; i < 256; ++i) {
            unsigned int c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
            tab[0][i] = c;
        }
        for (unsigned int i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) tab[s][i] = (tab[s - 1][i] >> 8) ^ tab[0][tab[s - 1][i] & 0xff];
        ready = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    unsigned int c = ~crc;
    while (n >= 8) {
        unsigned int lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = tab[7][lo & 0xff] ^ tab[6][(lo >> 8) & 0xff] ^ tab[5][(lo >> 16) & 0xff] ^ tab[4][lo >> 24] ^
            tab[3][hi & 0xff] ^ tab[2][(hi >> 8) & 0xff] ^ tab[1][(hi >> 16) & 0xff] ^ tab[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n-- > 0) c = tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

extern "C" int odtk_device_info(int* num_cu, char* name_buf, int name_buf_len) {
    int dev = 0;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ODTK_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (name_buf && name_buf_len > 0) {
        snprintf(name_buf, name_buf_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return ODTK_OK;
}
