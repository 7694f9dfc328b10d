// TEST INFRASTRUCTURE ONLY: the two library-internal helpers csrc/api.hip provides, for the CPU build of the emulated kernels (tests/hip_cpu/hip/hip_runtime.h)
#include <stdarg.h>
#include "common.h"

namespace odtk {
static char g_err[1024];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
int zero_async(void* p, size_t bytes, hipStream_t) { memset(p, 0, bytes); return 0; }
namespace cv {
static bool g_deterministic = true;
bool get_wgrad_deterministic() { return g_deterministic; }      // (csrc/conv_v3.hip's switch: elementwise.hip's L2-norm gamma gradient follows it)
int get_scratch_slot() { return 0; }                            // (csrc/conv_v3.hip: the calling thread's scratch slot)
int misc_scratch(size_t bytes, hipStream_t, char** out) {       // (csrc/conv_v3.hip: the box-side kernels' partial-sum arena; here one grow-only host buffer)
    static char* base = nullptr;
    static size_t have = 0;
    if (have < bytes) { free(base); base = (char*)malloc(bytes); have = bytes; }
    *out = base;
    return 0;
}
}  // namespace cv
}  // namespace odtk
extern "C" const char* odtk_last_error(void) { return odtk::g_err; }

// the three host-side entry points of csrc/conv.hip / api.hip that the emulated files' tests use (the convolution files themselves are not built)
extern "C" int odtk_debug_set(int key, int value) {
    if (key == 3) { odtk::set_nms_legacy(value != 0); return ODTK_OK; }
    if (key == 5) { odtk::cv::g_deterministic = value != 0; return ODTK_OK; }
    if (key == 4) { odtk::set_bn_small_rows(value); return ODTK_OK; }      // (incl. -7 / -8: the ticket finalize)
    if (key == 7) { odtk::set_gn_small_rows(value); return ODTK_OK; }
    odtk::set_error("debug_set (CPU emulation): key %d belongs to the convolution files", key);
    return ODTK_ERR_ARG;
}
extern "C" int odtk_zero(void* p, long long bytes, void*) { memset(p, 0, (size_t)bytes); return ODTK_OK; }
extern "C" int odtk_version(void) { return 100; }
extern "C" long long hipcpu_launch_count(void) { return hipcpu::launch_counter(); }
