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
}  // namespace odtk
extern "C" const char* odtk_last_error(void) { return odtk::g_err; }
