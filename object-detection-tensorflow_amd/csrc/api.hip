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
}  // namespace odtk

extern "C" const char* odtk_last_error(void) { return odtk::g_err; }
extern "C" int odtk_version(void) { return 100; }
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
