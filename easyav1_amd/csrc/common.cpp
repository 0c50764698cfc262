// Error plumbing + device query of the C ABI.
#include "ea1_common.h"

namespace ea1 {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ea1

extern "C" const char *ea1_last_error(void) { return ea1::err_buf(); }

extern "C" int ea1_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return ea1::fail(EA1_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
