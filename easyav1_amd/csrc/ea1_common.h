// Internal helpers shared by the HIP translation units of libeasyav1_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "easyav1_amd_audio.h"

namespace ea1 {

// thread-local error text behind ea1_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define EA1_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ea1::fail(EA1_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                            \
    } while (0)

#define EA1_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess)                                                                \
            return ea1::fail(EA1_EHIP, "kernel launch failed: %s (%s:%d)",                    \
                             hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

// RAII device switch: every entry point runs on its object's device and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace ea1
