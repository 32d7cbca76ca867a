// Host-side helpers of libwxhip.so: status/error plumbing shared by every entry point.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/wxhip.h"

namespace wx {

char* last_error_buf();  // thread-local, 512 bytes (wx_api.hip)

inline wx_status fail(wx_status st, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return st;
}

#define WX_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return ::wx::fail(WX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                              \
    } while (0)

// Every entry point enqueues on a caller-supplied stream.  A stream belongs to one device, and a launch on it
// from a thread whose CURRENT device is another one fails with an invalid-handle error (a process that drives
// several GPUs, or a rank whose current device is not the plan's): make the stream's device current for the
// duration of the call.  Costs two runtime queries per call; a no-op for the null stream.
struct StreamDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit StreamDeviceGuard(hipStream_t st) {
        hipDevice_t dev = -1;
        if (st != nullptr && hipStreamGetDevice(st, &dev) == hipSuccess && hipGetDevice(&prev) == hipSuccess &&
            (int)dev != prev)
            switched = hipSetDevice((int)dev) == hipSuccess;
    }
    ~StreamDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    StreamDeviceGuard(const StreamDeviceGuard&) = delete;
    StreamDeviceGuard& operator=(const StreamDeviceGuard&) = delete;
};
#define WX_STREAM(st, stream)                             \
    hipStream_t st = static_cast<hipStream_t>(stream);    \
    ::wx::StreamDeviceGuard st##_device_guard(st)

}  // namespace wx
