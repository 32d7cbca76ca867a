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

}  // namespace wx
