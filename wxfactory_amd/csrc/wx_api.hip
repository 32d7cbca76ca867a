// libwxhip.so: error string, version, device probe.
#include "wx_common.h"
#include "wx_mfma.h"

#ifndef WX_K2_DIAG
#define WX_K2_DIAG 0
#endif
#define WX_STR2(x) #x
#define WX_STR(x) WX_STR2(x)

namespace wx {
char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace wx

extern "C" {

const char* wx_last_error(void) { return wx::last_error_buf(); }

const char* wx_version(void) { return "wxhip 0.1.0 gfx950"; }

// the build switches this library was compiled with (measurement provenance: bench.py refuses counter figures taken
// on another variant)
const char* wx_build_info(void) { return "WX_MFMA=" WX_STR(WX_MFMA) " WX_K2_DIAG=" WX_STR(WX_K2_DIAG); }

int wx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

}  // extern "C"
