// libwxhip.so: error string, version, device probe.
#include "wx_common.h"

namespace wx {
char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace wx

extern "C" {

const char* wx_last_error(void) { return wx::last_error_buf(); }

const char* wx_version(void) { return "wxhip 0.1.0 gfx950"; }

int wx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

}  // extern "C"
