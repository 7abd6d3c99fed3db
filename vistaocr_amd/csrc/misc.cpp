// libvocr: error reporting and ABI/device queries (host only).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vocr.h"

static thread_local char g_err[512] = "";

void vocr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vocr_last_error(void) { return g_err; }
extern "C" int vocr_abi_version(void) { return 2; }
extern "C" int vocr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        vocr_set_error("vocr_device_count: no HIP device visible");
        return VOCR_ENODEVICE;
    }
    return n;
}
