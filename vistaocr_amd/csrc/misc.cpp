// libvocr: error reporting and ABI/device queries (host only).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
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
extern "C" int vocr_abi_version(void) { return VOCR_ABI_VERSION; }
extern "C" int vocr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        vocr_set_error("vocr_device_count: no HIP device visible");
        return VOCR_ENODEVICE;
    }
    return n;
}

// ---- profiling ranges (rocprofv3 --marker-trace): roctx is dlopen'ed on first use, so libvocr.so has no link-time dependency on
// it and a process that never asks for ranges never loads it.  The reference's only instrumentation is wall-clock prints around the
// training step (src/train_cnn_lstm.py:382-393); these are the same phase boundaries, visible on the profiler's timeline.
namespace {
struct Roctx {
    int state = 0;                      // 0 not tried, 1 loaded, -1 unavailable
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx& roctx() {
    static Roctx r;
    if (r.state == 0) {
        r.state = -1;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            r.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            r.pop = (int (*)())dlsym(h, "roctxRangePop");
            if (r.push && r.pop) { r.state = 1; break; }
        }
    }
    return r;
}
}  // namespace

extern "C" int vocr_profile_range_push(const char* name) {
    if (!name) { vocr_set_error("vocr_profile_range_push: null name"); return VOCR_EINVAL; }
    Roctx& r = roctx();
    if (r.state != 1) return 1;         // roctx is not on this machine: ranges are simply not recorded
    r.push(name);
    return VOCR_OK;
}

extern "C" int vocr_profile_range_pop(void) {
    Roctx& r = roctx();
    if (r.state != 1) return 1;
    r.pop();
    return VOCR_OK;
}
