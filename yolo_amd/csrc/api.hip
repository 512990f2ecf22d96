// Error reporting and version for libyolo3hip.so (see include/yolo3hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void y3_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int y3_check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        y3_set_error("%s: %s", what, hipGetErrorString(e));
        return Y3_E_LAUNCH;
    }
    return Y3_OK;
}

int y3_device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int slot = dev >= 0 && dev < 64 ? dev : 0;
    int v = __atomic_load_n(&cached[slot], __ATOMIC_RELAXED);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
            (void)hipGetLastError();
            v = 256;
        }
        __atomic_store_n(&cached[slot], v, __ATOMIC_RELAXED);
    }
    return v;
}

extern "C" int y3_version(void) { return Y3_VERSION; }
extern "C" const char *y3_last_error(void) { return g_err; }
