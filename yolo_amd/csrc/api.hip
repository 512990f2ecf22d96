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

extern "C" int y3_version(void) { return Y3_VERSION; }
extern "C" const char *y3_last_error(void) { return g_err; }
