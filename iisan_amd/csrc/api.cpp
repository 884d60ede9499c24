// Library-level entry points: version / arch / thread-local error string.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void iisan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* iisan_version(void) { return "iisan_hip 0.5 (round 4b)"; }
extern "C" const char* iisan_arch(void) { return "gfx950"; }
extern "C" const char* iisan_last_error(void) { return g_err; }

int iisan_cu_count() {
    static int cached[IISAN_MAX_DEVICES] = {};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IISAN_MAX_DEVICES) return 256;
    if (cached[dev] > 0) return cached[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
    return cus;
}
