// Library-level entry points: version / arch / thread-local error string.
#include <stdarg.h>

#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void iisan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* iisan_version(void) { return "iisan_hip 0.3 (round 3)"; }
extern "C" const char* iisan_arch(void) { return "gfx950"; }
extern "C" const char* iisan_last_error(void) { return g_err; }

// ---- route notes -------------------------------------------------------------------------------------------------
// A forward call whose kernel route depends on a process-wide knob (iisan_set_ce_fast, iisan_set_x3) notes the route it TOOK
// under its workspace pointer; the matching backward call dispatches on that note instead of on the knob's current value, so
// a knob flipped between the two calls can no longer make a backward read workspace the forward never wrote.
// Host memory only (the ABI never syncs): a small ring, newest note of a (workspace, kind) pair wins.
namespace {
struct RouteNote { const void* ws; uint32_t kind; uint64_t value; };
constexpr int ROUTE_NOTES = 256;
RouteNote g_notes[ROUTE_NOTES];
int g_note_next = 0;
std::mutex g_note_mu;
}  // namespace

void iisan_route_note(const void* ws, uint32_t kind, uint64_t value) {
    std::lock_guard<std::mutex> lk(g_note_mu);
    for (int i = 0; i < ROUTE_NOTES; ++i)
        if (g_notes[i].ws == ws && g_notes[i].kind == kind) { g_notes[i].value = value; return; }
    g_notes[g_note_next] = RouteNote{ws, kind, value};
    g_note_next = (g_note_next + 1) % ROUTE_NOTES;
}

bool iisan_route_find(const void* ws, uint32_t kind, uint64_t* value) {
    std::lock_guard<std::mutex> lk(g_note_mu);
    for (int i = 0; i < ROUTE_NOTES; ++i)
        if (g_notes[i].ws == ws && g_notes[i].kind == kind) { *value = g_notes[i].value; return true; }
    return false;
}

int iisan_cu_count() {
    static int cached[IISAN_MAX_DEVICES] = {};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IISAN_MAX_DEVICES) return 256;
    if (cached[dev] > 0) return cached[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
    return cus;
}
