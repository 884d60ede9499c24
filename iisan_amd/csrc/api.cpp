// Library-level entry points: version / arch / thread-local error string.
#include <stdarg.h>

#include <deque>
#include <functional>
#include <mutex>
#include <unordered_map>

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
// Host memory only (the ABI never syncs): a hash map, newest note of a (workspace, kind) pair wins.
namespace {
struct RouteKey {
    const void* ws; uint32_t kind;
    bool operator==(const RouteKey& o) const { return ws == o.ws && kind == o.kind; }
};
struct RouteHash {
    size_t operator()(const RouteKey& k) const { return std::hash<const void*>()(k.ws) * 31u + k.kind; }
};
constexpr size_t ROUTE_NOTES_MAX = 1u << 16;      // forwards whose backward has not run yet: far more than any training loop keeps
std::unordered_map<RouteKey, uint64_t, RouteHash> g_notes;
std::deque<RouteKey> g_note_order;                // insertion order, to forget the oldest beyond the cap
std::mutex g_note_mu;
}  // namespace

void iisan_route_note(const void* ws, uint32_t kind, uint64_t value) {
    std::lock_guard<std::mutex> lk(g_note_mu);
    const RouteKey k{ws, kind};
    auto it = g_notes.find(k);
    if (it != g_notes.end()) { it->second = value; return; }
    g_notes.emplace(k, value);
    g_note_order.push_back(k);
    while (g_note_order.size() > ROUTE_NOTES_MAX) {
        g_notes.erase(g_note_order.front());
        g_note_order.pop_front();
    }
}

bool iisan_route_find(const void* ws, uint32_t kind, uint64_t* value) {
    std::lock_guard<std::mutex> lk(g_note_mu);
    auto it = g_notes.find(RouteKey{ws, kind});
    if (it == g_notes.end()) return false;
    *value = it->second;
    return true;
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
