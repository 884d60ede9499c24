// Development switches of the library: ONE declared entry point pair (include/iisan_hip.h, DEV section) instead of a setter per
// switch.  Every kernel file registers its process-wide route / ablation variables here under a name (IISAN_DEV_KNOB in common.h);
// tests and bench A/Bs reach them through iisan_dev_set / iisan_dev_get, iisan_dev_state lists what is NOT at its library default
// (tests/conftest.py asserts that list is empty after every test), iisan_dev_reset restores the defaults.  A product process
// never calls any of these: the defaults ARE the product routes.
#include <string>
#include <vector>

#include "common.h"

namespace {
std::vector<IisanDevKnob>& knobs() {
    static std::vector<IisanDevKnob> k;      // function-local: registration runs from other files' static initialisers
    return k;
}
const IisanDevKnob* find(const char* name) {
    if (!name) return nullptr;
    for (const auto& k : knobs())
        if (strcmp(k.name, name) == 0) return &k;
    return nullptr;
}
}  // namespace

int iisan_dev_register(const IisanDevKnob& k) {
    knobs().push_back(k);
    return (int)knobs().size();
}

extern "C" int32_t iisan_dev_set(const char* name, int64_t value) {
    const IisanDevKnob* k = find(name);
    if (!k) {
        iisan_set_error("iisan_dev_set: no development switch named '%s'", name ? name : "(null)");
        return IISAN_EBADSHAPE;
    }
    k->set(value);
    return IISAN_OK;
}

extern "C" int64_t iisan_dev_get(const char* name) {
    const IisanDevKnob* k = find(name);
    if (!k) {
        iisan_set_error("iisan_dev_get: no development switch named '%s'", name ? name : "(null)");
        return INT64_MIN;
    }
    return k->get();
}

extern "C" void iisan_dev_reset(void) {
    for (const auto& k : knobs()) k.set(k.def == INT64_MIN ? 0 : k.def);      // (INT64_MIN: a route counter, common.h IISAN_DEV_COUNTER)
}

// "name=value,name=value" of every switch that is not at its library default ("" = the product routes); with all != 0 every
// switch is listed.  Returns the length needed (without the terminator); writes at most cap - 1 characters + '\0'.
extern "C" size_t iisan_dev_state(char* buf, size_t cap, int32_t all) {
    std::string s;
    for (const auto& k : knobs()) {
        const int64_t v = k.get();
        if (k.def == INT64_MIN) { if (!all) continue; }          // a route counter is not a switch: only the full listing shows it
        else if (!all && v == k.def) continue;
        if (!s.empty()) s += ',';
        s += k.name;
        s += '=';
        s += std::to_string((long long)v);
    }
    if (buf && cap > 0) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = '\0';
    }
    return s.size();
}
