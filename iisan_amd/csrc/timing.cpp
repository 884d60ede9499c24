// Optional per-launch timing of the dominant kernel (gemm16) with HIP events recorded on the launching stream.
// bench.py enables it around its timed region to report the roofline fraction of that kernel; disabled it costs
// one branch per launch.  Not part of the product ABI (declared in common.h only; exported for bench.py).
#include <vector>

#include "common.h"

namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; };
double g_last_bytes = 0;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

bool iisan_timing_on() { return g_on; }
void iisan_timing_pre(hipStream_t s, double flops, double bytes) {
    Rec r{take(), take(), flops, bytes};
    hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void iisan_timing_post(hipStream_t s) { hipEventRecord(g_recs.back().b, s); }

extern "C" void iisan_timing_enable(int on) { g_on = on != 0; }
// Synchronises on the recorded events; returns the number of launches and fills total milliseconds / total FLOPs.
extern "C" int64_t iisan_timing_collect(double* total_ms, double* total_flops) {
    double ms = 0, fl = 0, by = 0;
    for (auto& r : g_recs) {
        hipEventSynchronize(r.b);
        float t = 0;
        hipEventElapsedTime(&t, r.a, r.b);
        ms += t;
        fl += r.flops;
        by += r.bytes;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    const int64_t n = (int64_t)g_recs.size();
    g_recs.clear();
    g_last_bytes = by;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return n;
}
// algorithmic bytes (operands read once + output written once) of the launches of the last iisan_timing_collect()
extern "C" double iisan_timing_last_bytes() { return g_last_bytes; }
