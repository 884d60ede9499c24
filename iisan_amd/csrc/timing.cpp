// Optional per-launch timing of the dominant kernel family with HIP events recorded on the launching stream: class 1 = the
// 16-bit encoder GEMM (gemm16*, the Uncached headline), class 2 = the f32-matrix-core GEMM family of the trainable side
// (gemm32.hip: tiled / K = 64 / N = 64 + fusion / weight-gradient kernels — the dominant family of the Cached and Versa steps).
// bench.py enables ONE class around its timed region to report that family's roofline fraction; disabled it costs one branch
// per launch.  Not part of the product ABI (declared in common.h only; exported for bench.py).
#include <vector>

#include "common.h"

namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; };
double g_last_bytes = 0;
int g_on = 0;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

// Stream filter (bench.py, overlapped towers): with the text tower on a second HIP stream its launches queue for compute units
// behind the image tower's, and the span between their events is waiting, not running — only the launches on the filter stream
// (the caller's main stream: the ViT tower, 87 % of the encoder GEMM FLOPs) are then a kernel measure.  null = every stream.
static hipStream_t g_only = nullptr;
static bool g_filter = false;
extern "C" void iisan_timing_only_stream(void* stream, int32_t on) { g_only = (hipStream_t)stream; g_filter = on != 0; }
bool iisan_timing_on(hipStream_t s) { return g_on == 1 && (!g_filter || s == g_only); }
int iisan_timing_class() { return g_on; }
void iisan_timing_pre(hipStream_t s, double flops, double bytes) {
    Rec r{take(), take(), flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void iisan_timing_post(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }

extern "C" void iisan_timing_enable(int cls) { g_on = cls; }
// Synchronises on the recorded events; returns the number of launches and fills total milliseconds / total FLOPs.
extern "C" int64_t iisan_timing_collect(double* total_ms, double* total_flops) {
    double ms = 0, fl = 0, by = 0;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0;
        (void)hipEventElapsedTime(&t, r.a, r.b);
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
