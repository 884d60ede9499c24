// Fused Adam over ONE flat fp32 parameter buffer with per-segment learning rates: the optimiser of
// Code_Uncached/run.py:323-336 (torch.optim.Adam defaults, five parameter groups) as a single HBM-bound launch
// (4 streams of 4.1 M floats) instead of 146 small tensor updates.  grad_scale folds the 1/world_size of the
// data-parallel gradient average into the same pass.
#include "common.h"

namespace {

struct Segs {
    int64_t end[8];
    float lr[8];
    int n;
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, Segs segs, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float lr = segs.lr[segs.n - 1];
        for (int s = 0; s < segs.n; ++s)
            if (i < segs.end[s]) {
                lr = segs.lr[s];
                break;
            }
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

}  // namespace

extern "C" int iisan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                               const float* seg_lr, int32_t n_seg, int32_t step, float beta1, float beta2, float eps,
                               float grad_scale, void* stream) {
    IISAN_CHECK_SHAPE(n > 0 && n_seg >= 1 && n_seg <= 8 && step >= 1, "adam_step: bad arguments (n=%lld n_seg=%d step=%d)", (long long)n, n_seg, step);
    Segs s{};
    s.n = n_seg;
    for (int i = 0; i < n_seg; ++i) {
        s.end[i] = seg_end[i];
        s.lr[i] = seg_lr[i];
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, s, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
