// Intra-/inter-modal Side Adapted Network: forward and backward executors + the fusion kernels.
// Replaces IISANAdaptedMModel.forward (Code_Uncached/model/model.py:209-271; Code_Cached/model/model.py:300-349)
// and the AdapterBlock it stacks (Code_*/model/modules.py:98-116), plus the backward PyTorch autograd derives.
//
//   for k in 0..n-1:   F_cv = g·tap_v[k] + (1-g)·cv      F_t = g·tap_t[k] + (1-g)·text      F_mm = mm + g·tap_v + (1-g)·tap_t
//                      state_z = Wu_z·act(Wd_z·F_z + bd_z) + bu_z + F_z                      (three towers z)
//   E_z = head_z(fc_z(state_z))                                            -> item3 = [E_cv | E_text | E_mm]
//
// All fp32.  The three towers have identical shapes, so every GEMM / column-sum / fusion step is ONE launch with
// three problems.  Fusion kernels are HBM-bound (16-byte lanes, taps read in place from the [M, L, D] tap tensor);
// GEMMs run on the f32 matrix cores (gemm32.hip).  Saved for backward: F, pre-activation U, activation, state per
// step (the [M,64] bottlenecks are tiny; F/state are 2·n·3·M·D floats).
#include "common.h"

int launch_colsum(const float* const* X, float* const* out, const int64_t* M, const int32_t* N, const int32_t* ld,
                  int nprob, hipStream_t s);

namespace {

struct FuseArgs {
    const float* taps_cv; const float* taps_text;     // [M, stride, D]
    int64_t M; int32_t D; int32_t stride_cv, stride_text, idx;
    const float* gate[3];      // device scalars (theta) or null when not gated
    const float* prev[3];      // previous state [M, D] or null (= zeros)
    int32_t prev_is_tap;       // remove_first, k == 0: prev_cv / prev_text are taps[:, first_index]
    int32_t first_index;
    float* F[3];               // outputs
};

__device__ __forceinline__ float gate_of(const float* theta) { return 1.0f / (1.0f + __expf(-theta[0] / 0.1f)); }

__global__ __launch_bounds__(256) void fuse_fwd_kernel(FuseArgs a) {
    const int z = blockIdx.y;
    const int d4 = a.D / 4;
    const int64_t total = a.M * d4;
    const bool gated = a.gate[0] != nullptr;
    const float g = gated ? gate_of(a.gate[z]) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / d4;
        const int c = (int)(i - m * d4) * 4;
        const f4 tv = *(const f4*)(a.taps_cv + (m * a.stride_cv + a.idx) * a.D + c);
        const f4 tt = *(const f4*)(a.taps_text + (m * a.stride_text + a.idx) * a.D + c);
        f4 pv = {0.f, 0.f, 0.f, 0.f};
        if (a.prev[z]) pv = *(const f4*)(a.prev[z] + m * a.D + c);
        else if (a.prev_is_tap && z == 0) pv = *(const f4*)(a.taps_cv + (m * a.stride_cv + a.first_index) * a.D + c);
        else if (a.prev_is_tap && z == 1) pv = *(const f4*)(a.taps_text + (m * a.stride_text + a.first_index) * a.D + c);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (z == 0) o[e] = gated ? g * tv[e] + (1.f - g) * pv[e] : tv[e] + pv[e];
            else if (z == 1) o[e] = gated ? g * tt[e] + (1.f - g) * pv[e] : tt[e] + pv[e];
            else o[e] = gated ? pv[e] + g * tv[e] + (1.f - g) * tt[e] : pv[e] + tv[e] + tt[e];
        }
        *(f4*)(a.F[z] + m * a.D + c) = o;
    }
}

struct FuseBwdArgs {
    const float* taps_cv; const float* taps_text;
    int64_t M; int32_t D; int32_t stride_cv, stride_text, idx;
    const float* gate[3];
    const float* prev[3];
    int32_t prev_is_tap, first_index;
    float* dF[3];              // in: grad wrt F_z ; out (in place): grad wrt prev_z
    float* dgate[3];           // accumulate d theta
};

// dprev = (1-g)·dF (cv,text) | dF (mm);  dtheta += [sum dF⊙(tap_a - b)] · g(1-g)/0.1
__global__ __launch_bounds__(256) void fuse_bwd_kernel(FuseBwdArgs a) {
    __shared__ float red[4];
    const int z = blockIdx.y;
    const int d4 = a.D / 4;
    const int64_t total = a.M * d4;
    const bool gated = a.gate[0] != nullptr;
    const float g = gated ? gate_of(a.gate[z]) : 1.0f;
    float part = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / d4;
        const int c = (int)(i - m * d4) * 4;
        float* dp = a.dF[z] + m * a.D + c;
        const f4 df = *(const f4*)dp;
        if (!gated) continue;                      // dprev = dF: already in place
        const f4 tv = *(const f4*)(a.taps_cv + (m * a.stride_cv + a.idx) * a.D + c);
        const f4 tt = *(const f4*)(a.taps_text + (m * a.stride_text + a.idx) * a.D + c);
        f4 pv = {0.f, 0.f, 0.f, 0.f};
        if (a.prev[z]) pv = *(const f4*)(a.prev[z] + m * a.D + c);
        else if (a.prev_is_tap && z == 0) pv = *(const f4*)(a.taps_cv + (m * a.stride_cv + a.first_index) * a.D + c);
        else if (a.prev_is_tap && z == 1) pv = *(const f4*)(a.taps_text + (m * a.stride_text + a.first_index) * a.D + c);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (z == 0) { part += df[e] * (tv[e] - pv[e]); o[e] = (1.f - g) * df[e]; }
            else if (z == 1) { part += df[e] * (tt[e] - pv[e]); o[e] = (1.f - g) * df[e]; }
            else { part += df[e] * (tv[e] - tt[e]); o[e] = df[e]; }
        }
        if (z != 2) *(f4*)dp = o;
    }
    if (!gated) return;
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.dgate[z], (red[0] + red[1] + red[2] + red[3]) * g * (1.f - g) / 0.1f);
}

// parameter table indices (see include/iisan_hip.h)
struct PIdx {
    int n;
    int wd(int z, int k) const { return (z * n + k) * 4 + 0; }
    int bd(int z, int k) const { return (z * n + k) * 4 + 1; }
    int wu(int z, int k) const { return (z * n + k) * 4 + 2; }
    int bu(int z, int k) const { return (z * n + k) * 4 + 3; }
    int gate(int z, int k) const { return 12 * n + z * n + k; }
    int fc_w(int z) const { return 15 * n + 2 * z; }
    int fc_b(int z) const { return 15 * n + 2 * z + 1; }
    int head_w(int z) const { return 15 * n + 6 + 2 * z; }
    int head_b(int z) const { return 15 * n + 6 + 2 * z + 1; }
};

struct SideBufs {
    float* F[IISAN_MAX_SIDE][3];
    float* U[IISAN_MAX_SIDE][3];
    float* A[IISAN_MAX_SIDE][3];     // activation(U)
    float* O[IISAN_MAX_SIDE][3];     // state after SANB k
    float* Y[3];                     // fc_z(state)
    float* dO[3];                    // backward scratch [M, D]
    float* dY[3];
    float* dU[3];                    // [M, down]
};

void carve(WsCarver& c, SideBufs& b, const iisan_side_cfg* cfg, int64_t M) {
    const size_t MD = (size_t)M * cfg->dim_cv, Mr = (size_t)M * cfg->down;
    for (int k = 0; k < cfg->n_side; ++k)
        for (int z = 0; z < 3; ++z) {
            b.F[k][z] = c.take<float>(MD);
            b.U[k][z] = c.take<float>(Mr);
            b.A[k][z] = c.take<float>(Mr);
            b.O[k][z] = c.take<float>(MD);
        }
    for (int z = 0; z < 3; ++z) {
        b.Y[z] = c.take<float>(MD);
        b.dO[z] = c.take<float>(MD);
        b.dY[z] = c.take<float>(MD);
        b.dU[z] = c.take<float>(Mr);
    }
}

int check_cfg(const iisan_side_cfg* cfg, int64_t M) {
    IISAN_CHECK_SHAPE(M > 0, "side_net: M must be positive");
    IISAN_CHECK_SHAPE(cfg->n_side >= 1 && cfg->n_side <= IISAN_MAX_SIDE, "side_net: n_side %d out of range", cfg->n_side);
    IISAN_CHECK_SHAPE(cfg->dim_cv == cfg->dim_text, "side_net: towers of different width (%d vs %d) are the Versa variant "
                      "(Code_Cached_Asym), not built yet", cfg->dim_cv, cfg->dim_text);
    IISAN_CHECK_SHAPE(cfg->dim_cv % 4 == 0 && cfg->down % 4 == 0 && cfg->emb % 4 == 0, "side_net: widths must be multiples of 4");
    for (int k = 0; k < cfg->n_side; ++k)
        IISAN_CHECK_SHAPE(cfg->tap_index[k] >= 0 && cfg->tap_index[k] < cfg->tap_stride_cv && cfg->tap_index[k] < cfg->tap_stride_text,
                          "side_net: tap index %d outside the tap tensor", cfg->tap_index[k]);
    return IISAN_OK;
}

Gemm32Prob prob(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int64_t M, int N,
                int64_t K, const float* resid = nullptr, int ldr = 0, const float* act_src = nullptr) {
    Gemm32Prob p{};
    p.A = A; p.B = B; p.bias = bias; p.resid = resid; p.act_src = act_src; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr ? ldr : ldc;
    return p;
}

unsigned ew_grid(int64_t M, int D) {
    const int64_t b = ceil_div(M * (D / 4), 256);
    return (unsigned)(b < 4096 ? b : 4096);
}

}  // namespace

extern "C" size_t iisan_side_net_ws_bytes(const iisan_side_cfg* cfg, int64_t M) {
    WsCarver c(nullptr, 0);
    SideBufs b;
    carve(c, b, cfg, M);
    return c.off;
}

extern "C" int iisan_side_net_fwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                                  const void* const* params, float* item3, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check_cfg(cfg, M));
    WsCarver c(ws, ws_bytes);
    SideBufs b;
    carve(c, b, cfg, M);
    if (c.overflow || !ws) {
        iisan_set_error("side_net_fwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int n = cfg->n_side, D = cfg->dim_cv, r = cfg->down, E = cfg->emb;
    const PIdx P{n};
    auto W = [&](int i) { return (const float*)params[i]; };
    const int act_flag = cfg->gelu ? G32_GELU : G32_RELU;
    for (int k = 0; k < n; ++k) {
        FuseArgs fa{};
        fa.taps_cv = taps_cv; fa.taps_text = taps_text; fa.M = M; fa.D = D;
        fa.stride_cv = cfg->tap_stride_cv; fa.stride_text = cfg->tap_stride_text; fa.idx = cfg->tap_index[k];
        fa.prev_is_tap = (k == 0 && cfg->remove_first) ? 1 : 0;
        fa.first_index = cfg->first_index;
        for (int z = 0; z < 3; ++z) {
            fa.gate[z] = cfg->gated ? W(P.gate(z, k)) : nullptr;
            fa.prev[z] = k > 0 ? b.O[k - 1][z] : nullptr;
            fa.F[z] = b.F[k][z];
        }
        hipLaunchKernelGGL(fuse_fwd_kernel, dim3(ew_grid(M, D), 3), dim3(256), 0, s, fa);
        IISAN_LAUNCH_OK();
        Gemm32Prob pr[3];
        for (int z = 0; z < 3; ++z)   // U = F Wd^T + bd (saved), A = act(U)
            pr[z] = prob(b.F[k][z], D, W(P.wd(z, k)), D, W(P.bd(z, k)), b.A[k][z], r, M, r, D, nullptr, 0, b.U[k][z]);
        IISAN_TRY(launch_gemm32(pr, 3, act_flag | G32_PREACT, s));
        for (int z = 0; z < 3; ++z)   // state = A Wu^T + bu + F
            pr[z] = prob(b.A[k][z], r, W(P.wu(z, k)), r, W(P.bu(z, k)), b.O[k][z], D, M, D, r, b.F[k][z], D);
        IISAN_TRY(launch_gemm32(pr, 3, 0, s));
    }
    Gemm32Prob pr[3];
    for (int z = 0; z < 3; ++z) pr[z] = prob(b.O[n - 1][z], D, W(P.fc_w(z)), D, W(P.fc_b(z)), b.Y[z], D, M, D, D);
    IISAN_TRY(launch_gemm32(pr, 3, 0, s));
    for (int z = 0; z < 3; ++z) pr[z] = prob(b.Y[z], D, W(P.head_w(z)), D, W(P.head_b(z)), item3 + z * E, 3 * E, M, E, D);
    IISAN_TRY(launch_gemm32(pr, 3, 0, s));
    return IISAN_OK;
}

extern "C" int iisan_side_net_bwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                                  const void* const* params, const float* d_item3, void* const* grads, void* ws,
                                  size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check_cfg(cfg, M));
    WsCarver c(ws, ws_bytes);
    SideBufs b;
    carve(c, b, cfg, M);
    if (c.overflow || !ws) {
        iisan_set_error("side_net_bwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int n = cfg->n_side, D = cfg->dim_cv, r = cfg->down, E = cfg->emb;
    const PIdx P{n};
    auto W = [&](int i) { return (const float*)params[i]; };
    auto G = [&](int i) { return (float*)grads[i]; };
    Gemm32Prob pr[3];
    const float* cs_x[3]; float* cs_o[3]; int64_t cs_m[3] = {M, M, M}; int32_t cs_n[3], cs_ld[3];

    // heads: E_z = Y_z Wh^T + bh
    for (int z = 0; z < 3; ++z) pr[z] = prob(d_item3 + z * E, 3 * E, W(P.head_w(z)), D, nullptr, b.dY[z], D, M, D, E);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TB, s));                                   // dY = dE · Wh
    for (int z = 0; z < 3; ++z) pr[z] = prob(d_item3 + z * E, 3 * E, b.Y[z], D, nullptr, G(P.head_w(z)), D, E, D, M);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));              // dWh += dE^T · Y
    for (int z = 0; z < 3; ++z) { cs_x[z] = d_item3 + z * E; cs_o[z] = G(P.head_b(z)); cs_n[z] = E; cs_ld[z] = 3 * E; }
    IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));
    // fc: Y_z = O_z Wf^T + bf
    for (int z = 0; z < 3; ++z) pr[z] = prob(b.dY[z], D, W(P.fc_w(z)), D, nullptr, b.dO[z], D, M, D, D);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TB, s));                                   // dO = dY · Wf
    for (int z = 0; z < 3; ++z) pr[z] = prob(b.dY[z], D, b.O[n - 1][z], D, nullptr, G(P.fc_w(z)), D, D, D, M);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));              // dWf += dY^T · O
    for (int z = 0; z < 3; ++z) { cs_x[z] = b.dY[z]; cs_o[z] = G(P.fc_b(z)); cs_n[z] = D; cs_ld[z] = D; }
    IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));

    for (int k = n - 1; k >= 0; --k) {
        // state = A Wu^T + bu + F ; A = act(U) ; U = F Wd^T + bd
        for (int z = 0; z < 3; ++z)
            pr[z] = prob(b.dO[z], D, W(P.wu(z, k)), r, nullptr, b.dU[z], r, M, r, D, nullptr, 0, b.U[k][z]);
        IISAN_TRY(launch_gemm32(pr, 3, G32_TB | (cfg->gelu ? G32_MUL_GELU_GRAD : G32_MUL_RELU_MASK), s));  // dU
        for (int z = 0; z < 3; ++z) pr[z] = prob(b.dO[z], D, b.A[k][z], r, nullptr, G(P.wu(z, k)), r, D, r, M);
        IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));          // dWu += dO^T · A
        for (int z = 0; z < 3; ++z) { cs_x[z] = b.dO[z]; cs_o[z] = G(P.bu(z, k)); cs_n[z] = D; cs_ld[z] = D; }
        IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));
        for (int z = 0; z < 3; ++z) pr[z] = prob(b.dU[z], r, b.F[k][z], D, nullptr, G(P.wd(z, k)), D, r, D, M);
        IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));          // dWd += dU^T · F
        for (int z = 0; z < 3; ++z) { cs_x[z] = b.dU[z]; cs_o[z] = G(P.bd(z, k)); cs_n[z] = r; cs_ld[z] = r; }
        IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));
        for (int z = 0; z < 3; ++z) pr[z] = prob(b.dU[z], r, W(P.wd(z, k)), D, nullptr, b.dO[z], D, M, D, r, b.dO[z], D);
        IISAN_TRY(launch_gemm32(pr, 3, G32_TB, s));                               // dF = dO + dU · Wd (in place)
        if (cfg->gated) {
            FuseBwdArgs fa{};
            fa.taps_cv = taps_cv; fa.taps_text = taps_text; fa.M = M; fa.D = D;
            fa.stride_cv = cfg->tap_stride_cv; fa.stride_text = cfg->tap_stride_text; fa.idx = cfg->tap_index[k];
            fa.prev_is_tap = (k == 0 && cfg->remove_first) ? 1 : 0;
            fa.first_index = cfg->first_index;
            for (int z = 0; z < 3; ++z) {
                fa.gate[z] = W(P.gate(z, k));
                fa.prev[z] = k > 0 ? b.O[k - 1][z] : nullptr;
                fa.dF[z] = b.dO[z];
                fa.dgate[z] = G(P.gate(z, k));
            }
            hipLaunchKernelGGL(fuse_bwd_kernel, dim3(ew_grid(M, D), 3), dim3(256), 0, s, fa);
            IISAN_LAUNCH_OK();
        }
        // not gated: dprev_z = dF_z, already in dO
    }
    return IISAN_OK;
}
