// SASRec user encoder: forward + backward executors and their small kernels.
// Replaces User_Encoder.forward (Code_Uncached/model/encoders.py:60-65) -> TransformerEncoder / TransformerBlock /
// MultiHeadedAttention / SelfAttention / PositionwiseFeedForward (Code_*/model/modules.py:6-96), and the backward
// autograd derives from them.  fp32 throughout (d_model 64, 2 heads, 10 positions: ~2 MFLOP per sequence).
//
//   X0 = LN(in + pos)                                              (eps 1e-6)
//   per block:  Q,K,V = X·Wq^T, X·Wk^T, X·Wv^T (no bias) ; P = softmax(QK^T/sqrt(dh) + mask) ; C = P·V
//               X1 = LN(X + C·Wfc^T) ; X2 = LN(X1 + W2·relu(W1·X1 + b1) + b2)
//   mask[q,k] = 0 if (k <= q and log_mask[k] != 0) else -1e9      (encoders.py:60-64)
//
// Projections/FFN run as fp32-MFMA GEMMs over all B*S rows (gemm32.hip); LayerNorm fwd/bwd are one-wave-per-row
// shuffle kernels; the 10x10 attention is a thread-per-query-row kernel.  Dropout (reference drop_rate in training:
// after the embedding LayerNorm, on the attention probabilities, on the fc output and on the FFN output,
// modules.py:17,31,62,94) uses the counter-based masks of common.h: site 0 = embedding, 1+3l / 2+3l / 3+3l = block l's
// attention / fc / FFN; cfg.dropout == 0 is the exact eval path.
#include "common.h"

int launch_colsum(const float* const* X, float* const* out, const int64_t* M, const int32_t* N, const int32_t* ld,
                  int nprob, hipStream_t s);

// one-launch forward / backward for the production shape (sasrec_fused.hip); works on the workspace slots carved here
struct SasFusedPtrs {
    float* Z0; float* X0;
    float* Q[8]; float* K[8]; float* V[8]; float* P[8]; float* C[8]; float* Zattn[8]; float* X1[8]; float* Hf[8]; float* Zffn[8]; float* X2[8];
};
bool sasrec_fused_ok(const iisan_sasrec_cfg* cfg);
bool sasrec_fused_shape_ok(const iisan_sasrec_cfg* cfg);
int launch_sasrec_fused_fwd(const iisan_sasrec_cfg* cfg, const float* x, const float* log_mask, int64_t B, const void* const* params,
                            float* y, const SasFusedPtrs& w, hipStream_t s);
int64_t sasrec_fused_slab_floats(const iisan_sasrec_cfg* cfg, int64_t B);
int launch_sasrec_fused_bwd(const iisan_sasrec_cfg* cfg, const float* log_mask, int64_t B, const void* const* params, const float* dy,
                            float* dx, void* const* grads, const SasFusedPtrs& w, float* slab, hipStream_t s);

namespace {

constexpr int MAXE = 256;   // d_model up to 256 (multiple of 64)

// y = LN(a + b) * g + beta ; optionally stores the pre-norm sum z = a + b (needed by backward).
// b_stride_rows: b is indexed by (row % b_rows) (position embedding) when b_rows > 0, else by row.
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t b_rows, const float* __restrict__ g,
                                                         const float* __restrict__ beta, float eps, float* __restrict__ zsum,
                                                         float* __restrict__ y, int64_t rows, int E, DropCfg drop) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int per = E / 64;
    float v[MAXE / 64];
    float s = 0.f;
    const int64_t brow = b_rows > 0 ? row % b_rows : row;
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        v[i] = a[row * E + c] + (b ? b[brow * E + c] : 0.f);
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
    for (int i = 0; i < per; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        if (zsum) zsum[row * E + c] = v[i];
        float o = (v[i] - mean) * rstd * g[c] + beta[c];
        if (drop.thr24) o *= drop_scale(drop.seed, drop.site, (uint64_t)(row * E + c), drop.thr24, drop.inv_keep);
        y[row * E + c] = o;
    }
}

// LayerNorm backward on rows z (pre-norm input): dz = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat));
// dgamma += sum dy*xhat ; dbeta += sum dy.  LNB_ROWS rows per block, block-level reduction, then atomics.
// (64 rows per block: 160 workgroups for the Cached batch, each wave a serial chain of 16 rows with four cross-lane sums per
// row — 27 us for 2.6 MB; 16 rows per block: four times the workgroups, a quarter of the chain.)
constexpr int LNB_ROWS = 16;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                     const float* __restrict__ g, float eps, float* __restrict__ dz,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int E,
                                                     DropCfg drop) {
    __shared__ float red[2][4][MAXE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = E / 64;
    float dg[MAXE / 64], db[MAXE / 64];
    for (int i = 0; i < per; ++i) dg[i] = db[i] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
    for (int rr = wave; rr < LNB_ROWS; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= rows) break;
        float v[MAXE / 64], d[MAXE / 64];
        float s = 0.f;
        for (int i = 0; i < per; ++i) {
            v[i] = z[row * E + i * 64 + lane];
            d[i] = dy[row * E + i * 64 + lane];
            if (drop.thr24) d[i] *= drop_scale(drop.seed, drop.site, (uint64_t)(row * E + i * 64 + lane), drop.thr24, drop.inv_keep);
            s += v[i];
        }
        const float mean = wave_sum(s) / (float)E;
        float q = 0.f;
        for (int i = 0; i < per; ++i) {
            v[i] -= mean;
            q += v[i] * v[i];
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < per; ++i) {
            const int c = i * 64 + lane;
            v[i] *= rstd;                    // xhat
            dg[i] += d[i] * v[i];
            db[i] += d[i];
            d[i] *= g[c];                    // dy * gamma
            s1 += d[i];
            s2 += d[i] * v[i];
        }
        s1 = wave_sum(s1) / (float)E;
        s2 = wave_sum(s2) / (float)E;
        for (int i = 0; i < per; ++i) dz[row * E + i * 64 + lane] = rstd * (d[i] - s1 - v[i] * s2);
    }
    for (int i = 0; i < per; ++i) {
        red[0][wave][i * 64 + lane] = dg[i];
        red[1][wave][i * 64 + lane] = db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// SASRec attention, one 64-thread workgroup per (sequence b, head h); S <= 32, dh <= 64.  Q/K/V: [B*S, E] (head h =
// columns h*dh..); P: [B,H,S,S]; C: [B*S,E].  The tiles are staged in LDS with coalesced loads and the three small
// products are spread over the lanes.  (The first version ran one THREAD per (b, h, q) — every thread a serial chain
// of S*dh dependent global loads: 83 us for bs = 128 and for bs = 1024 alike, 0.3 ms per step with the backward.)
// The summation orders are those of the first version (e ascending, k ascending), so results are unchanged.
constexpr int SA_LD = 65, SA_PL = 33;

__global__ __launch_bounds__(64) void sas_attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, const float* __restrict__ log_mask,
                                                          float* __restrict__ P, float* __restrict__ C, int64_t B, int S, int H,
                                                          int dh, DropCfg drop) {
    __shared__ float sQ[32 * SA_LD], sK[32 * SA_LD], sV[32 * SA_LD], sS[32 * SA_PL];
    const int t = threadIdx.x;
    const int64_t b = blockIdx.x / H;
    const int h = (int)(blockIdx.x - b * H);
    const int E = H * dh;
    const float temp = sqrtf((float)dh);
    for (int idx = t; idx < S * dh; idx += 64) {
        const int r = idx / dh, e = idx - r * dh;
        const int64_t g = (b * S + r) * E + h * dh + e;
        sQ[r * SA_LD + e] = Q[g];
        sK[r * SA_LD + e] = K[g];
        sV[r * SA_LD + e] = V[g];
    }
    __syncthreads();
    for (int idx = t; idx < S * S; idx += 64) {
        const int q = idx / S, k = idx - q * S;
        float d = 0.f;
        for (int e = 0; e < dh; ++e) d += sQ[q * SA_LD + e] * sK[k * SA_LD + e];
        const float m = (k <= q && log_mask[b * S + k] != 0.f) ? 0.f : -1e9f;
        sS[q * SA_PL + k] = d / temp + m;
    }
    __syncthreads();
    if (t < S) {
        const int q = t;
        float mx = -INFINITY;
        for (int k = 0; k < S; ++k) mx = fmaxf(mx, sS[q * SA_PL + k]);
        float sum = 0.f;
        for (int k = 0; k < S; ++k) {
            const float p = expf(sS[q * SA_PL + k] - mx);
            sS[q * SA_PL + k] = p;
            sum += p;
        }
        float* pr = P + ((b * H + h) * S + q) * S;
        for (int k = 0; k < S; ++k) {
            float p = sS[q * SA_PL + k] / sum;
            pr[k] = p;                 // probabilities BEFORE dropout are kept for backward
            if (drop.thr24) p *= drop_scale(drop.seed, drop.site, (uint64_t)(((b * H + h) * S + q) * S + k), drop.thr24, drop.inv_keep);
            sS[q * SA_PL + k] = p;
        }
    }
    __syncthreads();
    for (int idx = t; idx < S * dh; idx += 64) {
        const int q = idx / dh, e = idx - q * dh;
        float acc = 0.f;
        for (int k = 0; k < S; ++k) acc += sS[q * SA_PL + k] * sV[k * SA_LD + e];
        C[(b * S + q) * E + h * dh + e] = acc;
    }
}

// attention backward, same decomposition: dP = (dC V^T) * mask, dS = P (dP - rowdot) / temp, dQ = dS K, dK = dS^T Q,
// dV = (P * mask)^T dC.
__global__ __launch_bounds__(64) void sas_attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, const float* __restrict__ P,
                                                          const float* __restrict__ dC, float* __restrict__ dQ,
                                                          float* __restrict__ dK, float* __restrict__ dV, int S, int H, int dh,
                                                          DropCfg drop) {
    __shared__ float sQ[32 * SA_LD], sK[32 * SA_LD], sV[32 * SA_LD], sD[32 * SA_LD];
    __shared__ float sP[32 * SA_PL], sM[32 * SA_PL], sDP[32 * SA_PL];      // P, dropout keep factors, dP then dS
    const int t = threadIdx.x;
    const int64_t b = blockIdx.x / H;
    const int h = (int)(blockIdx.x - b * H);
    const int E = H * dh;
    const float temp = sqrtf((float)dh);
    for (int idx = t; idx < S * dh; idx += 64) {
        const int r = idx / dh, e = idx - r * dh;
        const int64_t g = (b * S + r) * E + h * dh + e;
        sQ[r * SA_LD + e] = Q[g];
        sK[r * SA_LD + e] = K[g];
        sV[r * SA_LD + e] = V[g];
        sD[r * SA_LD + e] = dC[g];
    }
    for (int idx = t; idx < S * S; idx += 64) {
        const int q = idx / S, k = idx - q * S;
        sP[q * SA_PL + k] = P[((b * H + h) * S + q) * S + k];
    }
    __syncthreads();
    for (int idx = t; idx < S * S; idx += 64) {
        const int q = idx / S, k = idx - q * S;
        float d = 0.f;
        for (int e = 0; e < dh; ++e) d += sD[q * SA_LD + e] * sV[k * SA_LD + e];
        const float mk = drop.thr24 ? drop_scale(drop.seed, drop.site, (uint64_t)(((b * H + h) * S + q) * S + k), drop.thr24, drop.inv_keep) : 1.f;
        sM[q * SA_PL + k] = mk;
        sDP[q * SA_PL + k] = d * mk;       // dP = dP_drop * mask/(1-p)
    }
    __syncthreads();
    if (t < S) {
        const int q = t;
        float dot = 0.f;
        for (int k = 0; k < S; ++k) dot += sP[q * SA_PL + k] * sDP[q * SA_PL + k];
        for (int k = 0; k < S; ++k) sDP[q * SA_PL + k] = sP[q * SA_PL + k] * (sDP[q * SA_PL + k] - dot) / temp;     // dS
    }
    __syncthreads();
    for (int idx = t; idx < S * dh; idx += 64) {
        const int r = idx / dh, e = idx - r * dh;
        const int64_t g = (b * S + r) * E + h * dh + e;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int k = 0; k < S; ++k) aq += sDP[r * SA_PL + k] * sK[k * SA_LD + e];
        for (int qq = 0; qq < S; ++qq) {
            ak += sDP[qq * SA_PL + r] * sQ[qq * SA_LD + e];
            av += sP[qq * SA_PL + r] * sM[qq * SA_PL + r] * sD[qq * SA_LD + e];
        }
        dQ[g] = aq;
        dK[g] = ak;
        dV[g] = av;
    }
}

// out = in * dropout_keep_factor  (gradient of a dropped GEMM output)
__global__ void drop_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, DropCfg drop) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i] * drop_scale(drop.seed, drop.site, (uint64_t)i, drop.thr24, drop.inv_keep);
}

struct BlockBufs {
    float *Q, *K, *V, *P, *C, *Zattn, *X1, *Hf, *Zffn, *X2;
};
struct SasBufs {
    float* Z0;              // in + pos
    float* X0;              // LN(Z0)
    BlockBufs blk[8];
    float *dA, *dB, *dQ, *dK, *dV, *dH, *dG;     // backward scratch
    float* slab;            // fused backward: per-workgroup partial parameter gradients (carved last)
};

void carve(WsCarver& c, SasBufs& b, const iisan_sasrec_cfg* cfg, int64_t B) {
    const size_t T = (size_t)B * cfg->seq, E = cfg->emb;
    b.Z0 = c.take<float>(T * E);
    b.X0 = c.take<float>(T * E);
    for (int l = 0; l < cfg->blocks; ++l) {
        BlockBufs& k = b.blk[l];
        k.Q = c.take<float>(T * E); k.K = c.take<float>(T * E); k.V = c.take<float>(T * E);
        k.P = c.take<float>((size_t)B * cfg->heads * cfg->seq * cfg->seq);
        k.C = c.take<float>(T * E); k.Zattn = c.take<float>(T * E); k.X1 = c.take<float>(T * E);
        k.Hf = c.take<float>(T * 4 * E); k.Zffn = c.take<float>(T * E); k.X2 = c.take<float>(T * E);
    }
    b.dA = c.take<float>(T * E); b.dB = c.take<float>(T * E);
    b.dQ = c.take<float>(T * E); b.dK = c.take<float>(T * E); b.dV = c.take<float>(T * E);
    b.dH = c.take<float>(T * 4 * E);
    b.dG = c.take<float>(T * E);
    b.slab = sasrec_fused_shape_ok(cfg) ? c.take<float>((size_t)sasrec_fused_slab_floats(cfg, B)) : nullptr;
}

int check_cfg(const iisan_sasrec_cfg* cfg, int64_t B) {
    IISAN_CHECK_SHAPE(B > 0, "sasrec: empty batch");
    IISAN_CHECK_SHAPE(cfg->emb % 64 == 0 && cfg->emb <= MAXE, "sasrec: d_model %d must be a multiple of 64 and <= %d", cfg->emb, MAXE);
    IISAN_CHECK_SHAPE(cfg->heads > 0 && cfg->emb % cfg->heads == 0, "sasrec: heads %d does not divide d_model %d", cfg->heads, cfg->emb);
    IISAN_CHECK_SHAPE(cfg->seq >= 1 && cfg->seq <= 32 && cfg->seq * cfg->heads <= 64, "sasrec: seq %d x heads %d unsupported", cfg->seq, cfg->heads);
    IISAN_CHECK_SHAPE(cfg->emb / cfg->heads <= 64, "sasrec: head width %d > 64 unsupported", cfg->emb / cfg->heads);
    IISAN_CHECK_SHAPE(cfg->blocks >= 1 && cfg->blocks <= 8, "sasrec: %d blocks unsupported", cfg->blocks);
    IISAN_CHECK_SHAPE(cfg->dropout >= 0.f && cfg->dropout < 1.f, "sasrec: dropout %.3f out of range", cfg->dropout);
    return IISAN_OK;
}

Gemm32Prob prob(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int64_t M, int N,
                int64_t K, const float* resid = nullptr, const float* act_src = nullptr) {
    Gemm32Prob p{};
    p.A = A; p.B = B; p.bias = bias; p.resid = resid; p.act_src = act_src; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldc;
    return p;
}

SasFusedPtrs fused_ptrs(const SasBufs& b, int blocks) {
    SasFusedPtrs w{};
    w.Z0 = b.Z0; w.X0 = b.X0;
    for (int l = 0; l < blocks; ++l) {
        const BlockBufs& k = b.blk[l];
        w.Q[l] = k.Q; w.K[l] = k.K; w.V[l] = k.V; w.P[l] = k.P; w.C[l] = k.C; w.Zattn[l] = k.Zattn; w.X1[l] = k.X1;
        w.Hf[l] = k.Hf; w.Zffn[l] = k.Zffn; w.X2[l] = k.X2;
    }
    return w;
}

// parameter table: 0 pos, 1 ln.w, 2 ln.b, then 12 per block:
// +0 wQ +1 wK +2 wV +3 fc +4 attn_ln.w +5 attn_ln.b +6 w1.w +7 w1.b +8 w2.w +9 w2.b +10 ffn_ln.w +11 ffn_ln.b
inline int pb(int l, int i) { return 3 + 12 * l + i; }

}  // namespace

extern "C" size_t iisan_sasrec_ws_bytes(const iisan_sasrec_cfg* cfg, int64_t B) {
    WsCarver c(nullptr, 0);
    SasBufs b;
    carve(c, b, cfg, B);
    return c.off;
}

extern "C" int iisan_sasrec_fwd(const iisan_sasrec_cfg* cfg, const float* x, const float* log_mask, int64_t B,
                                const void* const* params, float* y, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check_cfg(cfg, B));
    WsCarver c(ws, ws_bytes);
    SasBufs b;
    carve(c, b, cfg, B);
    if (c.overflow || !ws) {
        iisan_set_error("sasrec_fwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    if (sasrec_fused_ok(cfg)) return launch_sasrec_fused_fwd(cfg, x, log_mask, B, params, y, fused_ptrs(b, cfg->blocks), s);
    const int S = cfg->seq, E = cfg->emb, H = cfg->heads, dh = E / H;
    const int64_t T = B * S;
    auto W = [&](int i) { return (const float*)params[i]; };
    const dim3 ln_grid((unsigned)ceil_div(T, 4)), blk(256);
    const float pd = cfg->dropout;
    const DropCfg nodrop = make_drop(0, 0, 0.f);
    hipLaunchKernelGGL(add_ln_fwd_kernel, ln_grid, blk, 0, s, x, W(0), (int64_t)S, W(1), W(2), 1e-6f, b.Z0, b.X0, T, E,
                       make_drop(cfg->seed, 0, pd));
    IISAN_LAUNCH_OK();
    const float* xin = b.X0;
    for (int l = 0; l < cfg->blocks; ++l) {
        BlockBufs& k = b.blk[l];
        Gemm32Prob pr[3] = {prob(xin, E, W(pb(l, 0)), E, nullptr, k.Q, E, T, E, E), prob(xin, E, W(pb(l, 1)), E, nullptr, k.K, E, T, E, E),
                            prob(xin, E, W(pb(l, 2)), E, nullptr, k.V, E, T, E, E)};
        IISAN_TRY(launch_gemm32(pr, 3, 0, s));
        hipLaunchKernelGGL(sas_attn_fwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, s, k.Q, k.K, k.V,
                           log_mask, k.P, k.C, B, S, H, dh, make_drop(cfg->seed, 1 + 3 * l, pd));
        IISAN_LAUNCH_OK();
        Gemm32Prob pf = prob(k.C, E, W(pb(l, 3)), E, nullptr, k.Zattn, E, T, E, E, xin);        // x + drop(fc(ctx))
        pf.drop = make_drop(cfg->seed, 2 + 3 * l, pd);
        IISAN_TRY(launch_gemm32(&pf, 1, pd > 0.f ? G32_DROPOUT : 0, s));
        hipLaunchKernelGGL(add_ln_fwd_kernel, ln_grid, blk, 0, s, k.Zattn, (const float*)nullptr, (int64_t)0, W(pb(l, 4)),
                           W(pb(l, 5)), 1e-6f, (float*)nullptr, k.X1, T, E, nodrop);
        IISAN_LAUNCH_OK();
        Gemm32Prob p1 = prob(k.X1, E, W(pb(l, 6)), E, W(pb(l, 7)), k.Hf, 4 * E, T, 4 * E, E);   // relu(W1 x + b1)
        IISAN_TRY(launch_gemm32(&p1, 1, G32_RELU, s));
        Gemm32Prob p2 = prob(k.Hf, 4 * E, W(pb(l, 8)), 4 * E, W(pb(l, 9)), k.Zffn, E, T, E, 4 * E, k.X1);   // x1 + drop(ffn)
        p2.drop = make_drop(cfg->seed, 3 + 3 * l, pd);
        IISAN_TRY(launch_gemm32(&p2, 1, pd > 0.f ? G32_DROPOUT : 0, s));
        float* out = (l == cfg->blocks - 1) ? y : k.X2;
        hipLaunchKernelGGL(add_ln_fwd_kernel, ln_grid, blk, 0, s, k.Zffn, (const float*)nullptr, (int64_t)0, W(pb(l, 10)),
                           W(pb(l, 11)), 1e-6f, (float*)nullptr, out, T, E, nodrop);
        IISAN_LAUNCH_OK();
        xin = out;
    }
    return IISAN_OK;
}

extern "C" int iisan_sasrec_bwd(const iisan_sasrec_cfg* cfg, const float* x, const float* log_mask, int64_t B,
                                const void* const* params, const float* dy, float* dx, void* const* grads, void* ws,
                                size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check_cfg(cfg, B));
    WsCarver c(ws, ws_bytes);
    SasBufs b;
    carve(c, b, cfg, B);
    if (c.overflow || !ws) {
        iisan_set_error("sasrec_bwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    if (sasrec_fused_ok(cfg) && b.slab)
        return launch_sasrec_fused_bwd(cfg, log_mask, B, params, dy, dx, grads, fused_ptrs(b, cfg->blocks), b.slab, s);
    const int S = cfg->seq, E = cfg->emb, H = cfg->heads, dh = E / H;
    const int64_t T = B * S;
    auto W = [&](int i) { return (const float*)params[i]; };
    auto G = [&](int i) { return (float*)grads[i]; };
    const dim3 lnb_grid((unsigned)ceil_div(T, LNB_ROWS)), blk(256);
    const float pd = cfg->dropout;
    const DropCfg nodrop = make_drop(0, 0, 0.f);
    const unsigned ew_grid = (unsigned)(ceil_div(T * E, 256) < 2048 ? ceil_div(T * E, 256) : 2048);
    const float* dcur = dy;        // gradient wrt the current block's output X2
    for (int l = cfg->blocks - 1; l >= 0; --l) {
        BlockBufs& k = b.blk[l];
        const float* xin = l == 0 ? b.X0 : b.blk[l - 1].X2;
        // X2 = LN(Zffn): dZffn -> dA
        hipLaunchKernelGGL(ln_bwd_kernel, lnb_grid, blk, 0, s, k.Zffn, dcur, W(pb(l, 10)), 1e-6f, b.dA, G(pb(l, 10)), G(pb(l, 11)), T, E, nodrop);
        IISAN_LAUNCH_OK();
        // Zffn = X1 + drop(Hf W2^T + b2): the FFN branch sees dA times the forward keep factors
        const float* gF = b.dA;
        if (pd > 0.f) {
            hipLaunchKernelGGL(drop_apply_kernel, dim3(ew_grid), dim3(256), 0, s, b.dA, b.dG, T * E, make_drop(cfg->seed, 3 + 3 * l, pd));
            IISAN_LAUNCH_OK();
            gF = b.dG;
        }
        Gemm32Prob p = prob(gF, E, W(pb(l, 8)), 4 * E, nullptr, b.dH, 4 * E, T, 4 * E, E, nullptr, k.Hf);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB | G32_MUL_RELU_MASK, s));                       // dH = (g·W2) ⊙ [Hf>0]
        p = prob(gF, E, k.Hf, 4 * E, nullptr, G(pb(l, 8)), 4 * E, E, 4 * E, T);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TA | G32_TB | G32_ACCUM, s));                      // dW2 += g^T·Hf
        p = prob(b.dH, 4 * E, k.X1, E, nullptr, G(pb(l, 6)), E, 4 * E, E, T);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TA | G32_TB | G32_ACCUM, s));                      // dW1 += dH^T·X1
        {
            const float* X[2] = {gF, b.dH};
            float* O[2] = {G(pb(l, 9)), G(pb(l, 7))};
            int64_t Ms[2] = {T, T};
            int32_t Ns[2] = {E, 4 * E}, lds[2] = {E, 4 * E};
            IISAN_TRY(launch_colsum(X, O, Ms, Ns, lds, 2, s));                                // db2, db1
        }
        p = prob(b.dH, 4 * E, W(pb(l, 6)), E, nullptr, b.dB, E, T, E, 4 * E, b.dA);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));                                           // dX1 = dA + dH·W1
        // X1 = LN(Zattn): dZattn -> dA
        hipLaunchKernelGGL(ln_bwd_kernel, lnb_grid, blk, 0, s, k.Zattn, b.dB, W(pb(l, 4)), 1e-6f, b.dA, G(pb(l, 4)), G(pb(l, 5)), T, E, nodrop);
        IISAN_LAUNCH_OK();
        // Zattn = xin + drop(C Wfc^T)
        const float* gA = b.dA;
        if (pd > 0.f) {
            hipLaunchKernelGGL(drop_apply_kernel, dim3(ew_grid), dim3(256), 0, s, b.dA, b.dG, T * E, make_drop(cfg->seed, 2 + 3 * l, pd));
            IISAN_LAUNCH_OK();
            gA = b.dG;
        }
        p = prob(gA, E, W(pb(l, 3)), E, nullptr, b.dB, E, T, E, E);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));                                           // dC = g·Wfc
        p = prob(gA, E, k.C, E, nullptr, G(pb(l, 3)), E, E, E, T);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TA | G32_TB | G32_ACCUM, s));                      // dWfc += g^T·C
        hipLaunchKernelGGL(sas_attn_bwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, s, k.Q, k.K, k.V, k.P, b.dB, b.dQ, b.dK, b.dV, S, H, dh,
                           make_drop(cfg->seed, 1 + 3 * l, pd));
        IISAN_LAUNCH_OK();
        {
            Gemm32Prob pr[3] = {prob(b.dQ, E, xin, E, nullptr, G(pb(l, 0)), E, E, E, T), prob(b.dK, E, xin, E, nullptr, G(pb(l, 1)), E, E, E, T),
                                prob(b.dV, E, xin, E, nullptr, G(pb(l, 2)), E, E, E, T)};
            IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));                  // dWq/k/v
        }
        // dxin = dA + dQ·Wq + dK·Wk + dV·Wv   (chained through the residual input, in place in dA)
        p = prob(b.dQ, E, W(pb(l, 0)), E, nullptr, b.dA, E, T, E, E, b.dA);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));
        p = prob(b.dK, E, W(pb(l, 1)), E, nullptr, b.dA, E, T, E, E, b.dA);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));
        float* dst = b.dQ;      // free after the three weight-gradient products
        p = prob(b.dV, E, W(pb(l, 2)), E, nullptr, dst, E, T, E, E, b.dA);
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));
        // dQ now holds d(xin); move it to dK so the next iteration can reuse dQ/dA/dB freely
        IISAN_HIP_OK(hipMemcpyAsync(b.dK, dst, (size_t)T * E * sizeof(float), hipMemcpyDeviceToDevice, s));
        dcur = b.dK;
    }
    // X0 = LN(Z0), Z0 = x + pos
    hipLaunchKernelGGL(ln_bwd_kernel, lnb_grid, blk, 0, s, b.Z0, dcur, W(1), 1e-6f, dx, G(1), G(2), T, E, make_drop(cfg->seed, 0, pd));
    IISAN_LAUNCH_OK();
    {
        const float* X[1] = {dx};
        float* O[1] = {G(0)};
        int64_t Ms[1] = {B};
        int32_t Ns[1] = {S * E}, lds[1] = {S * E};
        IISAN_TRY(launch_colsum(X, O, Ms, Ns, lds, 1, s));                                    // dpos[s,e] = sum_b dx[b,s,e]
    }
    return IISAN_OK;
}
