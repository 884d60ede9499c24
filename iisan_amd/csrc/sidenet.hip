// Intra-/inter-modal Side Adapted Network: forward and backward executors + the fusion kernels.
// Replaces IISANAdaptedMModel.forward of all three reference variants and the AdapterBlock it stacks
// (Code_*/model/modules.py:98-116), plus the backward PyTorch autograd derives:
//   * Uncached  Code_Uncached/model/model.py:209-271      * Cached  Code_Cached/model/model.py:300-349
//   * Versa     Code_Cached_Asym/model/model.py:322-429  (towers of different depth and width: the longer tower first
//     runs its surplus leading SANBs alone — "group layer-drop", :353-378 — then aligned pairs; the wider modality's
//     tap goes through a per-pair Linear(max_dim -> min_dim) — "dim-align", :400-411 — before the inter-modal sum)
//
//   aligned step i (cv block kc = i + diff_cv, text block kt = i + diff_t, mm block i):
//       F_cv = g·tap_v[kc] + (1-g)·cv      F_t = g·tap_t[kt] + (1-g)·text      F_mm = mm + g·a_v + (1-g)·a_t
//       state_z = Wu_z·act(Wd_z·F_z + bd_z) + bu_z + F_z                 (a_v / a_t = tap or its dim-aligned projection)
//   E_z = head_z(fc_z(state_z))                                          -> item3 = [E_cv | E_text | E_mm]
//
// All fp32.  Every GEMM / column-sum / fusion step of the (up to) three towers is ONE launch with several problems
// (per-problem shapes, so Versa's 8192-wide text tower and 1024-wide image tower share launches too).  Fusion kernels
// are HBM-bound (16-byte lanes, taps read in place from the [M, L, D] tap tensor); GEMMs run on the f32 matrix cores
// (gemm32.hip), except the LARGE Linear layers (fc_z 768x768 at Cached batch sizes, Versa's 8192 -> 1024 dim-align;
// forward, dX and dW), which run as split-operand fp16 MFMA GEMMs (split.hip: ~fp32 accuracy at 1/3 of the 16-bit rate).
// Saved for backward: F, pre-activation U, activation, state per step, the dim-aligned taps.
#include "common.h"

int launch_colsum(const float* const* X, float* const* out, const int64_t* M, const int32_t* N, const int32_t* ld,
                  int nprob, hipStream_t s);

// fused SANB step (sanb.hip)
struct SanbTowerDesc {
    const float* a; const float* b; const float* prev; int64_t lda, ldb, ldp; const float* gate; int32_t D, type;
    const float* Wd; const float* bd; const float* Wu; const float* bu;
    float* F; float* U; float* A; float* O;
    const float* dO; const float* Upre; float* dU; float* dprev; float* da; float* db; float* dgate; float* dbu; float* dbd;
};
bool sanb_fused_ok(int D, int down);
int launch_sanb_fwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s);
int launch_sanb_bwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s);
int launch_sanb_transpose(const float* const* in, float* const* out, const int32_t* rows, const int32_t* cols, int n, hipStream_t s);

double gemm_x3_get_min_flops();     // split.hip

namespace {

// one tower of one fusion step.  type 0: F = g·a + (1-g)·prev ; type 1: F = prev + g·a + (1-g)·b  (not gated: plain sums)
struct FuseTower {
    const float* a; const float* b; const float* prev;      // row r at a + r*lda (floats); prev null = zeros
    int64_t lda, ldb, ldp;
    const float* gate;                                       // device scalar theta, null = not gated
    float* F;                                                // fwd: output [M,D] ; bwd: dF in, dprev out (in place)
    float* da; float* db;                                    // bwd only: optional gradients wrt a / b ([M,D])
    float* dgate;                                            // bwd only: accumulates d theta
    int32_t D, type;
};
struct FuseArgs {
    FuseTower t[3];
    int64_t M;
};

__device__ __forceinline__ float gate_of(const float* theta) { return 1.0f / (1.0f + __expf(-theta[0] / 0.1f)); }

__global__ __launch_bounds__(256) void fuse_fwd_kernel(FuseArgs args) {
    const FuseTower& t = args.t[blockIdx.y];
    const int d4 = t.D / 4;
    const int64_t total = args.M * d4;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;
    // (row, column) advanced incrementally: a 64-bit division per element was most of this kernel's time
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    const int64_t dm = step / d4;
    const int dc = (int)(step - dm * d4);
    int64_t m = i0 / d4;
    int c4 = (int)(i0 - m * d4);
    for (int64_t i = i0; i < total; i += step, m += dm, c4 += dc) {
        if (c4 >= d4) { c4 -= d4; ++m; }
        const int c = c4 * 4;
        const f4 av = *(const f4*)(t.a + m * t.lda + c);
        f4 pv = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
        if (t.prev) pv = *(const f4*)(t.prev + m * t.ldp + c);
        if (t.type == 1) bv = *(const f4*)(t.b + m * t.ldb + c);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (t.type == 0) o[e] = gated ? g * av[e] + (1.f - g) * pv[e] : av[e] + pv[e];
            else o[e] = gated ? pv[e] + g * av[e] + (1.f - g) * bv[e] : pv[e] + av[e] + bv[e];
        }
        *(f4*)(t.F + m * t.D + c) = o;
    }
}

// in: dF (in t.F).  out: dprev in place ((1-g)·dF for type 0, dF for type 1), optional da = g·dF, db = (1-g)·dF,
// dtheta += [sum dF ⊙ (a - prev | a - b)] · g(1-g)/0.1
__global__ __launch_bounds__(256) void fuse_bwd_kernel(FuseArgs args) {
    __shared__ float red[4];
    const FuseTower& t = args.t[blockIdx.y];
    const int d4 = t.D / 4;
    const int64_t total = args.M * d4;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;
    float part = 0.f;
    // (row, column) advanced incrementally: a 64-bit division per element was most of this kernel's time
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    const int64_t dm = step / d4;
    const int dc = (int)(step - dm * d4);
    int64_t m = i0 / d4;
    int c4 = (int)(i0 - m * d4);
    // CH iterations' loads are issued together (one HBM round trip per CH iterations instead of one per iteration: the store
    // at the end of an iteration keeps the compiler from hoisting the next one's loads; 127 us for Versa's 184 MB before)
    constexpr int CH = 4;
    const float ca = gated ? g : 1.f, cb = gated ? 1.f - g : 1.f;
    for (int64_t i = i0; i < total; i += CH * step) {
        f4 df[CH], av[CH], ov[CH];
        int64_t mm[CH];
        int cc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (c4 >= d4) { c4 -= d4; ++m; }
            mm[j] = m; cc[j] = c4 * 4;
            df[j] = av[j] = ov[j] = (f4){0.f, 0.f, 0.f, 0.f};
            if (i + j * step < total) {
                df[j] = *(const f4*)(t.F + m * t.D + cc[j]);
                if (gated) {
                    av[j] = *(const f4*)(t.a + m * t.lda + cc[j]);
                    if (t.type == 1) ov[j] = *(const f4*)(t.b + m * t.ldb + cc[j]);
                    else if (t.prev) ov[j] = *(const f4*)(t.prev + m * t.ldp + cc[j]);
                }
            }
            m += dm; c4 += dc;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (i + j * step >= total) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) part += df[j][e] * (av[j][e] - ov[j][e]);
            const int64_t o = mm[j] * t.D + cc[j];
            if (t.da) *(f4*)(t.da + o) = (f4){ca * df[j][0], ca * df[j][1], ca * df[j][2], ca * df[j][3]};
            if (t.db) *(f4*)(t.db + o) = (f4){cb * df[j][0], cb * df[j][1], cb * df[j][2], cb * df[j][3]};
            if (t.type == 0 && gated) *(f4*)(t.F + o) = (f4){cb * df[j][0], cb * df[j][1], cb * df[j][2], cb * df[j][3]};
        }
    }
    if (!gated) return;
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(t.dgate, (red[0] + red[1] + red[2] + red[3]) * g * (1.f - g) / 0.1f);
}

// ---- static description of one configuration -----------------------------------------------------------------
struct Plan {
    int n[3];            // SANBs per tower (cv, text, mm)
    int D[3];            // tower widths
    int H[3];            // width between fc_z and head_z
    int r, E;
    int diff_cv, diff_t; // surplus leading blocks of the longer tower
    bool align;          // dim-align projections present
    bool text_wide;      // which tap is projected
    // parameter table offsets
    int p_adapter[3], p_gate[3], p_dp, p_fc[3], p_head[3], n_params;
    int wd(int z, int k) const { return p_adapter[z] + 4 * k; }
    int gate(int z, int k) const { return p_gate[z] + k; }
    int dpw(int i) const { return p_dp + 2 * i; }
};

int make_plan(const iisan_side_cfg* c, Plan& p) {
    IISAN_CHECK_SHAPE(c->n_side >= 1 && c->n_side <= IISAN_MAX_SIDE, "side_net: n_side %d out of range", c->n_side);
    const int n_t = c->versa ? c->n_side_text : c->n_side;
    IISAN_CHECK_SHAPE(n_t >= 1 && n_t <= IISAN_MAX_SIDE, "side_net: n_side_text %d out of range", n_t);
    IISAN_CHECK_SHAPE(c->dim_cv % 4 == 0 && c->dim_text % 4 == 0 && c->down % 4 == 0 && c->emb % 4 == 0,
                      "side_net: widths must be multiples of 4");
    IISAN_CHECK_SHAPE(c->versa || c->dim_cv == c->dim_text, "side_net: towers of different width (%d vs %d) need versa = 1 "
                      "(Code_Cached_Asym)", c->dim_cv, c->dim_text);
    p.n[0] = c->n_side; p.n[1] = n_t; p.n[2] = p.n[0] < p.n[1] ? p.n[0] : p.n[1];
    p.D[0] = c->dim_cv; p.D[1] = c->dim_text; p.D[2] = c->dim_cv < c->dim_text ? c->dim_cv : c->dim_text;
    p.r = c->down; p.E = c->emb;
    p.diff_cv = p.n[0] > p.n[1] ? p.n[0] - p.n[1] : 0;
    p.diff_t = p.n[1] > p.n[0] ? p.n[1] - p.n[0] : 0;
    p.align = c->versa && c->dim_cv != c->dim_text;
    p.text_wide = c->dim_text > c->dim_cv;
    if (c->versa) { p.H[0] = p.E; p.H[1] = p.E; p.H[2] = p.D[2]; }
    else { p.H[0] = p.D[0]; p.H[1] = p.D[1]; p.H[2] = p.D[2]; }
    int o = 0;
    for (int z = 0; z < 3; ++z) { p.p_adapter[z] = o; o += 4 * p.n[z]; }
    for (int z = 0; z < 3; ++z) { p.p_gate[z] = o; o += p.n[z]; }
    p.p_dp = o; if (p.align) o += 2 * p.n[2];
    for (int z = 0; z < 3; ++z) { p.p_fc[z] = o; o += 2; }
    for (int z = 0; z < 3; ++z) { p.p_head[z] = o; o += 2; }
    p.n_params = o;
    for (int k = 0; k < p.n[0]; ++k)
        IISAN_CHECK_SHAPE(c->tap_index[k] >= 0 && c->tap_index[k] < c->tap_stride_cv, "side_net: cv tap index %d outside the tap tensor", c->tap_index[k]);
    for (int k = 0; k < p.n[1]; ++k) {
        const int ti = c->versa ? c->tap_index_text[k] : c->tap_index[k];
        IISAN_CHECK_SHAPE(ti >= 0 && ti < c->tap_stride_text, "side_net: text tap index %d outside the tap tensor", ti);
    }
    return IISAN_OK;
}

struct SideBufs {
    float* F[IISAN_MAX_SIDE][3];
    float* U[IISAN_MAX_SIDE][3];
    float* A[IISAN_MAX_SIDE][3];     // activation(U)
    float* O[IISAN_MAX_SIDE][3];     // state after SANB k
    float* DP[IISAN_MAX_SIDE];       // dim-aligned taps [M, d]
    float* Y[3];                     // fc_z(state)
    float* dO[3];                    // backward scratch [M, D_z]
    float* dY[3];                    // [M, H_z]
    float* dU[3];                    // [M, down]
    float* dDP[IISAN_MAX_SIDE];      // [M, d]: gradient wrt the dim-aligned tap of every aligned step (kept: their weight-gradient products run as one group after the chain)
    void* x3; size_t x3_bytes;       // scratch of the split-operand GEMM (operand images + scales), null = not used
    float* WT[3][2];                 // backward of the fused SANB step: fc_down^T [D, 64] and fc_up^T [64, D] of the current step
    float* skws; size_t skws_floats; // split-K scratch of the skinny long-K products (gemm32_set_scratch)
    // amax slots shared by the split-operand products that read the same tensor: [0..2] final tower state O_z, [3..5] fc weight,
    // [9 + i] the wide tap of dim-align step i (6..8 unused).  Forward zeroes and fills them, backward reuses them (same
    // workspace, tensors unchanged); the amax of dY_z lives in the backward block zb[0..2].
    uint32_t* amax;
    // per-product scratch words of the split-operand GEMM (private amax, 1/scale, lo-plane flags: 12 words each), zeroed by the
    // SAME memset as the amax slots of the call instead of one 48-byte memset per product (a fill kernel is ~5 us: 23 of them per
    // Versa step).  Forward block: amax[16 + MAX_SIDE] | slots; backward block (zb): dY amax [3] + pad | slots.
    uint32_t* x3z_f; uint32_t* zb; uint32_t* x3z_b;
    mutable uint32_t* x3z_next; mutable int x3z_left;
};
constexpr int X3Z_WORDS = 12, X3Z_SLOTS_F = IISAN_MAX_SIDE + 4, X3Z_SLOTS_B = IISAN_MAX_SIDE + 8, ZB_HEAD = 4;

// 1 (default): the large Linear layers (fc_z, Versa dim-align; forward, dX and dW) run as split-operand fp16 MFMA GEMMs
// (split.hip); 0: everything on the f32-input matrix cores (gemm32.hip).  Test / bench knob.
int g_use_x3 = 1;
// A SANB step whose active towers share a supported width can run as ONE fused launch per direction (sanb.hip) instead of the
// fusion kernel + separate products.  1 (default): fused below SANB_FUSED_MAX_ROWS item slots, unfused from there on — same-box
// A/B with the K = 64 and weight-gradient kernels of gemm32.hip in place: M = 1,408 (Uncached, bs = 128) fused 66.22 vs 66.37 ms,
// M = 4,373 (Cached on distinct ids) unfused 4.33 vs 4.57 ms, M = 11,264 (Cached) unfused 5.95 vs 6.00 ms; 2: always fused;
// 0: never.  Test / bench knob.
constexpr int64_t SANB_FUSED_MAX_ROWS = 4096;
int g_use_sanb = 1;
int g_dw_merge = 1;                // 1 (default): dWu and dWd of a SANB step in one launch; 0: two (A/B knob)

// what decides which products take the split-operand route: the on/off knob and the FLOP threshold, as one comparable word
uint64_t x3_route_word() {
    const float f = (float)gemm_x3_get_min_flops();
    uint32_t bits;
    memcpy(&bits, &f, 4);
    return ((uint64_t)(g_use_x3 ? 1 : 0) << 32) | bits;
}

size_t x3_need(const Plan& p, int64_t M) {
    if (!g_use_x3) return 0;
    // the products of a launch group build their operand images together (split.hip: launch_gemm_x3_group): the workspace holds a whole group
    size_t need = 0;
    int64_t gm[IISAN_MAX_SIDE], gn[IISAN_MAX_SIDE], gk[IISAN_MAX_SIDE];
    int ng = 0;
    auto consider = [&](int64_t m, int64_t n, int64_t k) {
        Gemm32Prob q{};
        q.M = m; q.N = (int32_t)n; q.K = k; q.lda = q.ldb = q.ldc = q.ldr = 4;
        if (gemm_x3_applicable(q, 0) && ng < IISAN_MAX_SIDE) { gm[ng] = m; gn[ng] = n; gk[ng] = k; ++ng; }
    };
    auto close_group = [&]() {
        for (int i = 0; i < ng; i += 8) {
            const size_t b = gemm_x3_group_ws_bytes(gm + i, gn + i, gk + i, ng - i < 8 ? ng - i : 8);
            if (b > need) need = b;
        }
        ng = 0;
    };
    for (int z = 0; z < 3; ++z) consider(M, p.H[z], p.D[z]);      // Y = O Wf^T
    close_group();
    for (int z = 0; z < 3; ++z) consider(M, p.D[z], p.H[z]);      // dO = dY Wf
    close_group();
    for (int z = 0; z < 3; ++z) consider(p.H[z], p.D[z], M);      // dWf += dY^T O
    close_group();
    if (p.align) {
        const int dw = p.text_wide ? p.D[1] : p.D[0];
        for (int i = 0; i < p.n[2]; ++i) consider(M, p.D[2], dw);          // DP = tap Pd^T
        close_group();
        for (int i = 0; i < p.n[2]; ++i) consider(p.D[2], dw, M);          // dPd += dDP^T tap
        close_group();
    }
    return need;
}

void carve(WsCarver& c, SideBufs& b, const Plan& p, int64_t M) {
    for (int z = 0; z < 3; ++z)
        for (int k = 0; k < p.n[z]; ++k) {
            b.F[k][z] = c.take<float>((size_t)M * p.D[z]);
            b.U[k][z] = c.take<float>((size_t)M * p.r);
            b.A[k][z] = c.take<float>((size_t)M * p.r);
            b.O[k][z] = c.take<float>((size_t)M * p.D[z]);
        }
    for (int i = 0; i < p.n[2]; ++i) b.DP[i] = p.align ? c.take<float>((size_t)M * p.D[2]) : nullptr;
    for (int z = 0; z < 3; ++z) {
        b.Y[z] = c.take<float>((size_t)M * p.H[z]);
        b.dO[z] = c.take<float>((size_t)M * p.D[z]);
        b.dY[z] = c.take<float>((size_t)M * p.H[z]);
        b.dU[z] = c.take<float>((size_t)M * p.r);
    }
    for (int i = 0; i < p.n[2]; ++i) b.dDP[i] = p.align ? c.take<float>((size_t)M * p.D[2]) : nullptr;
    for (int z = 0; z < 3; ++z) {
        b.WT[z][0] = c.take<float>((size_t)p.D[z] * p.r);
        b.WT[z][1] = c.take<float>((size_t)p.D[z] * p.r);
    }
    b.skws_floats = (size_t)3 * 8 * align_up((size_t)M * p.r, 64);     // three towers x 8 K-splits x [M, down]
    {   // ... and the split-K partials of the adapters' weight-gradient products: three towers x <= 32 splits x [D, down]
        size_t wg = 0;
        for (int z = 0; z < 3; ++z) wg += (size_t)2 * 32 * align_up((size_t)p.D[z] * p.r + p.D[z], 64);    // (dWu and dWd of a step share a launch; + the column-sum partials)
        if (wg > b.skws_floats) b.skws_floats = wg;
    }
    b.skws = c.take<float>(b.skws_floats);
    {   // one carve per direction: [amax | slots] and [dY amax | slots] are each zeroed by a single memset
        uint32_t* zf = c.take<uint32_t>(16 + IISAN_MAX_SIDE + X3Z_WORDS * X3Z_SLOTS_F);
        b.amax = zf; b.x3z_f = zf ? zf + 16 + IISAN_MAX_SIDE : nullptr;
        b.zb = c.take<uint32_t>(ZB_HEAD + X3Z_WORDS * X3Z_SLOTS_B);
        b.x3z_b = b.zb ? b.zb + ZB_HEAD : nullptr;
        b.x3z_next = nullptr; b.x3z_left = 0;
    }
    b.x3_bytes = x3_need(p, M);
    b.x3 = b.x3_bytes ? (void*)c.take<char>(b.x3_bytes) : nullptr;
}

// the problems of one launch group: the large ones as split-operand GEMMs, the rest together on the f32 matrix cores
int gemm_group(const Gemm32Prob* pr, int n, int flags, const SideBufs& b, hipStream_t s) {
    Gemm32Prob rest[IISAN_MAX_SIDE], big[IISAN_MAX_SIDE];
    int nr = 0, nb = 0;
    IISAN_CHECK_SHAPE(n <= IISAN_MAX_SIDE, "gemm_group: %d problems", n);
    for (int i = 0; i < n; ++i) {
        if (b.x3 && gemm_x3_applicable(pr[i], flags)) {
            Gemm32Prob q = pr[i];
            if (b.x3z_left > 0) { q.x3_zeroed = b.x3z_next; b.x3z_next += X3Z_WORDS; --b.x3z_left; }
            big[nb++] = q;
        }
        else rest[nr++] = pr[i];
    }
    // Round 6: the amax passes of every operand the group's split-operand products still need go out as ONE launch (they were one
    // launch per operand and product: 5 - 16 us each, latency-bound).  An operand without a caller-owned slot uses the first / second
    // word of the product's private zeroed block (what launch_gemm_x3 would have used itself); products without one keep their own pass.
    if (nb > 1 || (nb == 1 && big[0].x3_zeroed)) {
        AmaxBatch ab{};
        int na = 0;
        auto want = [&](const float* x, bool trans, int64_t op_rows, int64_t K, int64_t ld, uint32_t* slot) {
            for (int k = 0; k < na; ++k) if (ab.out[k] == slot) return;            // the same tensor twice in one group
            if (na >= 16) return;
            ab.x[na] = x; ab.rows[na] = trans ? K : op_rows; ab.cols[na] = trans ? op_rows : K; ab.ld[na] = ld; ab.out[na] = slot;
            ++na;
        };
        for (int i = 0; i < nb; ++i) {
            Gemm32Prob& q = big[i];
            if (!q.x3_zeroed && !(q.amax_a && q.amax_b)) continue;
            uint32_t* pa = q.amax_a ? q.amax_a : q.x3_zeroed;
            uint32_t* pb = q.amax_b ? q.amax_b : q.x3_zeroed + 1;
            const bool full = na + 2 > 16;
            if (full) break;
            if (!q.amax_a_ready) want(q.A, (flags & G32_TA) != 0, q.M, q.K, q.lda, pa);
            if (!q.amax_b_ready) want(q.B, (flags & G32_TB) != 0, q.N, q.K, q.ldb, pb);
            q.amax_a = pa; q.amax_b = pb; q.amax_a_ready = 1; q.amax_b_ready = 1;
        }
        IISAN_TRY(launch_amax_batch(ab, na, s));
    }
    const int gmax = gemm_x3_group_max();
    for (int i = 0; i < nb; i += gmax) IISAN_TRY(launch_gemm_x3_group(big + i, nb - i < gmax ? nb - i : gmax, flags, b.x3, b.x3_bytes, s));
    // A skinny long-K product (Versa's fc_bert: [1408, 8192] -> 64) must not share a launch with short-K ones: alone it takes the
    // split-K route through the executor's scratch (33 us); grouped, the launch has "enough" workgroups, nothing is split and its
    // 22 workgroups walk 128 K-tiles each while the rest of the chip idles (195 us per Versa step, profiles/r3a_versa_*).
    if (nr > 1 && !(flags & G32_ACCUM)) {
        int64_t kmin = INT64_MAX;
        for (int i = 0; i < nr; ++i) kmin = rest[i].K < kmin ? rest[i].K : kmin;
        for (int i = 0; i < nr;) {
            if (rest[i].N <= 64 && rest[i].K >= 4096 && rest[i].K >= 4 * kmin) {
                IISAN_TRY(launch_gemm32(&rest[i], 1, flags & ~G32_HINT_B_EXACT16, s));
                for (int k = i; k + 1 < nr; ++k) rest[k] = rest[k + 1];
                --nr;
            } else ++i;
        }
    }
    for (int i = 0; i < nr; i += 4)          // (a launch takes four problems: Gemm32Batch)
        IISAN_TRY(launch_gemm32(rest + i, nr - i < 4 ? nr - i : 4, flags & ~G32_HINT_B_EXACT16, s));
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

// which towers take part in global step g (0 .. diff + n_mm - 1) and with which block index
struct StepMap {
    int nact; int z[3]; int k[3]; int mm_i;     // mm_i = -1 when the step is a surplus (single-tower) step
};
StepMap step_map(const Plan& p, int g) {
    StepMap s{};
    const int diff = p.diff_cv + p.diff_t;
    if (g < diff) {
        s.nact = 1; s.z[0] = p.diff_cv ? 0 : 1; s.k[0] = g; s.mm_i = -1;
    } else {
        const int i = g - diff;
        s.nact = 3; s.mm_i = i;
        s.z[0] = 0; s.k[0] = i + p.diff_cv;
        s.z[1] = 1; s.k[1] = i + p.diff_t;
        s.z[2] = 2; s.k[2] = i;
    }
    return s;
}

struct ScratchGuard {       // registers the executor's split-K scratch with gemm32 for the duration of a call
    ScratchGuard(float* p, size_t n) { gemm32_set_scratch(p, n); }
    ~ScratchGuard() { gemm32_set_scratch(nullptr, 0); }
};

struct Ctx {
    const iisan_side_cfg* cfg; Plan p; SideBufs b;
    const float* taps[2]; int64_t M;
    const void* const* params;
    const float* W(int i) const { return (const float*)params[i]; }
    int tap_idx(int z, int k) const { return (z == 1 && cfg->versa) ? cfg->tap_index_text[k] : cfg->tap_index[k]; }
    int64_t tap_ld(int z) const { return (int64_t)(z == 0 ? cfg->tap_stride_cv : cfg->tap_stride_text) * p.D[z]; }
    const float* tap(int z, int k) const { return taps[z] + (int64_t)tap_idx(z, k) * p.D[z]; }
    int first_idx(int z) const { return (z == 1 && cfg->versa) ? cfg->first_index_text : cfg->first_index; }
    // fusion operands of tower z at its block k (mm: k = i)
    void fuse_operands(FuseTower& t, int z, int k, const StepMap& sm) const {
        t.D = p.D[z];
        t.gate = cfg->gated ? W(p.gate(z, k)) : nullptr;
        t.prev = nullptr; t.ldp = p.D[z];
        if (k > 0) t.prev = b.O[k - 1][z];
        else if (cfg->remove_first && z != 2) { t.prev = taps[z] + (int64_t)first_idx(z) * p.D[z]; t.ldp = tap_ld(z); }
        if (z != 2) {
            t.type = 0; t.a = tap(z, k); t.lda = tap_ld(z); t.b = nullptr; t.ldb = 0;
        } else {
            t.type = 1;
            const int kc = sm.k[0], kt = sm.k[1];
            t.a = tap(0, kc); t.lda = tap_ld(0);
            t.b = tap(1, kt); t.ldb = tap_ld(1);
            if (p.align) {
                if (p.text_wide) { t.b = b.DP[k]; t.ldb = p.D[2]; }
                else { t.a = b.DP[k]; t.lda = p.D[2]; }
            }
        }
    }
};

// the active towers of a step can share one fused launch
bool step_fusable(const Plan& p, const StepMap& sm, int64_t M) {
    if (!g_use_sanb || (g_use_sanb == 1 && M >= SANB_FUSED_MAX_ROWS)) return false;
    const int D = p.D[sm.z[0]];
    for (int a = 0; a < sm.nact; ++a)
        if (p.D[sm.z[a]] != D) return false;
    return sanb_fused_ok(D, p.r);
}

void tower_desc(SanbTowerDesc& d, const Ctx& c, int z, int k, const StepMap& sm) {
    FuseTower ft{};
    c.fuse_operands(ft, z, k, sm);
    d = SanbTowerDesc{};
    d.a = ft.a; d.b = ft.b; d.prev = ft.prev; d.lda = ft.lda; d.ldb = ft.ldb; d.ldp = ft.ldp; d.gate = ft.gate;
    d.D = ft.D; d.type = ft.type;
}

int setup(Ctx& c, const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
          const void* const* params, void* ws, size_t ws_bytes, const char* who) {
    IISAN_CHECK_SHAPE(M > 0, "side_net: M must be positive");
    c.cfg = cfg;
    IISAN_TRY(make_plan(cfg, c.p));
    WsCarver w(ws, ws_bytes);
    carve(w, c.b, c.p, M);
    if (w.overflow || !ws) {
        iisan_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, w.off);
        return IISAN_EWORKSPACE;
    }
    c.taps[0] = taps_cv; c.taps[1] = taps_text; c.M = M; c.params = params;
    return IISAN_OK;
}

}  // namespace

IISAN_DEV_KNOB(sanb_fused, g_use_sanb);
IISAN_DEV_KNOB(sidenet_dw_merge, g_dw_merge);

void gemm_x3_set_min_flops(double f);     // negative = the library default (split.hip: X3_DEFAULT_MIN_FLOPS, 4 GFLOP)
// 0 = off, 1 = library default (the state of a process that never calls this knob), 2 = every product whose shape allows
// it (tests: the small golden fixtures then run through the split-operand path too)
static int g_x3_mode = 1;
IISAN_DEV_KNOB_FN(x3, g_x3_mode, (g_x3_mode = (int)v, g_use_x3 = v != 0, gemm_x3_set_min_flops(v == 2 ? 0.0 : -1.0)));

extern "C" size_t iisan_side_net_ws_bytes(const iisan_side_cfg* cfg, int64_t M) {
    Plan p;
    if (make_plan(cfg, p) != IISAN_OK) return 0;
    WsCarver c(nullptr, 0);
    SideBufs b;
    carve(c, b, p, M);
    return c.off;
}

extern "C" int32_t iisan_side_net_num_params(const iisan_side_cfg* cfg) {
    Plan p;
    if (make_plan(cfg, p) != IISAN_OK) return -1;
    return p.n_params;
}

extern "C" int iisan_side_net_fwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                                  const void* const* params, float* item3, void* ws, size_t ws_bytes, uint64_t* fwd_token,
                                  void* stream) {
    hipStream_t s = (hipStream_t)stream;
    Ctx c;
    IISAN_TRY(setup(c, cfg, taps_cv, taps_text, M, params, ws, ws_bytes, "side_net_fwd"));
    const Plan& p = c.p;
    SideBufs& b = c.b;
    ScratchGuard guard(b.skws, b.skws_floats);
    const int act_flag = cfg->gelu ? G32_GELU : G32_RELU;
    const int nsteps = p.diff_cv + p.diff_t + p.n[2];
    IISAN_CHECK_SHAPE(fwd_token != nullptr, "side_net_fwd: fwd_token must not be null");
    *fwd_token = 0x51DE000000000000ull ^ x3_route_word();     // the backward call must see the same split-operand routing
    IISAN_HIP_OK(hipMemsetAsync(b.amax, 0, (size_t)(16 + IISAN_MAX_SIDE + X3Z_WORDS * X3Z_SLOTS_F) * sizeof(uint32_t), s));
    b.x3z_next = b.x3z_f; b.x3z_left = X3Z_SLOTS_F;
    // taps known to be exact in fp16 (cfg->taps_exact16): their amax slots are PRESET to 8192.0f, which the split kernels turn into
    // scale 1 (scale_of: amax in [2^13, 2^14) -> 2^0; an fp16 value cannot exceed 65504, so nothing overflows), and marked ready —
    // the dim-align products skip the tap's amax pass (15 us each at Versa's [1408, 8192]), forward and weight gradient.
    const bool taps_preset = cfg->taps_exact16 && p.align && p.n[2] > 0;
    if (taps_preset) IISAN_HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(b.amax + 9), 0x46000000, (size_t)p.n[2], s));
    if (p.align && p.n[2] > 0) {
        // dim-align the wider modality's taps (Code_Cached_Asym/model/model.py:404-411).  The products depend on taps and weights only, not on
        // the chain: round 6 issues them as ONE group ahead of it (they used to sit in front of their steps, one amax launch per weight each)
        Gemm32Prob pd[IISAN_MAX_SIDE];
        const int zw = p.text_wide ? 1 : 0;
        for (int i = 0; i < p.n[2]; ++i) {
            const StepMap sm = step_map(p, p.diff_cv + p.diff_t + i);
            pd[i] = prob(c.tap(zw, sm.k[zw]), (int)c.tap_ld(zw), c.W(p.dpw(i)), p.D[zw], c.W(p.dpw(i) + 1), b.DP[i], p.D[2], M, p.D[2], p.D[zw]);
            pd[i].amax_a = b.amax + 9 + i;           // the tap's amax: read again by the weight-gradient product of this step
            pd[i].amax_a_ready = taps_preset ? 1 : 0;
            pd[i].exact16_a = taps_preset ? 1 : 0;
        }
        IISAN_TRY(gemm_group(pd, p.n[2], 0, b, s));
    }
    for (int g = 0; g < nsteps; ++g) {
        const StepMap sm = step_map(p, g);
        if (step_fusable(p, sm, M)) {         // fusion + down + activation + up of every active tower in one launch
            SanbTowerDesc td[3];
            for (int a = 0; a < sm.nact; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                tower_desc(td[a], c, z, k, sm);
                // both weights as stored: fc_down [64, D] and fc_up [D, 64] have the contraction index contiguous
                td[a].Wd = c.W(p.wd(z, k)); td[a].bd = c.W(p.wd(z, k) + 1); td[a].Wu = c.W(p.wd(z, k) + 2); td[a].bu = c.W(p.wd(z, k) + 3);
                td[a].F = b.F[k][z]; td[a].U = b.U[k][z]; td[a].A = b.A[k][z]; td[a].O = b.O[k][z];
            }
            IISAN_TRY(launch_sanb_fwd(td, sm.nact, M, cfg->gelu, s));
            continue;
        }
        FuseArgs fa{};
        fa.M = M;
        int maxD = 0;
        for (int a = 0; a < sm.nact; ++a) {
            c.fuse_operands(fa.t[a], sm.z[a], sm.k[a], sm);
            fa.t[a].F = b.F[sm.k[a]][sm.z[a]];
            if (p.D[sm.z[a]] > maxD) maxD = p.D[sm.z[a]];
        }
        Gemm32Prob pr[3];
        // fusion + down projection + activation in one launch where the shapes allow (gemm32_n64f_kernel: F is formed in the
        // registers that feed the product and written once); else the fusion kernel, then the product
        bool fed = false;
        if (p.r == 64) {
            N64FDesc nd[3];
            for (int a = 0; a < sm.nact; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                const FuseTower& ft = fa.t[a];
                nd[a] = N64FDesc{};
                nd[a].a = ft.a; nd[a].b = ft.b; nd[a].prev = ft.prev; nd[a].lda = ft.lda; nd[a].ldb = ft.ldb; nd[a].ldp = ft.ldp;
                nd[a].gate = ft.gate; nd[a].type = ft.type;
                nd[a].F = b.F[k][z]; nd[a].W = c.W(p.wd(z, k)); nd[a].ldw = p.D[z]; nd[a].bias = c.W(p.wd(z, k) + 1);
                nd[a].U = b.U[k][z]; nd[a].A = b.A[k][z]; nd[a].M = M; nd[a].K = p.D[z];
            }
            if (gemm32_n64f_ok(nd, sm.nact)) {
                IISAN_TRY(launch_gemm32_n64f(nd, sm.nact, cfg->gelu, s));
                fed = true;
            }
        }
        if (!fed) {
            hipLaunchKernelGGL(fuse_fwd_kernel, dim3(ew_grid(M, maxD), sm.nact), dim3(256), 0, s, fa);
            IISAN_LAUNCH_OK();
            for (int a = 0; a < sm.nact; ++a) {   // U = F Wd^T + bd (saved), A = act(U)
                const int z = sm.z[a], k = sm.k[a];
                pr[a] = prob(b.F[k][z], p.D[z], c.W(p.wd(z, k)), p.D[z], c.W(p.wd(z, k) + 1), b.A[k][z], p.r, M, p.r, p.D[z], nullptr, 0, b.U[k][z]);
            }
            IISAN_TRY(launch_gemm32(pr, sm.nact, act_flag | G32_PREACT, s));
        }
        for (int a = 0; a < sm.nact; ++a) {   // state = A Wu^T + bu + F
            const int z = sm.z[a], k = sm.k[a];
            pr[a] = prob(b.A[k][z], p.r, c.W(p.wd(z, k) + 2), p.r, c.W(p.wd(z, k) + 3), b.O[k][z], p.D[z], M, p.D[z], p.r, b.F[k][z], p.D[z]);
        }
        IISAN_TRY(launch_gemm32(pr, sm.nact, 0, s));
    }
    Gemm32Prob pr[3];
    for (int z = 0; z < 3; ++z) {
        pr[z] = prob(b.O[p.n[z] - 1][z], p.D[z], c.W(p.p_fc[z]), p.D[z], c.W(p.p_fc[z] + 1), b.Y[z], p.H[z], M, p.H[z], p.D[z]);
        pr[z].amax_a = b.amax + z;                   // O_z and the fc weight: their amax is left for the backward products
        pr[z].amax_b = b.amax + 3 + z;
    }
    IISAN_TRY(gemm_group(pr, 3, 0, b, s));
    for (int z = 0; z < 3; ++z) pr[z] = prob(b.Y[z], p.H[z], c.W(p.p_head[z]), p.H[z], c.W(p.p_head[z] + 1), item3 + z * p.E, 3 * p.E, M, p.E, p.H[z]);
    IISAN_TRY(launch_gemm32(pr, 3, 0, s));
    return IISAN_OK;
}

extern "C" int iisan_side_net_bwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                                  const void* const* params, const float* d_item3, void* const* grads, void* ws,
                                  size_t ws_bytes, uint64_t fwd_token, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    Ctx c;
    {   // (before setup(): a changed routing also changes the workspace layout.)  The amax slots this call marks "ready" were filled by the forward call only on the routes IT took
        if ((fwd_token >> 48) != 0x51DE) {
            iisan_set_error("side_net_bwd: fwd_token %llx did not come from side_net_fwd", (unsigned long long)fwd_token);
            return IISAN_EBADSHAPE;
        }
        if ((fwd_token ^ 0x51DE000000000000ull) != x3_route_word()) {
            iisan_set_error("side_net_bwd: the split-operand routing (iisan_set_x3) changed since side_net_fwd filled this workspace");
            return IISAN_EBADSHAPE;
        }
    }
    IISAN_TRY(setup(c, cfg, taps_cv, taps_text, M, params, ws, ws_bytes, "side_net_bwd"));
    const Plan& p = c.p;
    SideBufs& b = c.b;
    ScratchGuard guard(b.skws, b.skws_floats);
    const int E = p.E, r = p.r;
    auto G = [&](int i) { return (float*)grads[i]; };
    Gemm32Prob pr[3];
    const float* cs_x[3]; float* cs_o[3]; int64_t cs_m[3] = {M, M, M}; int32_t cs_n[3], cs_ld[3];

    // heads: E_z = Y_z Wh^T + bh
    for (int z = 0; z < 3; ++z) pr[z] = prob(d_item3 + z * E, 3 * E, c.W(p.p_head[z]), p.H[z], nullptr, b.dY[z], p.H[z], M, p.H[z], E);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TB, s));                                   // dY = dE · Wh
    for (int z = 0; z < 3; ++z) pr[z] = prob(d_item3 + z * E, 3 * E, b.Y[z], p.H[z], nullptr, G(p.p_head[z]), p.H[z], E, p.H[z], M);
    IISAN_TRY(launch_gemm32(pr, 3, G32_TA | G32_TB | G32_ACCUM, s));              // dWh += dE^T · Y
    for (int z = 0; z < 3; ++z) { cs_x[z] = d_item3 + z * E; cs_o[z] = G(p.p_head[z] + 1); cs_n[z] = E; cs_ld[z] = 3 * E; }
    IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));
    // fc: Y_z = O_z Wf^T + bf
    IISAN_HIP_OK(hipMemsetAsync(b.zb, 0, (size_t)(ZB_HEAD + X3Z_WORDS * X3Z_SLOTS_B) * sizeof(uint32_t), s));
    b.x3z_next = b.x3z_b; b.x3z_left = X3Z_SLOTS_B;
    int fwd_x3[3];      // did the forward fc product of tower z take the split-operand route (and leave its operands' amax)?
    for (int z = 0; z < 3; ++z) {
        const Gemm32Prob f = prob(b.O[p.n[z] - 1][z], p.D[z], c.W(p.p_fc[z]), p.D[z], c.W(p.p_fc[z] + 1), b.Y[z], p.H[z], M, p.H[z], p.D[z]);
        fwd_x3[z] = (b.x3 && gemm_x3_applicable(f, 0)) ? 1 : 0;
    }
    for (int z = 0; z < 3; ++z) {
        pr[z] = prob(b.dY[z], p.H[z], c.W(p.p_fc[z]), p.D[z], nullptr, b.dO[z], p.D[z], M, p.D[z], p.H[z]);
        pr[z].amax_a = b.zb + z;                                                  // dY_z: computed here, reused by dWf below
        if (fwd_x3[z]) { pr[z].amax_b = b.amax + 3 + z; pr[z].amax_b_ready = 1; } // the fc weight: from the forward call
    }
    IISAN_TRY(gemm_group(pr, 3, G32_TB, b, s));                                   // dO = dY · Wf
    int dy_ready[3];
    for (int z = 0; z < 3; ++z) dy_ready[z] = (b.x3 && gemm_x3_applicable(pr[z], G32_TB)) ? 1 : 0;
    for (int z = 0; z < 3; ++z) {
        pr[z] = prob(b.dY[z], p.H[z], b.O[p.n[z] - 1][z], p.D[z], nullptr, G(p.p_fc[z]), p.D[z], p.H[z], p.D[z], M);
        pr[z].amax_a = b.zb + z; pr[z].amax_a_ready = dy_ready[z];          // (zeroed above; filled by the dX product if it took this route)
        if (fwd_x3[z]) { pr[z].amax_b = b.amax + z; pr[z].amax_b_ready = 1; }     // O_z: from the forward call
    }
    IISAN_TRY(gemm_group(pr, 3, G32_TA | G32_TB | G32_ACCUM, b, s));              // dWf += dY^T · O
    for (int z = 0; z < 3; ++z) { cs_x[z] = b.dY[z]; cs_o[z] = G(p.p_fc[z] + 1); cs_n[z] = p.H[z]; cs_ld[z] = p.H[z]; }
    IISAN_TRY(launch_colsum(cs_x, cs_o, cs_m, cs_n, cs_ld, 3, s));

    const int nsteps = p.diff_cv + p.diff_t + p.n[2];
    for (int g = nsteps - 1; g >= 0; --g) {
        const StepMap sm = step_map(p, g);
        const int na = sm.nact;
        if (step_fusable(p, sm, M) && !(sm.mm_i >= 0 && p.align)) {
            // dWu += dO^T · A first (needs dO as it arrives), then ONE launch turns dO into dprev in place (tile-local:
            // a workgroup reads its rows of dO into LDS before it writes them) and leaves dU, db_u, db_d, dθ; dWd += dU^T · F last
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                pr[a] = prob(b.dO[z], p.D[z], b.A[k][z], r, nullptr, G(p.wd(z, k) + 2), r, p.D[z], r, M);
            }
            IISAN_TRY(launch_gemm32(pr, na, G32_TA | G32_TB | G32_ACCUM, s));
            SanbTowerDesc td[3];
            const float* tin[6]; float* tout[6]; int32_t trows[6], tcols[6];
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                tin[2 * a] = c.W(p.wd(z, k)); tout[2 * a] = b.WT[z][0]; trows[2 * a] = p.r; tcols[2 * a] = p.D[z];              // fc_down [64,D] -> [D,64]
                tin[2 * a + 1] = c.W(p.wd(z, k) + 2); tout[2 * a + 1] = b.WT[z][1]; trows[2 * a + 1] = p.D[z]; tcols[2 * a + 1] = p.r;  // fc_up [D,64] -> [64,D]
                tower_desc(td[a], c, z, k, sm);
                td[a].Wd = b.WT[z][1]; td[a].Wu = b.WT[z][0];      // dA = dO · fc_up: fc_up^T [64, D];  dU · fc_down: fc_down^T [D, 64] (contraction index contiguous)
                td[a].dO = b.dO[z]; td[a].Upre = b.U[k][z]; td[a].dU = b.dU[z];
                td[a].dprev = k > 0 ? b.dO[z] : nullptr;     // block 0 starts from zeros / a tap: nobody reads that gradient
                td[a].dgate = cfg->gated ? G(p.gate(z, k)) : nullptr;
                td[a].dbu = G(p.wd(z, k) + 3); td[a].dbd = G(p.wd(z, k) + 1);
            }
            IISAN_TRY(launch_sanb_transpose(tin, tout, trows, tcols, 2 * na, s));
            IISAN_TRY(launch_sanb_bwd(td, na, M, cfg->gelu, s));
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                pr[a] = prob(b.dU[z], r, b.F[k][z], p.D[z], nullptr, G(p.wd(z, k)), p.D[z], r, p.D[z], M);
            }
            IISAN_TRY(launch_gemm32(pr, na, G32_TA | G32_TB | G32_ACCUM, s));         // dWd += dU^T · F
            continue;
        }
        // state = A Wu^T + bu + F ; A = act(U) ; U = F Wd^T + bd
        for (int a = 0; a < na; ++a) {
            const int z = sm.z[a], k = sm.k[a];
            pr[a] = prob(b.dO[z], p.D[z], c.W(p.wd(z, k) + 2), r, nullptr, b.dU[z], r, M, r, p.D[z], nullptr, 0, b.U[k][z]);
        }
        IISAN_TRY(launch_gemm32(pr, na, G32_TB | (cfg->gelu ? G32_MUL_GELU_GRAD : G32_MUL_RELU_MASK), s));  // dU
        {   // both weight gradients of the step, every tower, in ONE launch (round 6: they were two launches + two reducers; dO is still
            // whole here — the dF product below overwrites it in place)
            Gemm32Prob pw[6];
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                pw[a] = prob(b.dO[z], p.D[z], b.A[k][z], r, nullptr, G(p.wd(z, k) + 2), r, p.D[z], r, M);             // dWu += dO^T · A
                pw[a].colsum_a = G(p.wd(z, k) + 3);                               // dbu += column sums of dO: out of the same product
                pw[na + a] = prob(b.dU[z], r, b.F[k][z], p.D[z], nullptr, G(p.wd(z, k)), p.D[z], r, p.D[z], M);       // dWd += dU^T · F
                pw[na + a].colsum_a = G(p.wd(z, k) + 1);                          // dbd += column sums of dU
            }
            if (g_dw_merge) IISAN_TRY(launch_gemm32(pw, 2 * na, G32_TA | G32_TB | G32_ACCUM, s));
            else {
                IISAN_TRY(launch_gemm32(pw, na, G32_TA | G32_TB | G32_ACCUM, s));
                IISAN_TRY(launch_gemm32(pw + na, na, G32_TA | G32_TB | G32_ACCUM, s));
            }
        }
        for (int a = 0; a < na; ++a) {
            const int z = sm.z[a], k = sm.k[a];
            pr[a] = prob(b.dU[z], r, c.W(p.wd(z, k)), p.D[z], nullptr, b.dO[z], p.D[z], M, p.D[z], r, b.dO[z], p.D[z]);
        }
        const bool need_dp = sm.mm_i >= 0 && p.align;
        // gated fusion: its backward rides in the epilogue of the dF product (gemm32_k64_kernel<true,
        // true>: gate gradient and the (1 - g) scaling while dF is in registers) — no separate pass over dF
        bool gate_folded = false;
        if (cfg->gated) {
            K64Gate kg[3];
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                FuseTower ft{};
                c.fuse_operands(ft, z, k, sm);
                kg[a] = K64Gate{};
                kg[a].gate = ft.gate; kg[a].ga = ft.a; kg[a].ldga = ft.lda;
                if (ft.type == 1) { kg[a].go = ft.b; kg[a].ldgo = ft.ldb; }
                else { kg[a].go = ft.prev; kg[a].ldgo = ft.ldp; }
                kg[a].dgate = G(p.gate(z, k));
                kg[a].scale_prev = ft.type == 0 ? 1 : 0;
                kg[a].store = k > 0 ? 1 : 0;          // block 0 starts from zeros / a tap: nobody reads that gradient
                if (z == 2 && need_dp) { kg[a].d2 = b.dDP[sm.mm_i]; kg[a].ldd2 = p.D[2]; kg[a].d2_is_b = p.text_wide ? 1 : 0; }      // gradient wrt the dim-aligned tap
            }
            if (gemm32_k64_gate_ok(pr, kg, na)) {
                IISAN_TRY(launch_gemm32_k64_gate(pr, kg, na, s));                 // dprev = (1 - g | 1) · (dO + dU · Wd), dθ (in place)
                gate_folded = true;
            }
        }
        if (!gate_folded) IISAN_TRY(launch_gemm32(pr, na, G32_TB, s));            // dF = dO + dU · Wd (in place)

        if (!gate_folded && (cfg->gated || need_dp)) {
            FuseArgs fa{};
            fa.M = M;
            int maxD = 0;
            for (int a = 0; a < na; ++a) {
                const int z = sm.z[a], k = sm.k[a];
                c.fuse_operands(fa.t[a], z, k, sm);
                fa.t[a].F = b.dO[z];
                fa.t[a].dgate = cfg->gated ? G(p.gate(z, k)) : nullptr;
                fa.t[a].da = nullptr; fa.t[a].db = nullptr;
                if (z == 2 && need_dp) { if (p.text_wide) fa.t[a].db = b.dDP[sm.mm_i]; else fa.t[a].da = b.dDP[sm.mm_i]; }
                if (p.D[z] > maxD) maxD = p.D[z];
            }
            // every workgroup ends with ONE atomic on its tower's gate gradient and same-address atomics serialise (~12 ns
            // each): 4096 workgroups spent ~50 us there (Versa: 130 us for 230 MB).  768 workgroups, more rows each.
            unsigned gb = ew_grid(M, maxD);
            if (gb > 768) gb = 768;
            hipLaunchKernelGGL(fuse_bwd_kernel, dim3(gb, na), dim3(256), 0, s, fa);
            IISAN_LAUNCH_OK();
        }
        // not gated: dprev_z = dF_z, already in dO
    }
    if (p.align && p.n[2] > 0) {
        // DP_i = tap_wide · Pd_i^T + bd_i: the weight gradients of every aligned step as ONE group behind the chain (round 6: they sat
        // inside their steps — an amax launch and a column-sum launch each; nothing in the chain reads them)
        Gemm32Prob pd[IISAN_MAX_SIDE];
        const float* X[IISAN_MAX_SIDE]; float* O[IISAN_MAX_SIDE]; int64_t Ms[IISAN_MAX_SIDE]; int32_t Ns[IISAN_MAX_SIDE], lds[IISAN_MAX_SIDE];
        const int zw = p.text_wide ? 1 : 0;
        for (int i = 0; i < p.n[2]; ++i) {
            const StepMap sm = step_map(p, p.diff_cv + p.diff_t + i);
            pd[i] = prob(b.dDP[i], p.D[2], c.tap(zw, sm.k[zw]), (int)c.tap_ld(zw), nullptr, G(p.dpw(i)), p.D[zw], p.D[2], p.D[zw], M);
            const Gemm32Prob f = prob(c.tap(zw, sm.k[zw]), (int)c.tap_ld(zw), c.W(p.dpw(i)), p.D[zw], c.W(p.dpw(i) + 1), b.DP[i], p.D[2], M, p.D[2], p.D[zw]);
            if (b.x3 && gemm_x3_applicable(f, 0)) {       // the tap: from the forward dim-align product
                pd[i].amax_b = b.amax + 9 + i; pd[i].amax_b_ready = 1;
                pd[i].exact16_b = cfg->taps_exact16 ? 1 : 0;     // (side_net_fwd preset the slot: scale 1)
            }
            X[i] = b.dDP[i]; O[i] = G(p.dpw(i) + 1); Ms[i] = M; Ns[i] = p.D[2]; lds[i] = p.D[2];
        }
        IISAN_TRY(gemm_group(pd, p.n[2], G32_TA | G32_TB | G32_ACCUM | G32_HINT_B_EXACT16, b, s));     // dPd_i += dDP_i^T · tap
        for (int i = 0; i < p.n[2]; i += 4) IISAN_TRY(launch_colsum(X + i, O + i, Ms + i, Ns + i, lds + i, p.n[2] - i < 4 ? p.n[2] - i : 4, s));
    }
    return IISAN_OK;
}
