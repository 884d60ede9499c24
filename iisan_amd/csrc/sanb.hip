// One SANB step of the side network as ONE launch per direction (up to three towers side by side):
//
//   forward   F = fuse(tap, prev)           U = F·Wd^T + bd       A = act(U)        O = A·Wu^T + bu + F
//   backward  dA = dO·Wu   dU = dA ⊙ act'(U)   dF = dO + dU·Wd     dθ += <dF, a - prev>·g(1-g)/0.1     dprev = (1-g)·dF
//             db_u += colsum(dO)   db_d += colsum(dU)             (dWu, dWd stay separate K = M reductions, gemm32.hip)
//
// Replaces, per step, fuse_fwd + two skinny GEMM launches (forward) and two GEMMs + fuse_bwd + two column sums (backward)
// of sidenet.hip — the reference's `AdapterBlock.forward` and the gated fusion around it
// (Code_Cached/model/modules.py:112-116, Code_Cached/model/model.py:320-341) and what autograd derives from them.
// Measured before (Cached, bs = 1024, rocprofv3): 193 us forward / 444 us backward per step in 3 + 7 launches, each
// streaming the [M, 768] state of every tower through HBM again; here a 32-row tile of the state stays in LDS between
// the fusion and the two products.
//
// Arithmetic: exact fp32 on the f32-input matrix cores (v_mfma_f32_32x32x2_f32 = an fp32 FMA chain).  Bound: the f32
// matrix rate (157 TF): 2·2·32·D·64 FLOP per tile and step.  One 512-thread workgroup per 32-row tile and tower; the
// fused row tile [32, D] fp32 lives in LDS (D <= 1024), weights come straight from L2 as 16-byte fragments: a lane's
// four consecutive k-values feed four MFMAs, with the same k-permutation applied to the other operand's LDS read.
#include "common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int R = 32;              // rows per tile
constexpr int RD = 64;             // adapter bottleneck (cfg->down)
constexpr int UST = RD + 1;        // LDS row stride of the [32, 64] buffers (odd: conflict-free column reads)

__device__ __forceinline__ float gate_of(const float* theta) { return 1.0f / (1.0f + __expf(-theta[0] / 0.1f)); }
__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// C/D layout of the 32x32 MFMA: register r of lane l is row crow(r, l), column l & 31
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// acc[32, 32] += X[32, k0 .. k0+nk) · B[k0 .. k0+nk, n0 .. n0+32):  X in LDS (row stride xs floats, odd), B K-MAJOR in
// global memory (row k at B + k*ldb, columns contiguous): a wave's load of one MFMA operand is two full 128-byte lines.
// (A first version took 16-byte fragments along k from an N-major matrix — 64 cache lines per load instruction: the
// kernel ran 5x off the matrix rate.)  PD loads in flight.
template <int PD>
__device__ __forceinline__ void mma_kmajor(f16v& acc, const float* __restrict__ Xs, int xs, const float* __restrict__ B,
                                           int64_t ldb, int n0, int k0, int nk, int lane) {
    const int i = lane & 31, kk = lane >> 5;
    const float* xp = Xs + i * xs + k0 + kk;
    const float* bp = B + (int64_t)(k0 + kk) * ldb + n0 + i;
    const int steps = nk / 2;
    float b[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p)
        if (p < steps) b[p] = bp[(int64_t)(2 * p) * ldb];
    for (int st = 0; st < steps; st += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            if (st + p < steps) {
                const float a = xp[2 * (st + p)];
                const float bb = b[p];
                if (st + p + PD < steps) b[p] = bp[(int64_t)(2 * (st + p + PD)) * ldb];
                acc = mfma32(a, bb, acc);
            }
        }
    }
}

struct SanbTower {
    // fusion operands (sidenet.hip FuseTower): type 0: F = g·a + (1-g)·prev ; type 1: F = prev + g·a + (1-g)·b ; no gate: sums
    const float* a; const float* b; const float* prev;
    int64_t lda, ldb, ldp;
    const float* gate;
    int32_t D, type;
    const float* Wd; const float* bd;        // fwd: fc_down^T [D, 64], [64]      bwd: fc_up   [D, 64]  (both K-major for the narrow product)
    const float* Wu; const float* bu;        // fwd: fc_up^T   [64, D], [D]       bwd: fc_down [64, D]  (K-major for the wide product)
    float* F; float* U; float* A; float* O;  // fwd outputs: [M,D] [M,64] [M,64] [M,D]
    // backward
    const float* dO; const float* Upre;      // [M,D] gradient wrt O ; saved pre-activation [M,64]
    float* dU; float* dprev;                 // [M,64] ; [M,D] (gradient wrt prev = dO of the previous step)
    float* da; float* db;                    // optional [M,D]: g·dF / (1-g)·dF (Versa dim-align inputs)
    float* dgate; float* dbu; float* dbd;    // accumulated: scalar, [D], [64]
};
struct SanbArgs {
    SanbTower t[3];
    int64_t M;
    int32_t gelu;
};

// LDS: tile [32][D+1] | 4 K-split partial products [4][32][65] | [32][65]
__host__ __device__ constexpr int lds_floats(int D) { return (R * (D + 1) + 3) / 4 * 4 + 5 * R * UST; }

// narrow product  P[kq] = X[32, quarter kq of D] · B[quarter, 64]  (wave = column fragment nf x K quarter kq); the four
// partial products are summed in a FIXED order by the caller: bit-reproducible (LDS float atomics were not)
template <int NF>
__device__ __forceinline__ void narrow_product(const float* Xs, float* Ps, const float* B, int wave, int lane) {
    constexpr int D = NF * 256, FS = D + 1;
    const int nf = wave >> 2, kq = wave & 3;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    mma_kmajor<12>(acc, Xs, FS, B, RD, nf * 32, kq * (D / 4), D / 4, lane);
    float* P = Ps + kq * R * UST;
#pragma unroll
    for (int r = 0; r < 16; ++r) P[crow(r, lane) * UST + nf * 32 + (lane & 31)] = acc[r];
}

template <int NF>      // NF = D / 256 column fragments per wave in the wide product
__global__ __launch_bounds__(512) void sanb_fwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SanbTower& t = args.t[blockIdx.y];
    constexpr int D = NF * 256, FS = D + 1;
    float* Fs = smem;                                   // [32][D + 1]
    float* Ps = smem + (R * FS + 3) / 4 * 4;            // [4][32][65]
    float* As = Ps + 4 * R * UST;                       // [32][65]  act(U)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * R;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;

    // ---- 1. fused input tile -> LDS and HBM -------------------------------------------------------------------
    constexpr int d4 = D / 4;
    for (int idx = tid; idx < R * d4; idx += 512) {
        const int row = idx / d4, c = (idx - row * d4) * 4;
        const int64_t m = m0 + row;
        f4 o = {0.f, 0.f, 0.f, 0.f};
        if (m < args.M) {
            const f4 av = *(const f4*)(t.a + m * t.lda + c);
            f4 pv = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
            if (t.prev) pv = *(const f4*)(t.prev + m * t.ldp + c);
            if (t.type == 1) bv = *(const f4*)(t.b + m * t.ldb + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t.type == 0) o[e] = gated ? g * av[e] + (1.f - g) * pv[e] : av[e] + pv[e];
                else o[e] = gated ? pv[e] + g * av[e] + (1.f - g) * bv[e] : pv[e] + av[e] + bv[e];
            }
            *(f4*)(t.F + m * D + c) = o;
        }
        float* fp = Fs + row * FS + c;
        fp[0] = o[0]; fp[1] = o[1]; fp[2] = o[2]; fp[3] = o[3];
    }
    __syncthreads();

    // ---- 2. U = F · Wd^T ---------------------------------------------------------------------------------------
    narrow_product<NF>(Fs, Ps, t.Wd, wave, lane);
    __syncthreads();

    // ---- 3. bias, activation; U (pre-activation) and A to HBM, A to LDS ------------------------------------------
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        f4 u, a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* p = Ps + row * UST + c + e;
            u[e] = ((p[0] + p[R * UST]) + p[2 * R * UST]) + p[3 * R * UST] + t.bd[c + e];
            a[e] = args.gelu ? gelu_erf(u[e]) : fmaxf(u[e], 0.f);
            As[row * UST + c + e] = a[e];
        }
        if (m < args.M) {
            *(f4*)(t.U + m * RD + c) = u;
            *(f4*)(t.A + m * RD + c) = a;
        }
    }
    __syncthreads();

    // ---- 4. O = A · Wu^T + bu + F : wave owns D/8 columns = NF fragments ------------------------------------------
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int n0 = wave * (D / 8) + 32 * f, n = n0 + (lane & 31);
        f16v acc;
        const float bu = t.bu[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Fs[crow(r, lane) * FS + n] + bu;
        mma_kmajor<16>(acc, As, UST, t.Wu, D, n0, 0, RD, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + crow(r, lane);
            if (m < args.M) t.O[m * D + n] = acc[r];
        }
    }
}

// backward of one step.  t.Wd = fc_up [D, 64] (K-major for dA = dO · Wu), t.Wu = fc_down [64, D] (K-major for dU · Wd)
template <int NF>
__global__ __launch_bounds__(512) void sanb_bwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[8];
    const SanbTower& t = args.t[blockIdx.y];
    constexpr int D = NF * 256, FS = D + 1;
    float* Gs = smem;                                   // [32][D + 1]  dO tile
    float* Ps = smem + (R * FS + 3) / 4 * 4;            // [4][32][65]  K-split partial products of dA
    float* Ds = Ps + 4 * R * UST;                       // [32][65]     act'(U), then dU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * R;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;

    // ---- 1. dO tile -> LDS ; act'(U) -> LDS ---------------------------------------------------------------------
    constexpr int d4 = D / 4;
    for (int idx = tid; idx < R * d4; idx += 512) {
        const int row = idx / d4, c = (idx - row * d4) * 4;
        const int64_t m = m0 + row;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < args.M) v = *(const f4*)(t.dO + m * D + c);
        float* gp = Gs + row * FS + c;
        gp[0] = v[0]; gp[1] = v[1]; gp[2] = v[2]; gp[3] = v[3];
    }
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        f4 u = {0.f, 0.f, 0.f, 0.f};
        if (m < args.M) u = *(const f4*)(t.Upre + m * RD + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Ds[row * UST + c + e] = (m < args.M) ? (args.gelu ? gelu_erf_grad(u[e]) : (u[e] > 0.f ? 1.f : 0.f)) : 0.f;
    }
    __syncthreads();

    // db_u += colsum(dO): one thread per column, fixed row order
    if (t.dbu) {
        for (int c = tid; c < D; c += 512) {
            float s = 0.f;
            for (int row = 0; row < R; ++row) s += Gs[row * FS + c];
            unsafeAtomicAdd(t.dbu + c, s);
        }
    }

    // ---- 2. dA = dO · Wu ----------------------------------------------------------------------------------------
    narrow_product<NF>(Gs, Ps, t.Wd, wave, lane);
    __syncthreads();

    // ---- 3. dU = dA ⊙ act'(U) -> LDS and HBM ; db_d += colsum(dU) -------------------------------------------------
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        f4 du;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* p = Ps + row * UST + c + e;
            du[e] = (((p[0] + p[R * UST]) + p[2 * R * UST]) + p[3 * R * UST]) * Ds[row * UST + c + e];
            Ds[row * UST + c + e] = du[e];            // the same thread wrote act'(U) here: no hazard
        }
        if (m < args.M) *(f4*)(t.dU + m * RD + c) = du;
    }
    __syncthreads();
    if (t.dbd && tid < RD) {
        float s = 0.f;
        for (int row = 0; row < R; ++row) s += Ds[row * UST + tid];
        unsafeAtomicAdd(t.dbd + tid, s);
    }

    // ---- 4. dF = dO + dU · Wd ; gate gradient ; dprev -------------------------------------------------------------
    float part = 0.f;
    const float ca = gated ? g : 1.f, cb = gated ? 1.f - g : 1.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int n0 = wave * (D / 8) + 32 * f, n = n0 + (lane & 31);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Gs[crow(r, lane) * FS + n];
        mma_kmajor<16>(acc, Ds, UST, t.Wu, D, n0, 0, RD, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + crow(r, lane);
            if (m >= args.M) continue;
            const float df = acc[r];
            if (gated) {
                const float av = t.a[m * t.lda + n];
                float ov = 0.f;
                if (t.type == 1) ov = t.b[m * t.ldb + n];
                else if (t.prev) ov = t.prev[m * t.ldp + n];
                part += df * (av - ov);
            }
            if (t.da) t.da[m * D + n] = ca * df;
            if (t.db) t.db[m * D + n] = cb * df;
            if (t.dprev) t.dprev[m * D + n] = (t.type == 0 && gated) ? cb * df : df;
        }
    }
    if (gated) {
        part = wave_sum(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w];
            atomicAdd(t.dgate, s * g * (1.f - g) / 0.1f);
        }
    }
}

// out[c][r] = in[r][c] for up to 6 small matrices per launch (the adapter weights of one step)
struct TransArgs { const float* in[6]; float* out[6]; int32_t rows[6], cols[6]; };
__global__ __launch_bounds__(256) void sanb_transpose_kernel(TransArgs a) {
    __shared__ float T[32][33];
    const int z = blockIdx.z;
    const int rows = a.rows[z], cols = a.cols[z];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= rows || c0 >= cols) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) T[j][tx] = a.in[z][(int64_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) a.out[z][(int64_t)(c0 + j) * rows + r0 + tx] = T[tx][j];
}

size_t lds_bytes(int D) { return (size_t)lds_floats(D) * sizeof(float); }

}  // namespace

// ---- host interface (sidenet.hip) -------------------------------------------------------------------------------
bool sanb_fused_ok(int D, int down) { return down == RD && (D == 768 || D == 512 || D == 256); }

struct SanbTowerDesc {       // plain-pointer mirror of SanbTower for the executor
    const float* a; const float* b; const float* prev; int64_t lda, ldb, ldp; const float* gate; int32_t D, type;
    const float* Wd; const float* bd; const float* Wu; const float* bu;
    float* F; float* U; float* A; float* O;
    const float* dO; const float* Upre; float* dU; float* dprev; float* da; float* db; float* dgate; float* dbu; float* dbd;
};

static void fill(SanbTower& t, const SanbTowerDesc& d) {
    t.a = d.a; t.b = d.b; t.prev = d.prev; t.lda = d.lda; t.ldb = d.ldb; t.ldp = d.ldp; t.gate = d.gate; t.D = d.D; t.type = d.type;
    t.Wd = d.Wd; t.bd = d.bd; t.Wu = d.Wu; t.bu = d.bu; t.F = d.F; t.U = d.U; t.A = d.A; t.O = d.O;
    t.dO = d.dO; t.Upre = d.Upre; t.dU = d.dU; t.dprev = d.dprev; t.da = d.da; t.db = d.db; t.dgate = d.dgate; t.dbu = d.dbu; t.dbd = d.dbd;
}

template <bool BWD>
static int launch_sanb(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s) {
    IISAN_CHECK_SHAPE(n >= 1 && n <= 3 && M > 0, "sanb: 1..3 towers per launch");
    SanbArgs a{};
    a.M = M; a.gelu = gelu;
    const int D = towers[0].D;
    for (int i = 0; i < n; ++i) {
        IISAN_CHECK_SHAPE(towers[i].D == D && sanb_fused_ok(D, RD), "sanb: towers of one launch must share a supported width");
        fill(a.t[i], towers[i]);
    }
    const dim3 grid((unsigned)ceil_div(M, R), (unsigned)n), block(512);
    const size_t lds = lds_bytes(D);
#define SANB_LAUNCH(NF)                                                                                                \
    do {                                                                                                               \
        auto k = BWD ? sanb_bwd_kernel<NF> : sanb_fwd_kernel<NF>;                                                      \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            IISAN_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(NF * 256))); \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                                 \
    } while (0)
    switch (D) {
        case 256: SANB_LAUNCH(1); break;
        case 512: SANB_LAUNCH(2); break;
        case 768: SANB_LAUNCH(3); break;
        default: iisan_set_error("sanb: width %d", D); return IISAN_EBADSHAPE;
    }
#undef SANB_LAUNCH
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_sanb_fwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s) { return launch_sanb<false>(towers, n, M, gelu, s); }
int launch_sanb_bwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s) { return launch_sanb<true>(towers, n, M, gelu, s); }

int launch_sanb_transpose(const float* const* in, float* const* out, const int32_t* rows, const int32_t* cols, int n, hipStream_t s) {
    IISAN_CHECK_SHAPE(n >= 1 && n <= 6, "sanb_transpose: 1..6 matrices per launch");
    TransArgs a{};
    int mr = 0, mc = 0;
    for (int i = 0; i < n; ++i) {
        a.in[i] = in[i]; a.out[i] = out[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i];
        if (rows[i] > mr) mr = rows[i];
        if (cols[i] > mc) mc = cols[i];
    }
    hipLaunchKernelGGL(sanb_transpose_kernel, dim3((unsigned)ceil_div(mc, 32), (unsigned)ceil_div(mr, 32), (unsigned)n), dim3(256), 0, s, a);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
