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
constexpr int UST = RD + 4;        // LDS row stride of the [32, 64] buffers

__device__ __forceinline__ float gate_of(const float* theta) { return 1.0f / (1.0f + __expf(-theta[0] / 0.1f)); }
__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// C/D layout of the 32x32 MFMA: register r of lane l is row crow(r, l), column l & 31
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// acc[32, 32] += X[32, K-range] · W[32 rows n0.., K-range]^T, X in LDS (row stride xs floats), W rows in global memory
// (row stride ws floats, contiguous along k).  K-range = [k0, k0 + 8*groups).
template <int PD>
__device__ __forceinline__ void mma_k(f16v& acc, const float* __restrict__ Xs, int xs, const float* __restrict__ Wrow, int k0,
                                      int groups, int lane) {
    const int i = lane & 31, kk = lane >> 5;
    const float* xp = Xs + i * xs + k0 + 4 * kk;
    const float* wp = Wrow + k0 + 4 * kk;
    f4 b[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p)
        if (p < groups) b[p] = *(const f4*)(wp + 8 * p);
    for (int g = 0; g < groups; g += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            if (g + p < groups) {
                const f4 a = *(const f4*)(xp + 8 * (g + p));
                const f4 bb = b[p];
                if (g + p + PD < groups) b[p] = *(const f4*)(wp + 8 * (g + p + PD));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(a[e], bb[e], acc);
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
    const float* Wd; const float* bd;        // fwd: fc_down [64, D], [64]        bwd: Wu^T [64, D]
    const float* Wu; const float* bu;        // fwd: fc_up   [D, 64], [D]         bwd: Wd^T [D, 64]
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

template <int NF>      // NF = D / 256 column fragments per wave in the wide product
__global__ __launch_bounds__(512) void sanb_fwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SanbTower& t = args.t[blockIdx.y];
    const int D = NF * 256, FS = D + 4;
    float* Fs = smem;                       // [32][D + 4]
    float* Us = Fs + R * FS;                // [32][68]  sum of the K-split partial products
    float* As = Us + R * UST;               // [32][68]  act(U)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * R;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;

    // ---- 1. fused input tile -> LDS and HBM -------------------------------------------------------------------
    const int d4 = D / 4;
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
        *(f4*)(Fs + row * FS + c) = o;
    }
    for (int idx = tid; idx < R * UST; idx += 512) Us[idx] = 0.f;
    __syncthreads();

    // ---- 2. U = F · Wd^T : wave = (column fragment nf, K quarter kq); partial sums meet in LDS --------------------
    {
        const int nf = wave >> 2, kq = wave & 3;
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_k<4>(acc, Fs, FS, t.Wd + (int64_t)(nf * 32 + (lane & 31)) * D, kq * (D / 4), D / 32, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(Us + crow(r, lane) * UST + nf * 32 + (lane & 31), acc[r]);
    }
    __syncthreads();

    // ---- 3. bias, activation; U (pre-activation) and A to HBM, A to LDS ------------------------------------------
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        f4 u = *(const f4*)(Us + row * UST + c);
        const f4 bd = *(const f4*)(t.bd + c);
        f4 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[e] += bd[e];
            a[e] = args.gelu ? gelu_erf(u[e]) : fmaxf(u[e], 0.f);
        }
        *(f4*)(As + row * UST + c) = a;
        if (m < args.M) {
            *(f4*)(t.U + m * RD + c) = u;
            *(f4*)(t.A + m * RD + c) = a;
        }
    }
    __syncthreads();

    // ---- 4. O = A · Wu^T + bu + F : wave owns D/8 columns = NF fragments ------------------------------------------
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int n = wave * (D / 8) + 32 * f + (lane & 31);
        f16v acc;
        const float bu = t.bu[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Fs[crow(r, lane) * FS + n] + bu;
        mma_k<8>(acc, As, UST, t.Wu + (int64_t)n * RD, 0, RD / 8, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + crow(r, lane);
            if (m < args.M) t.O[m * D + n] = acc[r];
        }
    }
}

// backward of one step.  t.Wd = Wu^T [64, D], t.Wu = Wd^T [D, 64] (transposed copies made by sanb_transpose_kernel)
template <int NF>
__global__ __launch_bounds__(512) void sanb_bwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[8];
    const SanbTower& t = args.t[blockIdx.y];
    const int D = NF * 256, FS = D + 4;
    float* Gs = smem;                       // [32][D + 4]  dO tile
    float* Us = Gs + R * FS;                // [32][68]     dA partial sums
    float* Ds = Us + R * UST;               // [32][68]     act'(U), then dU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * R;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;

    // ---- 1. dO tile -> LDS ; act'(U) -> LDS ---------------------------------------------------------------------
    const int d4 = D / 4;
    for (int idx = tid; idx < R * d4; idx += 512) {
        const int row = idx / d4, c = (idx - row * d4) * 4;
        const int64_t m = m0 + row;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < args.M) v = *(const f4*)(t.dO + m * D + c);
        *(f4*)(Gs + row * FS + c) = v;
    }
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        f4 d = {0.f, 0.f, 0.f, 0.f};
        if (m < args.M) {
            const f4 u = *(const f4*)(t.Upre + m * RD + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = args.gelu ? gelu_erf_grad(u[e]) : (u[e] > 0.f ? 1.f : 0.f);
        }
        *(f4*)(Ds + row * UST + c) = d;
        *(f4*)(Us + row * UST + c) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // db_u += colsum(dO): one thread per column quad
    if (t.dbu && tid < d4) {
        f4 s = {0.f, 0.f, 0.f, 0.f};
        for (int row = 0; row < R; ++row) {
            const f4 v = *(const f4*)(Gs + row * FS + tid * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(t.dbu + tid * 4 + e, s[e]);
    }

    // ---- 2. dA = dO · Wu  (B rows = Wu^T [64, D]) ---------------------------------------------------------------
    {
        const int nf = wave >> 2, kq = wave & 3;
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_k<4>(acc, Gs, FS, t.Wd + (int64_t)(nf * 32 + (lane & 31)) * D, kq * (D / 4), D / 32, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(Us + crow(r, lane) * UST + nf * 32 + (lane & 31), acc[r]);
    }
    __syncthreads();

    // ---- 3. dU = dA ⊙ act'(U) -> LDS and HBM ; db_d += colsum(dU) -------------------------------------------------
    {
        const int row = tid >> 4, c = (tid & 15) * 4;
        const int64_t m = m0 + row;
        const f4 da = *(const f4*)(Us + row * UST + c);
        f4 du = *(const f4*)(Ds + row * UST + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) du[e] *= da[e];
        *(f4*)(Ds + row * UST + c) = du;          // same thread wrote act'(U) here: no hazard
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
        const int n = wave * (D / 8) + 32 * f + (lane & 31);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Gs[crow(r, lane) * FS + n];
        mma_k<8>(acc, Ds, UST, t.Wu + (int64_t)n * RD, 0, RD / 8, lane);
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

size_t lds_bytes(int D) { return (size_t)(R * (D + 4) + 2 * R * UST) * sizeof(float); }

}  // namespace

// ---- host interface (sidenet.hip) -------------------------------------------------------------------------------
bool sanb_fused_ok(int D, int down) { return down == RD && (D == 768 || D == 1024 || D == 512 || D == 256); }

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
        case 1024: SANB_LAUNCH(4); break;
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
