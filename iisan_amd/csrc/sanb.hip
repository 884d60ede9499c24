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
// streaming the [M, 768] state of every tower through HBM again; here a 16-row tile of the state stays in LDS between
// the fusion and the two products.
//
// Arithmetic: exact fp32 on the f32-input matrix cores (v_mfma_f32_16x16x4_f32 = an fp32 FMA chain).  2·2·16·D·64 FLOP per
// tile and step: 12.3k matrix-pipe cycles per wave and tile, ~64 of the 150 us of a Cached launch; the rest is the streaming
// of the tile (what was tried to overlap or shrink either part).  One 256-thread workgroup per 16-row tile and
// tower, three per CU; the fused row tile [16, D] fp32 lives in LDS (D <= 1024), weights come straight from L2 as 16-byte
// fragments: a lane's four consecutive k-values feed four MFMAs, with the same k-assignment on the other operand's LDS read.
#include "common.h"

namespace {


constexpr int R = 16;              // rows per tile
constexpr int RD = 64;             // adapter bottleneck (cfg->down)
constexpr int UST = RD + 4;        // LDS row stride of the [16, 64] buffer: 16-byte aligned rows, 4 banks apart
constexpr int NT = 256;            // threads per workgroup (4 waves), three workgroups per CU (LDS: 3 x 53.5 KB)

__device__ __forceinline__ float gate_of(const float* theta) { return 1.0f / (1.0f + __expf(-theta[0] / 0.1f)); }
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x4_f32: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C/D register r = row 4*(l>>4)+r, column l&15

struct SanbTower {
    // fusion operands (sidenet.hip FuseTower): type 0: F = g·a + (1-g)·prev ; type 1: F = prev + g·a + (1-g)·b ; no gate: sums
    const float* a; const float* b; const float* prev;
    int64_t lda, ldb, ldp;
    const float* gate;
    int32_t D, type;
    const float* Wd; const float* bd;        // narrow product's weight [64, D] (K contiguous): fwd fc_down as stored, bwd fc_up^T ; bias [64]
    const float* Wu; const float* bu;        // wide product's weight   [D, 64] (K contiguous): fwd fc_up as stored,   bwd fc_down^T ; bias [D]
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

// Workgroups are persistent (grid.x = one per CU and tower, each walks its tiles).  Ablation at the Cached batch size
// (tools/sanb_ablate.py): 93 us of a launch are its HBM phases, 64 us its two products — the sum is the measured 157 us, i.e.
// the phases of the three co-resident workgroups do NOT overlap: they start together and stay in lock-step.  A one-off start
// offset per tower (this function) was tried and is off by default: 2 / 4 / 8 units of ~4 us made the Cached step 6.55 /
// 6.75 / 6.80 ms against 6.51 — at 150 us per launch the delay costs more than the overlap returns.
// LDS: tile [16][D+4] | [16][68]
__host__ __device__ constexpr int lds_floats(int D) { return R * (D + 4) + R * UST; }

// Both products read 16 bytes per lane and instruction on BOTH operands: lane (i = lane&15, kq = lane>>4) loads the four
// consecutive k-values 16 s + 4 kq .. +3 of its row (LDS) / of its output feature's weight row (global, K contiguous) and
// feeds them to four MFMAs — the hardware contracts lane group kq of one operand with lane group kq of the other, so any
// k-assignment works as long as both sides use the same one.  (The first version read one float per lane and MFMA on each
// side: 2 x 192 four-byte loads + their address arithmetic per product and wave; the product loops are sensitive to every
// instruction between the MFMAs — tools/sanb_ablate.py.)
//
// narrow product: returns this wave's 16 x 16 fragment (features 16*wave ..) of  X[16, D] · W^T,  X in LDS (row stride FS),
// W = [64, D] row-major in global memory (the weight as stored for the forward pass, its transpose for the backward one).
// Full K per wave: no partial sums to combine, bit-reproducible.  Two accumulators break the dependent-MFMA latency.
template <int D>
__device__ __forceinline__ f4 narrow_product(const float* Xs, const float* __restrict__ W, int wave, int lane) {
    constexpr int FS = D + 4, NS = D / 16, PD = 8;
    const f4* xp = (const f4*)(Xs + (lane & 15) * FS + 4 * (lane >> 4));                       // step s: xp[4 s]
    const f4* wp = (const f4*)(W + (int64_t)(wave * 16 + (lane & 15)) * D + 4 * (lane >> 4));  // step s: wp[4 s]
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f4 b[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) b[p] = wp[4 * p];
    static_assert(NS % PD == 0, "K steps");
    // The ring only exists if the scheduler is fenced: unfenced, hipcc sinks every weight load to one or two steps before its
    // use (register pressure heuristics) and each step waits a whole L2 round trip — s_waitcnt vmcnt(0..1) before every group
    // of four MFMAs in the ISA, the matrix pipe 30 % busy and the waves 52 % of their time parked (rocprofv3 PMC).
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < NS; s0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const f4 a = xp[4 * (s0 + p)];
            const f4 bb = b[p];
            if (s0 + p + PD < NS) b[p] = wp[4 * (s0 + p + PD)];
            __builtin_amdgcn_sched_barrier(0);
            acc0 = mfma16(a[0], bb[0], acc0);
            acc1 = mfma16(a[1], bb[1], acc1);
            acc0 = mfma16(a[2], bb[2], acc0);
            acc1 = mfma16(a[3], bb[3], acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc0 + acc1;
}

// wide product in place:  T[16, D] += X[16, 64] · W^T  (+ bias), T in LDS (row stride D+4), X in LDS (stride UST), W = [D, 64]
// row-major in global memory.  A wave owns D/4 features; fragments go in pairs (independent accumulators).
template <int D>
__device__ __forceinline__ void wide_product(float* Ts, const float* Xs, const float* __restrict__ W, const float* __restrict__ bias,
                                             int wave, int lane) {
    constexpr int FS = D + 4;
    const int col = lane & 15, rg = lane >> 4;
    f4 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *(const f4*)(Xs + col * UST + 16 * s + 4 * rg);
    const f4* wp = (const f4*)(W + (int64_t)(wave * (D / 4) + col) * RD + 4 * rg);      // feature n, step s: wp[n * 16 + 4 s]
    f4 b0[4], b1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { b0[s] = wp[4 * s]; b1[s] = wp[16 * 16 + 4 * s]; }
#pragma unroll 1
    for (int n0 = wave * (D / 4); n0 < (wave + 1) * (D / 4); n0 += 32) {
        // the next pair's weights are requested before this pair's MFMAs
        f4 nb0[4], nb1[4];
        const bool more = n0 + 32 < (wave + 1) * (D / 4);
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { nb0[s] = wp[32 * 16 + 4 * s]; nb1[s] = wp[48 * 16 + 4 * s]; }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the requests above the MFMAs (see narrow_product)
        f4 c0, c1;
        float* t0 = Ts + (4 * rg) * FS + n0 + col;
        const float bi0 = bias ? bias[n0 + col] : 0.f, bi1 = bias ? bias[n0 + 16 + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { c0[r] = t0[r * FS] + bi0; c1[r] = t0[r * FS + 16] + bi1; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) { c0 = mfma16(a[s][e], b0[s][e], c0); c1 = mfma16(a[s][e], b1[s][e], c1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { t0[r * FS] = c0[r]; t0[r * FS + 16] = c1[r]; }
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { b0[s] = nb0[s]; b1[s] = nb1[s]; }
        }
        wp += 32 * 16;
    }
}

template <int D>
__global__ __launch_bounds__(NT, 3) void sanb_fwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SanbTower& t = args.t[blockIdx.y];
    constexpr int FS = D + 4, d4 = D / 4;
    float* Fs = smem;                 // [16][D + 4]   F, then O in place
    float* As = smem + R * FS;        // [16][66]      act(U)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;
    const int64_t ntiles = (args.M + R - 1) / R;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * R;

    // ---- 1. fused input tile -> LDS and HBM.  Loads of CH iterations are issued together: one HBM round trip per CH ----
    constexpr int IT = R * d4 / NT, CH = IT % 4 == 0 ? 4 : (IT % 3 == 0 ? 3 : 1);
    for (int i0 = 0; i0 < IT; i0 += CH) {
        f4 av[CH], pv[CH], bv[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = tid + (i0 + j) * NT, row = idx / d4, c = (idx - row * d4) * 4;
            const int64_t m = m0 + row;
            av[j] = pv[j] = bv[j] = (f4){0.f, 0.f, 0.f, 0.f};
            if (m < args.M) {
                av[j] = *(const f4*)(t.a + m * t.lda + c);
                if (t.prev) pv[j] = *(const f4*)(t.prev + m * t.ldp + c);
                if (t.type == 1) bv[j] = *(const f4*)(t.b + m * t.ldb + c);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = tid + (i0 + j) * NT, row = idx / d4, c = (idx - row * d4) * 4;
            const int64_t m = m0 + row;
            f4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t.type == 0) o[e] = gated ? g * av[j][e] + (1.f - g) * pv[j][e] : av[j][e] + pv[j][e];
                else o[e] = gated ? pv[j][e] + g * av[j][e] + (1.f - g) * bv[j][e] : pv[j][e] + av[j][e] + bv[j][e];
            }
            if (m < args.M) *(f4*)(t.F + m * D + c) = o;
            f2* fp = (f2*)(Fs + row * FS + c);        // rows are 8-byte aligned (FS even)
            fp[0] = (f2){o[0], o[1]};
            fp[1] = (f2){o[2], o[3]};
        }
    }
    __syncthreads();

    // ---- 2. U = F · Wd^T + bd ; A = act(U) : one 16 x 16 fragment per wave -----------------------------------------
    {
        const f4 u4 = narrow_product<D>(Fs, t.Wd, wave, lane);
        const int col = wave * 16 + (lane & 15);
        const float bd = t.bd[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r;
            const int64_t m = m0 + row;
            const float u = u4[r] + bd;
            const float a = args.gelu ? gelu_erf(u) : fmaxf(u, 0.f);
            As[row * UST + col] = a;
            if (m < args.M) {
                t.U[m * RD + col] = u;
                t.A[m * RD + col] = a;
            }
        }
    }
    __syncthreads();

    // ---- 3. O = A · Wu^T + bu + F, in place in LDS -----------------------------------------------------------------
    wide_product<D>(Fs, As, t.Wu, t.bu, wave, lane);
    __syncthreads();

    // ---- 4. tile -> HBM, full rows -----------------------------------------------------------------------------------
    for (int idx = tid; idx < R * d4; idx += NT) {
        const int row = idx / d4, c = (idx - row * d4) * 4;
        const int64_t m = m0 + row;
        if (m >= args.M) continue;
        const f2* fp = (const f2*)(Fs + row * FS + c);
        const f2 lo = fp[0], hi = fp[1];
        *(f4*)(t.O + m * D + c) = (f4){lo[0], lo[1], hi[0], hi[1]};
    }
    __syncthreads();          // the tile buffer is rewritten by the next tile's phase 1
    }
}

// backward of one step.  t.Wd = fc_up^T [64, D] (dA = dO · fc_up), t.Wu = fc_down^T [D, 64] (dU · fc_down): contraction index contiguous
template <int D>
__global__ __launch_bounds__(NT, 3) void sanb_bwd_kernel(SanbArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4];
    const SanbTower& t = args.t[blockIdx.y];
    constexpr int FS = D + 4, d4 = D / 4;
    float* Gs = smem;                 // [16][D + 4]  dO, then dF in place
    float* Ds = smem + R * FS;        // [16][66]     dU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool gated = t.gate != nullptr;
    const float g = gated ? gate_of(t.gate) : 1.0f;
    const int64_t ntiles = (args.M + R - 1) / R;
    float gate_part = 0.f;
    float dbu_acc[D / NT], dbd_acc = 0.f;
#pragma unroll
    for (int i = 0; i < D / NT; ++i) dbu_acc[i] = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * R;

    // ---- 1. dO tile -> LDS (all loads of a thread in flight together) -----------------------------------------------------
    constexpr int IT = R * d4 / NT, CH = IT % 4 == 0 ? 4 : (IT % 3 == 0 ? 3 : 1);
    {
        f4 v[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int idx = tid + j * NT, row = idx / d4, c = (idx - row * d4) * 4;
            const int64_t m = m0 + row;
            v[j] = (f4){0.f, 0.f, 0.f, 0.f};
            if (m < args.M) v[j] = *(const f4*)(t.dO + m * D + c);
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int idx = tid + j * NT, row = idx / d4, c = (idx - row * d4) * 4;
            f2* gp = (f2*)(Gs + row * FS + c);
            gp[0] = (f2){v[j][0], v[j][1]};
            gp[1] = (f2){v[j][2], v[j][3]};
        }
    }
    __syncthreads();

    // db_u += colsum(dO): one thread per column, fixed row order inside the tile; the thread -> column assignment is the same
    // for every tile of this persistent workgroup, so the sums stay in registers and ONE atomic per column and workgroup
    // goes out at the end (one per column and TILE was 1.6 M atomics per Cached launch)
    if (t.dbu) {
#pragma unroll
        for (int i = 0; i < D / NT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int row = 0; row < R; ++row) s += Gs[row * FS + tid + i * NT];
            dbu_acc[i] += s;
        }
    }

    // ---- 2. dU = (dO · Wu) ⊙ act'(U) -> LDS and HBM ------------------------------------------------------------------
    {
        const f4 da = narrow_product<D>(Gs, t.Wd, wave, lane);
        const int col = wave * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r;
            const int64_t m = m0 + row;
            float du = 0.f;
            if (m < args.M) {
                const float u = t.Upre[m * RD + col];
                du = da[r] * (args.gelu ? gelu_erf_grad(u) : (u > 0.f ? 1.f : 0.f));
                t.dU[m * RD + col] = du;
            }
            Ds[row * UST + col] = du;
        }
    }
    __syncthreads();
    if (t.dbd && tid < RD) {
        float s = 0.f;
#pragma unroll
        for (int row = 0; row < R; ++row) s += Ds[row * UST + tid];
        dbd_acc += s;
    }

    // ---- 3. dF = dO + dU · Wd, in place in LDS ---------------------------------------------------------------------------
    wide_product<D>(Gs, Ds, t.Wu, nullptr, wave, lane);
    __syncthreads();

    // ---- 4. gate gradient, dprev (and the dim-align gradients) with full-row accesses ---------------------------------------
    float part = 0.f;
    const float ca = gated ? g : 1.f, cb = gated ? 1.f - g : 1.f;
    for (int i0 = 0; i0 < IT; i0 += CH) {
        f4 av[CH], ov[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = tid + (i0 + j) * NT, row = idx / d4, c = (idx - row * d4) * 4;
            const int64_t m = m0 + row;
            av[j] = ov[j] = (f4){0.f, 0.f, 0.f, 0.f};
            if (gated && m < args.M) {
                av[j] = *(const f4*)(t.a + m * t.lda + c);
                if (t.type == 1) ov[j] = *(const f4*)(t.b + m * t.ldb + c);
                else if (t.prev) ov[j] = *(const f4*)(t.prev + m * t.ldp + c);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = tid + (i0 + j) * NT, row = idx / d4, c = (idx - row * d4) * 4;
            const int64_t m = m0 + row;
            if (m >= args.M) continue;
            const f2* gp = (const f2*)(Gs + row * FS + c);
            const f2 lo = gp[0], hi = gp[1];
            const f4 df = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) part += df[e] * (av[j][e] - ov[j][e]);
            if (t.da) *(f4*)(t.da + m * D + c) = (f4){ca * df[0], ca * df[1], ca * df[2], ca * df[3]};
            if (t.db) *(f4*)(t.db + m * D + c) = (f4){cb * df[0], cb * df[1], cb * df[2], cb * df[3]};
            if (t.dprev) {
                const float sc = (t.type == 0 && gated) ? cb : 1.f;
                *(f4*)(t.dprev + m * D + c) = (f4){sc * df[0], sc * df[1], sc * df[2], sc * df[3]};
            }
        }
    }
    gate_part += part;
    __syncthreads();          // the tile buffers are rewritten by the next tile's phase 1
    }
    if (t.dbu) {
#pragma unroll
        for (int i = 0; i < D / NT; ++i) unsafeAtomicAdd(t.dbu + tid + i * NT, dbu_acc[i]);
    }
    if (t.dbd && tid < RD) unsafeAtomicAdd(t.dbd + tid, dbd_acc);
    if (gated) {              // one atomic per workgroup for all its tiles
        gate_part = wave_sum(gate_part);
        if (lane == 0) red[wave] = gate_part;
        __syncthreads();
        if (tid == 0) atomicAdd(t.dgate, (red[0] + red[1] + red[2] + red[3]) * g * (1.f - g) / 0.1f);
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
bool sanb_fused_ok(int D, int down) { return down == RD && (D == 1024 || D == 768 || D == 512 || D == 256); }

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

// (round 5: the ablation bits, the start-stagger experiment and the non-persistent grid lost their switches: measured, documented in
//  DESIGN 6c / 6d, never the product route)
static constexpr int g_sanb_persist = 1;

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
    const int cus = iisan_cu_count();
    const int64_t ntile = ceil_div(M, R);
    // persistent: one workgroup per CU and tower (three co-resident per CU), each walking tiles x, x + grid.x, ...
    const unsigned gx = (unsigned)((g_sanb_persist && ntile > cus) ? cus : ntile);
    const dim3 grid(gx, (unsigned)n), block(NT);
    const size_t lds = lds_bytes(D);
#define SANB_LAUNCH(DD)                                                                                                \
    do {                                                                                                               \
        auto k = BWD ? sanb_bwd_kernel<DD> : sanb_fwd_kernel<DD>;                                                      \
        static OncePerDevice attr;                                                                                     \
        if (attr.first())                                                                                              \
            IISAN_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(DD))); \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                                 \
    } while (0)
    switch (D) {
        case 256: SANB_LAUNCH(256); break;
        case 512: SANB_LAUNCH(512); break;
        case 768: SANB_LAUNCH(768); break;
        case 1024: SANB_LAUNCH(1024); break;
        default: iisan_set_error("sanb: width %d", D); return IISAN_EBADSHAPE;
    }
#undef SANB_LAUNCH
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

static int64_t g_cnt_sanb_fwd = 0, g_cnt_sanb_bwd = 0;       // fused one-launch SANB steps (route counters, common.h)
IISAN_DEV_COUNTER(sanb_fused_fwd, g_cnt_sanb_fwd);
IISAN_DEV_COUNTER(sanb_fused_bwd, g_cnt_sanb_bwd);
int launch_sanb_fwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s) { ++g_cnt_sanb_fwd; return launch_sanb<false>(towers, n, M, gelu, s); }
int launch_sanb_bwd(const SanbTowerDesc* towers, int n, int64_t M, int gelu, hipStream_t s) { ++g_cnt_sanb_bwd; return launch_sanb<true>(towers, n, M, gelu, s); }

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
