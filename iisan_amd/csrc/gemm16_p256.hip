// Second-generation 16-bit MFMA GEMM for the frozen encoders: persistent 256x256x64 tiles.
//
// Why (profiles/r1a): with K = 768 the v1 kernel (128x128 tiles, one vmcnt(0)+barrier per K-tile) spends a large
// share of every 12-step K loop in its prologue/epilogue, and 128x128 tiles need ~2x the L2->LDS bytes per FLOP.
//
// Structure (gfx950, one 512-thread workgroup per CU, 128 KiB LDS):
//   * 8 waves as 2(M) x 4(N), each owning 128x64 of the 256x256 tile = 4x2 fragments of v_mfma_f32_32x32x16
//     (128 accumulator registers);
//   * 2-deep LDS ring of full K-tiles (A 256x64 + W 256x64, 64 KiB each).  Loads for flat step s+2 are issued in
//     the MIDDLE of step s — right after a barrier that every wave passes only once ALL its operand fragments of
//     step s are in registers — so each DMA has ~1.5 K-steps to land, and waits are COUNTED (`s_waitcnt vmcnt(8)`
//     leaves the younger K-tile in flight); barriers are raw `s_barrier` (a __syncthreads would drain vmcnt);
//   * PERSISTENT: a workgroup walks its list of output tiles and the flat (tile, k) step sequence, so the loads of
//     the next tile's first two K-steps are in flight during the current tile's epilogue;
//   * LDS rows are 128 B; 16-byte slots XOR-swizzled with (row>>1)&7 — conflict-free for the ds_read_b128 lane
//     groups of the 32x32 fragment pattern (rows lane&31) — applied on the DMA source address and on the read;
//   * MFMA operands swapped (W rows as "A") and W rows permuted at staging so a lane owns 16 CONSECUTIVE output
//     columns of one row: 64-byte fp32 / 32-byte 16-bit epilogue stores with bias / GELU / residual fused;
//   * XCD-aware tile order: the 32 workgroups of one XCD walk 32 consecutive tiles (n fastest), sharing A row
//     panels and the weight matrix in that XCD's L2.
#include "common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int P_OP_BYTES = PBM * PBK * 2;        // 32 KiB per operand tile
constexpr int P_STAGE_BYTES = 2 * P_OP_BYTES;    // 64 KiB per K-tile (A + W)

template <typename T> struct Mfma32;
template <> struct Mfma32<F16> {
    static __device__ __forceinline__ f16v run(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32<BF16> {
    static __device__ __forceinline__ f16v run(b8 a, b8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// LDS row q of the W tile holds W row n0 + nperm32(q): accumulator register r of lane-half hh of a 32-row fragment
// is then output column 16*hh + r of that 32-column block.
__device__ __forceinline__ int nperm32(int q) { return (q & ~31) + 16 * ((q >> 2) & 1) + 4 * ((q & 31) >> 3) + (q & 3); }

#define P256_BARRIER() asm volatile("s_barrier" ::: "memory")

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_p256_kernel(Gemm16Args p, int tiles_m, int tiles_n) {
    typedef typename T::v8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * P_STAGE_BYTES ring + N floats of bias
    float* sBias = (float*)(smem + 2 * P_STAGE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;

    // persistent tile walk: workgroup b sits on XCD b%8; logical position pid = (b%8)*(G/8) + b/8 gives each XCD a
    // contiguous run of tiles per round (G is a multiple of 8).
    const int G = gridDim.x;
    const int pid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntiles = tiles_m * tiles_n;
    const int my_tiles = pid < ntiles ? (ntiles - pid + G - 1) / G : 0;
    const int nk = p.K / PBK;
    const int nsteps = my_tiles * nk;
    if (nsteps == 0) return;

    // staging offsets: chunk c = wave*4 + j (0..31) covers LDS rows 8c..8c+7; lane -> row q = 8c + (lane>>3),
    // physical slot lane&7 holds logical slot (lane&7) ^ ((q>>1)&7).
    int a_off[4], w_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 8 + (lane >> 3);
        const int slog = (lane & 7) ^ ((q >> 1) & 7);
        a_off[j] = q * p.lda * 2 + slog * 16;
        w_off[j] = nperm32(q) * p.ldw * 2 + slog * 16;
    }
    const char* Abase = (const char*)p.A;
    const char* Wbase = (const char*)p.W;

    auto issue = [&](int s) {       // DMA of flat step s into ring slot s&1
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
        const char* Ag = Abase + ((int64_t)tm * PBM * p.lda + (int64_t)kt * PBK) * 2;
        const char* Wg = Wbase + ((int64_t)tn * PBN * p.ldw + (int64_t)kt * PBK) * 2;
        char* sA = smem + (s & 1) * P_STAGE_BYTES + wave * 4096;
        char* sW = sA + P_OP_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(Ag + a_off[j], sA + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(Wg + w_off[j], sW + j * 1024);
    };

    f16v acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment reads: row = base32 + (lane&31); logical slot = 2*ks + (lane>>5); swizzle (row>>1)&7 = (lane>>1)&7 ^ const
    const int frow = lane & 31, fh = lane >> 5;
    int xoff[4], woff2[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xoff[mi] = (wave_m * 128 + mi * 32 + frow) * 128;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) woff2[ni] = (wave_n * 64 + ni * 32 + frow) * 128;
    const int fsw = (frow >> 1) & 7;     // bases are multiples of 32 rows -> (row>>1)&7 == (frow>>1)&7

    // the whole bias vector lives in LDS: the epilogue then needs no VGPR-destination global load, whose wait hipcc
    // can only express as vmcnt(0) — a full drain of the DMA pipeline once per output tile
    if (p.bias)
        for (int i = tid; i < p.N; i += 512) sBias[i] = p.bias[i];
    __syncthreads();

    issue(0);
    if (nsteps > 1) issue(1);
    // de-synchronise the workgroups: identical tiles keep every CU in lock-step, so all epilogues (pure HBM writes)
    // and all main loops (pure MFMA) would coincide chip-wide; a one-off start offset of up to ~one tile time spreads
    // the store bursts under other CUs' compute for the rest of the launch.
    if (p.debug & 4) {
        const int units = ((pid * 7919) >> 2) & 63;          // 0..48 us in 0.76 us units: spans a whole tile period
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(25);
    }

    bool waited = false;      // the DMA of this step was already waited for (before the previous tile's epilogue)
    for (int s = 0; s < nsteps; ++s) {
        if (!waited) {
            if (s + 1 < nsteps && !(p.debug & 2)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        waited = false;
        P256_BARRIER();
        const char* sA = smem + (s & 1) * P_STAGE_BYTES;
        const char* sW = sA + P_OP_BYTES;
        V8 wf[2][4], xf[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) wf[ni][ks] = *(const V8*)(sW + woff2[ni] + slot);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int mi = 2; mi < 4; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
        // first half of the MFMAs (rows 0..63 of the wave's 128)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Mfma32<T>::run(wf[ni][ks], xf[mi][ks], acc[mi][ni]);
        // every fragment of this step is in registers: the slot may be overwritten once all waves get here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P256_BARRIER();
        if (s + 2 < nsteps && !(p.debug & 2)) issue(s + 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 2; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Mfma32<T>::run(wf[ni][ks], xf[mi][ks], acc[mi][ni]);

        const int ti = s / nk, kt = s - ti * nk;
        if (kt != nk - 1) continue;

        // Retire the NEXT step's DMA before the epilogue stores enter the vmcnt queue (vmcnt counts stores too and a
        // counted wait cannot tell them from loads): the stores then have a whole K-step to drain before the wait at
        // the top of step s+2 has to cover them, instead of stalling step s+1 (ablation: stores cost QKV 20 %).
        if (s + 1 < nsteps) {
            if (s + 2 < nsteps && !(p.debug & 2)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            waited = true;
        }
        // ---- epilogue of output tile (pid + ti*G): lane (frow, fh) owns row .. + frow, 16 consecutive columns ----
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int64_t m = (int64_t)tm * PBM + wave_m * 128 + mi * 32 + frow;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = tn * PBN + wave_n * 64 + ni * 32 + 16 * fh;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[mi][ni][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
                if (m >= p.M || (p.debug & 1)) continue;
                if (p.bias) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f4 bb = *(const f4*)(sBias + n + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += bb[e];
                    }
                }
                if constexpr (EPI == EPI_OUT16 || EPI == EPI_GELU16 || EPI == EPI_QKVH16) {
                    typename T::elem* op;
                    if constexpr (EPI == EPI_QKVH16) {
                        const int Dm = p.qkv_heads * 64;
                        const int64_t item = m / p.qkv_S;
                        const int tok = (int)(m - item * p.qkv_S);
                        const int wq_ = n / Dm, hd = (n - wq_ * Dm) >> 6, d = n & 63, which = wq_ + p.qkv_which0;
                        op = (typename T::elem*)p.out + (((item * p.qkv_heads + hd) * 3 + which) * p.qkv_S + tok) * 64 + d;
                    } else {
                        op = (typename T::elem*)p.out + m * p.ldo + n;
                    }
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        V8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = T::from_f32(EPI == EPI_GELU16 ? gelu_erf_fast(v[8 * h2 + e]) : v[8 * h2 + e]);
                        if (p.debug & 8) asm volatile("" ::"v"(o));           // ablation: math without the store
                        else *(V8*)(op + 8 * h2) = o;
                    }
                } else {
                    const float* rp;
                    float* op;
                    if constexpr (EPI == EPI_RESID32) {
                        rp = p.resid + m * p.ldo + n;
                        op = (float*)p.out + m * p.ldo + n;
                    } else {   // EPI_PATCH32
                        const int64_t img = m / p.patch_P;
                        const int pp = (int)(m - img * p.patch_P);
                        rp = p.pos + (int64_t)(1 + pp) * p.N + n;
                        op = (float*)p.out + (img * (p.patch_P + 1) + 1 + pp) * p.ldo + n;
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f4 rr = *(const f4*)(rp + 4 * q4);
                        *(f4*)(op + 4 * q4) = (f4){v[4 * q4] + rr[0], v[4 * q4 + 1] + rr[1], v[4 * q4 + 2] + rr[2], v[4 * q4 + 3] + rr[3]};
                    }
                }
            }
        }
    }
}

template <typename T, int EPI>
int launch_epi(const Gemm16Args& a, hipStream_t s) {
    static OncePerDevice attr;
    auto kern = gemm16_p256_kernel<T, EPI>;
    if (attr.first())
        IISAN_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE_BYTES + 8192 * 4));
    const int tiles_m = (int)ceil_div(a.M, PBM), tiles_n = a.N / PBN;
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    const int cus = iisan_cu_count();
    int grid = (int)(ntiles < cus ? ntiles : cus);
    grid = (grid + 7) / 8 * 8;          // the XCD walk needs a multiple of 8; surplus workgroups exit immediately
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * P_STAGE_BYTES + (size_t)a.N * 4, s, a, tiles_m, tiles_n);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

template <typename T>
int launch_t(int mode, const Gemm16Args& a, hipStream_t s) {
    switch (mode) {
        case EPI_OUT16: return launch_epi<T, EPI_OUT16>(a, s);
        case EPI_GELU16: return launch_epi<T, EPI_GELU16>(a, s);
        case EPI_RESID32: return launch_epi<T, EPI_RESID32>(a, s);
        case EPI_PATCH32: return launch_epi<T, EPI_PATCH32>(a, s);
        case EPI_QKVH16: return launch_epi<T, EPI_QKVH16>(a, s);
        default: iisan_set_error("gemm16_p256: bad epilogue mode %d", mode); return IISAN_EBADSHAPE;
    }
}

}  // namespace

// usable when N is a multiple of 256, K a multiple of 64, and the operands are addressable with 32-bit byte offsets
// inside one tile (lda*2*256 < 2^31).  A must be readable for ceil(M/256)*256 rows.
bool gemm16_p256_applicable(const Gemm16Args& a) {
    return a.N % PBN == 0 && a.N <= 8192 && a.K % PBK == 0 && (int64_t)a.lda * 2 * PBM < (1ll << 31) && (int64_t)a.ldw * 2 * PBN < (1ll << 31);
}

int launch_gemm16_p256(int dtype16, int mode, const Gemm16Args& a, hipStream_t s) {
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(mode, a, s) : launch_t<F16>(mode, a, s);
}
