// Third-generation 16-bit MFMA GEMM: persistent 256x256x64 tiles with two STAGGERED wave groups.
//
// Ablation of the second generation (gemm16_p256.hip): with DMA and epilogue removed its K loop still reached only
// 55 % of the MFMA peak — all 8 waves read their fragments from LDS at the same time and then all issue MFMAs at the
// same time, so the two waves sharing a SIMD never overlap (LDS-read phase: matrix pipe idle; MFMA phase: LDS idle).
//
// Here the workgroup is split into two groups of four waves, A = wave_m 0 (rows 0..127 of the tile) and B = wave_m 1
// (rows 128..255); waves w and w+4 share a SIMD.  Time is cut into SLOTS separated by one `s_barrier`; each group
// alternates a READ slot R(s) (24 ds_read_b128: every fragment of K-step s into registers) and an MFMA slot M(s)
// (32 x v_mfma_f32_32x32x16), and B runs ONE SLOT BEHIND A:
//
//        slot:   2s        2s+1       2s+2       2s+3
//        A:      R(s)      M(s)       R(s+1)     M(s+1)
//        B:      M(s-1)    R(s)       M(s)       R(s+1)
//
// so on every SIMD one wave feeds the matrix pipe while its sibling reads LDS.  ALL LDS-DMA is issued from READ
// slots (issuing a `global_load_lds` costs the issuing wave ~60-180 cycles — measured: DMA issued from MFMA slots cost
// 21 % of the loop — while a read slot has ~400 cycles of slack next to the sibling's 1024-cycle MFMA slot).  The ring
// pieces are recycled independently: an A-operand half is private to its group (free when that group's R slot ends),
// the W tile is shared (free when B's R slot ends):
//        A, in R(s) (slot 2s)   : the whole W tile of step s+1                 (needed 2 slots later)
//        B, in R(s) (slot 2s+1) : B's A-half of step s+1, then A's A-half of step s+2   (2 / 3 slots)
// with counted waits only (A: vmcnt(0) at the end of M; B: vmcnt(8) at the end of R, vmcnt(4) at the end of M).
// Tile walk, LDS swizzle, operand swap / W-row permutation and the 16-bit epilogues are those of gemm16_p256.hip.
#include "common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int SBM = 256, SBN = 256, SBK = 64;
constexpr int S_OP_BYTES = SBM * SBK * 2;        // 32 KiB per operand tile
constexpr int S_STAGE_BYTES = 2 * S_OP_BYTES;    // 64 KiB per K-step

template <typename T> struct Mfma32s;
template <> struct Mfma32s<F16> {
    static __device__ __forceinline__ f16v run(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32s<BF16> {
    static __device__ __forceinline__ f16v run(b8 a, b8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ int nperm32s(int q) { return (q & ~31) + 16 * ((q >> 2) & 1) + 4 * ((q & 31) >> 3) + (q & 3); }

#define S256_BARRIER() asm volatile("s_barrier" ::: "memory")
#define S256_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S256_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_s256_kernel(Gemm16Args p, int tiles_m, int tiles_n) {
    typedef typename T::v8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * S_STAGE_BYTES ring + N floats of bias
    float* sBias = (float*)(smem + 2 * S_STAGE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;            // 0 = A (tile rows 0..127), 1 = B (rows 128..255)
    const int wq = wave & 3;              // wave within the group = its 64-column slice of the tile

    const int G = gridDim.x;
    const int pid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntiles = tiles_m * tiles_n;
    const int my_tiles = pid < ntiles ? (ntiles - pid + G - 1) / G : 0;
    const int nk = p.K / SBK;
    const int nsteps = my_tiles * nk;
    if (nsteps == 0) return;

    // staging offsets, kept to 5 VGPRs: lane -> row-in-chunk r8 = lane>>3, physical slot lane&7; the logical slot is
    // physical ^ ((row>>1)&7) = s0 ^ 4*(j&1) for chunk j of a 32-row-aligned run, s0 = (lane&7) ^ (lane>>4).
    const int r8 = lane >> 3;
    const int s0 = (lane & 7) ^ (lane >> 4);
    const int slotx0 = s0 * 16, slotx1 = (s0 ^ 4) * 16;
    const int rowA = r8 * p.lda * 2;                                             // A-operand: LDS row == global row
    const int rowW = (16 * ((r8 >> 2) & 1) + (r8 & 3)) * p.ldw * 2;              // W: permuted rows (nperm32s)
    const char* Abase = (const char*)p.A;
    const char* Wbase = (const char*)p.W;

    // group B: half `h` (0 = A's rows 0..127, 1 = B's rows 128..255) of the A-operand tile of flat step s;
    // wave wq stages rows h*128 + wq*32 .. +31 as 4 chunks of 8 rows
    auto issue_Ahalf = [&](int s, int h) {
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n;
        const int q0 = h * 128 + wq * 32;
        const char* Ag = Abase + (((int64_t)tm * SBM + q0) * p.lda + (int64_t)kt * SBK) * 2 + rowA;
        char* sA = smem + (s & 1) * S_STAGE_BYTES + q0 * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(Ag + j * 8 * p.lda * 2 + ((j & 1) ? slotx1 : slotx0), sA + j * 1024);
    };
    // group A: the whole W tile of flat step s; wave wq stages LDS rows wq*64 .. +63 as 8 chunks
    auto issue_Wtile = [&](int s) {
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
        const int q0 = wq * 64;
        const char* Wg = Wbase + (((int64_t)tn * SBN + q0) * p.ldw + (int64_t)kt * SBK) * 2 + rowW;
        char* sW = smem + (s & 1) * S_STAGE_BYTES + S_OP_BYTES + q0 * 128;
#pragma unroll
        for (int j = 0; j < 8; ++j)      // chunk j: LDS rows q0+8j.. -> W rows q0 + 32*(j>>2) + 4*(j&3) + {0,16}+{0..3}
            glds16(Wg + (32 * (j >> 2) + 4 * (j & 3)) * p.ldw * 2 + ((j & 1) ? slotx1 : slotx0), sW + j * 1024);
    };

    f16v acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    int xoff[4], woff2[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xoff[mi] = (grp * 128 + mi * 32 + frow) * 128;   // group g reads only its A-half
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) woff2[ni] = (wq * 64 + ni * 32 + frow) * 128;

    V8 wf[2][4], xf[4][4];
    auto read_step = [&](int s) {
        const char* sA = smem + (s & 1) * S_STAGE_BYTES;
        const char* sW = sA + S_OP_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) wf[ni][ks] = *(const V8*)(sW + woff2[ni] + slot);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
    };
    auto mfma_step = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Mfma32s<T>::run(wf[ni][ks], xf[mi][ks], acc[mi][ni]);
    };
    auto epilogue = [&](int s) {
        const int ti = s / nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int64_t m = (int64_t)tm * SBM + grp * 128 + mi * 32 + frow;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = tn * SBN + wq * 64 + ni * 32 + 16 * fh;
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
                typename T::elem* op;
                if constexpr (EPI == EPI_QKVH16) {
                    const int Dm = p.qkv_heads * 64;
                    const int64_t item = m / p.qkv_S;
                    const int tok = (int)(m - item * p.qkv_S);
                    const int which = n / Dm, hd = (n - which * Dm) >> 6, d = n & 63;
                    op = (typename T::elem*)p.out + (((item * p.qkv_heads + hd) * 3 + which) * p.qkv_S + tok) * 64 + d;
                } else {
                    op = (typename T::elem*)p.out + m * p.ldo + n;
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    V8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = T::from_f32(EPI == EPI_GELU16 ? gelu_erf_fast(v[8 * h2 + e]) : v[8 * h2 + e]);
                    *(V8*)(op + 8 * h2) = o;
                }
            }
        }
    };

    const bool nodma = (p.debug & 2) != 0;     // ablation: reuse stale LDS, no steady-state DMA
    if (p.bias)
        for (int i = tid; i < p.N; i += 512) sBias[i] = p.bias[i];
    __syncthreads();

    // ---- prologue: step 0 and the pieces of step 1 that the steady-state rules would have issued before slot 0 ----
    if (grp == 0) {
        issue_Wtile(0);
        S256_VMCNT(0);
    } else {
        issue_Ahalf(0, 0);
        issue_Ahalf(0, 1);
        if (nsteps > 1) { issue_Ahalf(1, 0); S256_VMCNT(4); } else { S256_VMCNT(0); }
    }
    S256_BARRIER();                                   // P: W(0) and both A-halves of step 0 are in LDS

    if (grp == 0) {
        // ================= group A =================
        for (int s = 0; s < nsteps; ++s) {
            // ---- slot 2s : R(s) ----
            if (s + 1 < nsteps && !nodma) issue_Wtile(s + 1);
            read_step(s);
            S256_LGKM0();
            S256_BARRIER();
            // ---- slot 2s+1 : M(s) ----
            mfma_step();
            S256_VMCNT(0);                            // W(s+1) landed (B's loads guarantee the A-halves)
            const int kt = s % nk;
            if (kt == nk - 1) epilogue(s);
            S256_BARRIER();
        }
        S256_BARRIER();                               // matches B's last slot
    } else {
        // ================= group B (one slot behind) =================
        S256_BARRIER();                               // slot 0
        for (int s = 0; s < nsteps; ++s) {
            // ---- slot 2s+1 : R(s) ----
            if (!nodma) {
                if (s + 1 < nsteps) issue_Ahalf(s + 1, 1);
                if (s + 2 < nsteps) issue_Ahalf(s + 2, 0);
            }
            read_step(s);
            S256_LGKM0();
            // A's A-half of step s+1 (issued one R slot ago) must be in LDS before A's R(s+1) in the next slot; the
            // loads issued in THIS slot may stay in flight
            if (nodma) S256_VMCNT(0);
            else if (s + 2 < nsteps) S256_VMCNT(8);
            else if (s + 1 < nsteps) S256_VMCNT(4);
            else S256_VMCNT(0);
            S256_BARRIER();
            // ---- slot 2s+2 : M(s) ----
            mfma_step();
            // B's own A-half of step s+1 (first of the two batches issued in R(s)) before R(s+1)
            if (s + 2 < nsteps && !nodma) S256_VMCNT(4); else S256_VMCNT(0);
            const int kt = s % nk;
            if (kt == nk - 1) epilogue(s);
            S256_BARRIER();
        }
    }
}

template <typename T, int EPI>
int launch_epi(const Gemm16Args& a, hipStream_t s) {
    static bool attr_set = false;
    auto kern = gemm16_s256_kernel<T, EPI>;
    if (!attr_set) {
        IISAN_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S_STAGE_BYTES + 8192 * 4));
        attr_set = true;
    }
    const int tiles_m = (int)ceil_div(a.M, SBM), tiles_n = a.N / SBN;
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = (int)(ntiles < cus ? ntiles : cus);
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * S_STAGE_BYTES + (size_t)a.N * 4, s, a, tiles_m, tiles_n);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

template <typename T>
int launch_t(int mode, const Gemm16Args& a, hipStream_t s) {
    switch (mode) {
        case EPI_OUT16: return launch_epi<T, EPI_OUT16>(a, s);
        case EPI_GELU16: return launch_epi<T, EPI_GELU16>(a, s);
        case EPI_QKVH16: return launch_epi<T, EPI_QKVH16>(a, s);
        default: iisan_set_error("gemm16_s256: epilogue mode %d not supported", mode); return IISAN_EBADSHAPE;
    }
}

}  // namespace

bool gemm16_s256_applicable(int mode, const Gemm16Args& a) {
    return (mode == EPI_OUT16 || mode == EPI_GELU16 || mode == EPI_QKVH16) && a.N % SBN == 0 && a.N <= 8192 && a.K % SBK == 0 &&
           a.K / SBK >= 2 && (int64_t)a.lda * 2 * SBM < (1ll << 31) && (int64_t)a.ldw * 2 * SBN < (1ll << 31);
}

int launch_gemm16_s256(int dtype16, int mode, const Gemm16Args& a, hipStream_t s) {
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(mode, a, s) : launch_t<F16>(mode, a, s);
}
