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
// so on every SIMD one wave feeds the matrix pipe while its sibling reads LDS.
//
// Where the LDS-DMA goes was settled with tools/micro/slot_dma.hip (cycles per K-step on one CU, 2048 = MFMA bound):
//   8 `global_load_lds` per wave at the head of the read slot 3592, after the ds_reads 2754, 4 + 4 split 2741,
//   ALL 8 INTERLEAVED WITH THE MFMAs (one after every 4) AND WAITED FOR ONE SLOT LATER: 2168.
// What costs is a wave waiting for its own loads or queueing at the texture addresser next to ds_reads, not the issue
// among MFMAs.  So read slots hold only the 24 ds_reads plus the wait, MFMA slots issue the DMA, and every piece has a
// full slot between its issue and its wait:
//        A in M(s)   (slot 2s+1) : B's A-operand half of step s+1 (read in slot 2s+3), A's own half of step s+2 (2s+4)
//        B in M(s)   (slot 2s+2) : the whole W tile of step s+2   (read in slots 2s+4, 2s+5)
//        A and B: `s_waitcnt vmcnt(0)` at the END of each read slot (so the stores of an epilogue get a slot as well)
// Each ring buffer was last read at least one slot before it is overwritten (2 stages of 64 KiB).
// The compiler must be fenced (`sched_barrier`) around every barrier / wait: MFMAs are pure register ops, and without
// the fences 28 of group B's 32 MFMAs were hoisted above the `s_barrier` into its read slot and the `s_waitcnt vmcnt`
// to the top of the MFMA slot — turning the stagger back into lock-step (seen in the ISA, cost ~10 %).
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

#define S256_FENCE() __builtin_amdgcn_sched_barrier(0)
#define S256_BARRIER()                                   \
    do {                                                 \
        S256_FENCE();                                    \
        asm volatile("s_barrier" ::: "memory");          \
        S256_FENCE();                                    \
    } while (0)
#define S256_VMCNT(n)                                        \
    do {                                                     \
        S256_FENCE();                                        \
        asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
        S256_FENCE();                                        \
    } while (0)
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

    // One 1-KiB piece (8 LDS rows) of the A-operand half `h` / the W half `h` of flat step s; wave wq owns rows
    // h*128 + wq*32 .. +31 of a half, piece j = rows +8j.
    auto piece_A = [&](int s, int h, int j) {
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n;
        const int q0 = h * 128 + wq * 32;
        const char* Ag = Abase + (((int64_t)tm * SBM + q0 + 8 * j) * p.lda + (int64_t)kt * SBK) * 2 + rowA;
        glds16(Ag + ((j & 1) ? slotx1 : slotx0), smem + (s & 1) * S_STAGE_BYTES + (q0 + 8 * j) * 128);
    };
    auto piece_W = [&](int s, int h, int j) {   // LDS rows q0+8j.. hold W rows q0 + 4j + {0,16} + {0..3} (nperm32s)
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
        const int q0 = h * 128 + wq * 32;
        const char* Wg = Wbase + (((int64_t)tn * SBN + q0 + 4 * j) * p.ldw + (int64_t)kt * SBK) * 2 + rowW;
        glds16(Wg + ((j & 1) ? slotx1 : slotx0), smem + (s & 1) * S_STAGE_BYTES + S_OP_BYTES + (q0 + 8 * j) * 128);
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
    // 32 MFMAs with this wave's 8 DMA pieces interleaved, one after every 4 MFMAs.  Group A: pieces 0..3 = B's
    // A-operand half of step s1 (if d1), 4..7 = A's own half of step s2 (if d2); group B: the W tile of step s2 (if d2).
    auto mfma_step = [&](int s1, bool d1, int s2, bool d2) {
        // wave-uniform bases first (scalar divisions), so that only adds sit between the MFMAs
        const char* g1 = nullptr; const char* g2 = nullptr;
        char* l1 = nullptr; char* l2 = nullptr;
        int64_t gstep = 0;
        if (grp == 0) {
            if (d1) {
                const int ti = s1 / nk, kt = s1 - ti * nk;
                const int tm = (pid + ti * G) / tiles_n;
                g1 = Abase + (((int64_t)tm * SBM + 128 + wq * 32) * p.lda + (int64_t)kt * SBK) * 2 + rowA;
                l1 = smem + (s1 & 1) * S_STAGE_BYTES + (128 + wq * 32) * 128;
            }
            if (d2) {
                const int ti = s2 / nk, kt = s2 - ti * nk;
                const int tm = (pid + ti * G) / tiles_n;
                g2 = Abase + (((int64_t)tm * SBM + wq * 32) * p.lda + (int64_t)kt * SBK) * 2 + rowA;
                l2 = smem + (s2 & 1) * S_STAGE_BYTES + (wq * 32) * 128;
            }
            gstep = (int64_t)8 * p.lda * 2;
        } else if (d2) {
            const int ti = s2 / nk, kt = s2 - ti * nk;
            const int tau = pid + ti * G;
            const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
            g2 = Wbase + (((int64_t)tn * SBN + wq * 32) * p.ldw + (int64_t)kt * SBK) * 2 + rowW;
            l2 = smem + (s2 & 1) * S_STAGE_BYTES + S_OP_BYTES + (wq * 32) * 128;
            gstep = (int64_t)4 * p.ldw * 2;
        }
        S256_FENCE();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Mfma32s<T>::run(wf[ni][ks], xf[mi][ks], acc[mi][ni]);
                if (mi & 1) {
                    const int pc = ks * 2 + (mi >> 1);          // 0..7
                    const int j = pc & 3;
                    S256_FENCE();
                    if (grp == 0) {
                        if (pc < 4) { if (d1) glds16(g1 + j * gstep + ((j & 1) ? slotx1 : slotx0), l1 + j * 1024); }
                        else if (d2) glds16(g2 + j * gstep + ((j & 1) ? slotx1 : slotx0), l2 + j * 1024);
                    } else if (d2) {
                        // W half pc>>2: LDS rows +128, global rows +128
                        glds16(g2 + (int64_t)(pc >> 2) * 128 * p.ldw * 2 + j * gstep + ((j & 1) ? slotx1 : slotx0), l2 + (pc >> 2) * 128 * 128 + j * 1024);
                    }
                    S256_FENCE();
                }
            }
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
    if (p.debug & 4) {                         // experiment: spread the CUs' tile phases over ~one tile time
        const int units = (int)(((unsigned)pid * 40503u) >> 5) & 63;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if (p.bias)
        for (int i = tid; i < p.N; i += 512) sBias[i] = p.bias[i];
    __syncthreads();

    // ---- prologue: what the steady-state rules would have issued before slot 0 ----
    if (grp == 0) {
        for (int j = 0; j < 4; ++j) piece_W(0, 0, j);
        for (int j = 0; j < 4; ++j) piece_A(0, 1, j);
        S256_VMCNT(0);
    } else {
        for (int j = 0; j < 4; ++j) piece_A(0, 0, j);
        for (int j = 0; j < 4; ++j) piece_W(0, 1, j);
        if (nsteps > 1) {
            for (int j = 0; j < 4; ++j) piece_A(1, 0, j);
            for (int j = 0; j < 8; ++j) piece_W(1, j >> 2, j & 3);
            S256_VMCNT(12);
        } else {
            S256_VMCNT(0);
        }
    }
    S256_BARRIER();                                   // P: W(0) and both A-halves of step 0 are in LDS

    if (grp == 0) {
        // ================= group A =================
        for (int s = 0; s < nsteps; ++s) {
            // ---- slot 2s : R(s) ----
            read_step(s);
            S256_LGKM0();
            S256_VMCNT(0);                            // B's half of step s and A's half of step s+1 (issued in M(s-1))
            S256_BARRIER();
            // ---- slot 2s+1 : M(s) ----
            mfma_step(s + 1, s + 1 < nsteps && !nodma, s + 2, s + 2 < nsteps && !nodma);
            const int kt = s % nk;
            if (kt == nk - 1) epilogue(s);
            S256_BARRIER();
        }
        S256_BARRIER();                               // matches B's last slot
    } else {
        // ================= group B (one slot behind) =================
        S256_BARRIER();                               // slot 0 (W(1) was issued in the prologue)
        for (int s = 0; s < nsteps; ++s) {
            // ---- slot 2s+1 : R(s) ----
            read_step(s);
            S256_LGKM0();
            S256_VMCNT(0);                            // the W tile of step s+1 (issued in M(s-1))
            S256_BARRIER();
            // ---- slot 2s+2 : M(s) ----
            mfma_step(0, false, s + 2, s + 2 < nsteps && !nodma);
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
