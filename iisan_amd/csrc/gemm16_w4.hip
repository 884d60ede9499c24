// Fourth-generation 16-bit MFMA GEMM: persistent 256x256x64 tiles, FOUR waves per workgroup — one per SIMD, 512
// registers each — so that a finished tile can wait in registers (packed to 16 bits) while the next one is multiplied.
//
// Why: with K = 768 a tile is 12 K-steps, and in the staggered 8-wave kernel (gemm16_s256.hip) its
// epilogue costs ~20 % (FC1 ~30 %): all CUs reach their epilogues together, the chip-wide store burst backs the memory
// pipeline up, and the two wave groups of a workgroup — coupled by one barrier per slot — both wait for whichever of
// them is issuing stores.  Hiding the stores needs the accumulators free again before they are issued, i.e. a packed
// copy of the tile (64 registers per wave there — which 2 x 256-register waves per SIMD do not have).
//
// Here each wave owns a 128 x 128 quarter of the tile: 256 accumulator registers (4 x 4 fragments of
// v_mfma_f32_32x32x16), 128 registers of packed results of the PREVIOUS tile whose 32 stores are issued a few per
// K-step during the current tile, and two register sets of operand fragments (the LDS reads of K-slice j+1 are in
// flight while the 16 MFMAs of slice j run — with one wave per SIMD there is no sibling to overlap with, so the overlap
// is inside the instruction stream).  One barrier per K-step: the LDS-DMA of step s+1 is issued right after the barrier
// that opens step s (every wave has finished reading step s-1's buffer by then) and has the whole step to land.
// LDS image, swizzle, W-row permutation and lane -> output mapping are those of gemm16_p256 / gemm16_s256: a lane owns 16
// consecutive output columns of a row (32-byte 16-bit stores).
//
// STATUS (round 2): correct (tests/test_gpu_primitives.py::test_gemm16_vs_torch, variant 4) but SLOWER than the staggered
// kernel, so the dispatcher only uses it when forced (iisan_set_gemm16_variant(4)).  tools/gemm_time.py, M = 277,376, fp16:
//                     QKV    O      FC1    FC2 (K = 3072)
//   s256              963    955    900    1173   TFLOP/s
//   this kernel       577    556    444     927
//   ... stores skipped 765    744    520    1009
// Two reasons, both visible in the ISA: (1) hipcc cannot hold the packed tile in registers next to the fragments — the
// accumulators fill all 256 AGPRs, everything else must fit the 256 architectural VGPRs, and the allocator spills 70-90
// virtual registers (the packed tile goes to scratch; each scratch reload in front of a store is an L2 round trip);
// (2) with one wave per SIMD nothing covers the K-step boundary (vmcnt(0) + barrier + the first fragment reads, ~500 of
// ~2,500 cycles): the long-K product, where tile ends are rare, still trails s256 by 14 %.  What it would take: a 3-4
// deep LDS ring so that the next step's first fragments are read BEFORE the barrier, and the packed tile pinned by hand
// (inline-asm register file management).  Kept as the starting point for that kernel.
#include "common.h"
#include <type_traits>

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int WBM = 256, WBN = 256, WBK = 64;
constexpr int W_OP_BYTES = WBM * WBK * 2;        // 32 KiB per operand tile
constexpr int W_STAGE_BYTES = 2 * W_OP_BYTES;    // 64 KiB per K-step

template <typename T> struct Mfma32w;
template <> struct Mfma32w<F16> {
    static __device__ __forceinline__ f16v run(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32w<BF16> {
    static __device__ __forceinline__ f16v run(b8 a, b8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W4_BARRIER()                                     \
    do {                                                 \
        W4_FENCE();                                      \
        asm volatile("s_barrier" ::: "memory");          \
        W4_FENCE();                                      \
    } while (0)
#define W4_VMCNT0()                                          \
    do {                                                     \
        W4_FENCE();                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        W4_FENCE();                                          \
    } while (0)

template <typename T, int EPI>
__global__ __launch_bounds__(256, 1) void gemm16_w4_kernel(Gemm16Args p, int tiles_m, int tiles_n) {
    typedef typename T::v8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * W_STAGE_BYTES ring + N floats of bias
    float* sBias = (float*)(smem + 2 * W_STAGE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int G = gridDim.x;
    const int pid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntiles = tiles_m * tiles_n;
    const int my_tiles = pid < ntiles ? (ntiles - pid + G - 1) / G : 0;
    const int nk = p.K / WBK;
    const int nsteps = my_tiles * nk;
    if (nsteps == 0) return;

    // ---- LDS-DMA staging: wave w moves LDS rows 64w .. 64w+63 of both operand tiles (8 pieces of 8 rows each per operand)
    // lane -> row-in-piece r8 = lane>>3, physical 16-byte slot lane&7, logical slot = physical ^ ((row>>1)&7)
    const int r8 = lane >> 3;
    const int s0 = (lane & 7) ^ (lane >> 4);
    const int slotx0 = s0 * 16, slotx1 = (s0 ^ 4) * 16;
    const int rowA = r8 * p.lda * 2;                                             // A operand: LDS row == global row
    const int rowW = (16 * ((r8 >> 2) & 1) + (r8 & 3)) * p.ldw * 2;              // W: permuted rows (nperm32)
    const char* Abase = (const char*)p.A;
    const char* Wbase = (const char*)p.W;
    // piece pc (0..15) of this wave: pc < 8 -> A rows 64w + 8pc ; pc >= 8 -> W (LDS) rows 64w + 8(pc-8).  Addresses are an
    // SGPR base (wave-uniform: tile, K-step, piece) + one of FOUR per-lane 32-bit offsets (operand x slot parity) — with the
    // builtin's 64-bit VGPR addresses the compiler kept 16 address pairs live across the loop and spilled the packed tile.
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;
    const uint32_t vA0 = (uint32_t)(rowA + slotx0), vA1 = (uint32_t)(rowA + slotx1);
    const uint32_t vW0 = (uint32_t)(rowW + slotx0), vW1 = (uint32_t)(rowW + slotx1);
    struct StepBase { uint64_t gA, gW; uint32_t dst; };
    auto step_base = [&](int s) {
        const int ti = s / nk, kt = s - ti * nk;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
        StepBase b;
        b.gA = (uint64_t)Abase + (((int64_t)tm * WBM + wave * 64) * p.lda + (int64_t)kt * WBK) * 2;
        b.gW = (uint64_t)Wbase + (((int64_t)tn * WBN + wave * 64) * p.ldw + (int64_t)kt * WBK) * 2;
        b.dst = smem_lds + (uint32_t)((s & 1) * W_STAGE_BYTES + wave * 64 * 128);
        return b;
    };
    const int64_t lda8 = (int64_t)8 * p.lda * 2, ldw4 = (int64_t)4 * p.ldw * 2, ldw32 = (int64_t)32 * p.ldw * 2;
    auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto piece = [&](const StepBase& b, int pc) __attribute__((always_inline)) {
        const int j = pc & 7;
        // LDS rows q0 + 8(j&3).. (q0 = 64w + 32(j>>2), 32-row aligned) of the W tile hold W rows q0 + 4(j&3) + {0,16} + {0..3}
        const uint64_t g = pc < 8 ? b.gA + (uint64_t)(j * lda8) : b.gW + (uint64_t)((j >> 2) * ldw32 + (j & 3) * ldw4);
        const uint32_t glo = sgpr((uint32_t)g), ghi = sgpr((uint32_t)(g >> 32));
        const uint32_t m0v = sgpr(b.dst + (pc < 8 ? 0u : (uint32_t)W_OP_BYTES) + (uint32_t)(8 * j * 128));
        const uint64_t gb = ((uint64_t)ghi << 32) | glo;
        const uint32_t vo = pc < 8 ? ((j & 1) ? vA1 : vA0) : ((j & 1) ? vW1 : vW0);
        W4_FENCE();
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(vo), "s"(gb) : "memory");
        W4_FENCE();
    };

    f16v acc[4][4];
    const int frow = lane & 31, fh = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    // fragment read addresses = per-lane base of (operand, K-slice) + a compile-time fragment offset (the ds_read immediate)
    int aL[4], wL[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int slot = ((2 * ks + fh) ^ fsw) << 4;
        aL[ks] = (wm * 128 + frow) * 128 + slot;
        wL[ks] = W_OP_BYTES + (wn * 128 + frow) * 128 + slot;
    }

    if (p.bias)
        for (int i = tid; i < p.N; i += 256) sBias[i] = p.bias[i];
    __syncthreads();

    // packed results of the previous tile: out16[mi][ni][h2] = 8 values (columns 16*fh + 8*h2 .. +7 of the fragment's row)
    V8 out16[4][4][2];
    int pend_tile = -1;          // flat tile index (ti) whose packed results are waiting, -1 = none
    int pend_group = 0;          // next group of three stores to issue

    // store #IDX (a compile-time index: a runtime-indexed register array would live in scratch) of the waiting tile
    auto store_one = [&](int ti, auto IDX) __attribute__((always_inline)) {
        constexpr int idx = decltype(IDX)::value;
        constexpr int mi = idx >> 3, ni = (idx >> 1) & 3, h2 = idx & 1;
        const int tau = pid + ti * G;
        const int tm = tau / tiles_n, tn = tau - tm * tiles_n;
        const int64_t m = (int64_t)tm * WBM + wm * 128 + mi * 32 + frow;
        if (m >= p.M || (p.debug & 1)) return;
        const int n = tn * WBN + wn * 128 + ni * 32 + 16 * fh + 8 * h2;
        typename T::elem* op;
        if constexpr (EPI == EPI_QKVH16) {
            const unsigned Dm = (unsigned)p.qkv_heads * 64u, qS = (unsigned)p.qkv_S;
            const unsigned n64 = (unsigned)(tn * WBN + wn * 128 + (ni >> 1) * 64);
            const unsigned wq_ = n64 / Dm, hd = (n64 - wq_ * Dm) >> 6, which = wq_ + (unsigned)p.qkv_which0;
            const unsigned item = (unsigned)m / qS, tok = (unsigned)m - item * qS;
            op = (typename T::elem*)p.out + ((((int64_t)item * p.qkv_heads + hd) * 3 + which) * qS + tok) * 64 + ((ni & 1) * 32 + 16 * fh + 8 * h2);
        } else {
            op = (typename T::elem*)p.out + m * p.ldo + n;
        }
        *(V8*)op = out16[mi][ni][h2];
    };
    // the waiting tile's stores #3*g .. #3*g+2 (group g = 0..10; group 10 has two), compile-time indices
    auto store_group = [&](int g) __attribute__((always_inline)) {
        if (pend_tile < 0) return;
#define W4_G(gg)                                                                                         \
        case gg:                                                                                         \
            store_one(pend_tile, std::integral_constant<int, 3 * gg>{});                                 \
            store_one(pend_tile, std::integral_constant<int, 3 * gg + 1>{});                             \
            if (3 * gg + 2 < 32) store_one(pend_tile, std::integral_constant<int, (3 * gg + 2 < 32 ? 3 * gg + 2 : 31)>{}); \
            break;
        switch (g) {
            W4_G(0) W4_G(1) W4_G(2) W4_G(3) W4_G(4) W4_G(5) W4_G(6) W4_G(7) W4_G(8) W4_G(9) W4_G(10)
            default: break;
        }
#undef W4_G
    };
    auto store_all = [&]() __attribute__((always_inline)) {            // short-K tiles (fewer than 12 K-steps) and the last tile: everything that is left
        if (pend_tile < 0) return;
        for (int g = pend_group; g < 11; ++g) store_group(g);
        pend_tile = -1;
    };

    // accumulators -> bias / GELU -> 16-bit, into out16 (the accumulators are free afterwards)
    auto pack_tile = [&](int ti) __attribute__((always_inline)) {
        const int tau = pid + ti * G;
        const int tn = tau % tiles_n;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = tn * WBN + wn * 128 + ni * 32 + 16 * fh;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[mi][ni][r];
                if (p.bias) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f4 bb = *(const f4*)(sBias + n + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += bb[e];
                    }
                }
                f2 g[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = (f2){v[2 * k], v[2 * k + 1]};
                if constexpr (EPI == EPI_GELU16) gelu_erf_fast2x8(g);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    V8 o;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        o[e] = T::from_f32(g[4 * h2 + e / 2][0]);
                        o[e + 1] = T::from_f32(g[4 * h2 + e / 2][1]);
                    }
                    out16[mi][ni][h2] = o;
                }
                // one fragment at a time (16 accumulator reads live, not 256), and the accumulator is dead afterwards: the next
                // K-step is a tile's first and starts from C = 0
                asm volatile("" : "=a"(acc[mi][ni]));
                W4_FENCE();
            }
        pend_tile = ti;
        pend_group = 0;
    };

    // ---- prologue: step 0 into ring slot 0 ----
    {
        const StepBase b0 = step_base(0);
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) piece(b0, pc);
    }

    // one K-step: wait + barrier, a group of the previous tile's stores, then 4 K-slices of 16 MFMAs with the LDS-DMA of the
    // next step interleaved.  FIRST = a tile's first step (C = 0).
    auto kstep = [&](auto FIRST, int s, int kt) __attribute__((always_inline)) {
        W4_VMCNT0();            // my pieces of step s have landed (and the stores issued during step s-1 have left)
        W4_BARRIER();           // everybody's have; everybody has finished reading ring slot (s+1)&1 (step s-1)
        const char* stg = smem + (s & 1) * W_STAGE_BYTES;
        if (pend_tile >= 0 && pend_group < 11) { store_group(pend_group); ++pend_group; if (pend_group >= 11) pend_tile = -1; }
        // unconditional: past the end of this workgroup's steps the last step is loaded again into a buffer nobody reads
        const StepBase nb = step_base(s + 1 < nsteps ? s + 1 : s);
        V8 wf[2][4], xf[2][4];
        auto read_slice = [&](int ks, int set) __attribute__((always_inline)) {
            const char* wp = stg + wL[ks];
            const char* ap = stg + aL[ks];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[set][ni] = *(const V8*)(wp + ni * 32 * 128);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xf[set][mi] = *(const V8*)(ap + mi * 32 * 128);
        };
        auto slice = [&](auto F, int ks) __attribute__((always_inline)) {
            f16v zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = Mfma32w<T>::run(wf[ks & 1][ni], xf[ks & 1][mi], decltype(F)::value ? zero : acc[mi][ni]);
                piece(nb, ks * 4 + mi);
            }
        };
        read_slice(0, 0);
        read_slice(1, 1);
        slice(FIRST, 0);
        read_slice(2, 0);
        slice(std::false_type{}, 1);
        read_slice(3, 1);
        slice(std::false_type{}, 2);
        slice(std::false_type{}, 3);
    };

    for (int ti = 0; ti < my_tiles; ++ti) {
        const int sb = ti * nk;
        kstep(std::true_type{}, sb, 0);
        for (int kt = 1; kt < nk; ++kt) kstep(std::false_type{}, sb + kt, kt);
        store_all();            // whatever is left of the previous tile (tiles of fewer than 12 K-steps)
        pack_tile(ti);
    }
    store_all();
    W4_VMCNT0();
}

template <typename T, int EPI>
int launch_epi(const Gemm16Args& a, hipStream_t s) {
    static bool attr_set = false;
    auto kern = gemm16_w4_kernel<T, EPI>;
    if (!attr_set) {
        IISAN_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_STAGE_BYTES + 8192 * 4));
        attr_set = true;
    }
    const int tiles_m = (int)ceil_div(a.M, WBM), tiles_n = a.N / WBN;
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = (int)(ntiles < cus ? ntiles : cus);
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 2 * W_STAGE_BYTES + (size_t)a.N * 4, s, a, tiles_m, tiles_n);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

template <typename T>
int launch_t(int mode, const Gemm16Args& a, hipStream_t s) {
    switch (mode) {
        case EPI_OUT16: return launch_epi<T, EPI_OUT16>(a, s);
        case EPI_GELU16: return launch_epi<T, EPI_GELU16>(a, s);
        case EPI_QKVH16: return launch_epi<T, EPI_QKVH16>(a, s);
        default: iisan_set_error("gemm16_w4: epilogue mode %d not supported", mode); return IISAN_EBADSHAPE;
    }
}

}  // namespace

bool gemm16_w4_applicable(int mode, const Gemm16Args& a) {
    return (mode == EPI_OUT16 || mode == EPI_GELU16 || mode == EPI_QKVH16) && a.N % WBN == 0 && a.N <= 8192 && a.K % WBK == 0 &&
           a.K / WBK >= 2 && (int64_t)a.lda * 2 * WBM < (1ll << 31) && (int64_t)a.ldw * 2 * WBN < (1ll << 31);
}

int launch_gemm16_w4(int dtype16, int mode, const Gemm16Args& a, hipStream_t s) {
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(mode, a, s) : launch_t<F16>(mode, a, s);
}
