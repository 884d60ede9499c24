// Micro-benchmark, second step of the VERDICT r3 1b gate: the one-wave-per-SIMD K loop of gemm_w1.hip WITH an epilogue (bias, fp16, row-major
// output) that costs the matrix pipe as little as this structure allows:
//   * a tile's accumulators are copied out block by block in the FIRST half-step of the next tile — block n (bias add, fp16 conversion:
//     16 accumulator reads + 8 v_pk_add_f32 + 8 conversions) right before the MFMA that restarts block n from C = 0 — into 128 packed
//     registers (the 512-register budget of one wave per SIMD is what pays for them: 256 accumulators + 128 + 64 fragments + ~50);
//   * their 32 stores per wave leave two at a time at the END of half-steps 1..16 of that tile (after the half-step's 8 LDS-DMA pieces:
//     `vmcnt` retires in order, and with the stores last in a half-step one `vmcnt(10)` per half-step leaves exactly the stores of the
//     half-step just issued in flight), selected by a `switch` outside the MFMA stream;
//   * W rows are permuted while staging (LDS row q of a 32-row block holds W row 16 ((q >> 2) & 1) + 4 (q >> 3) + (q & 3)) so that a lane's
//     16 results of a block are 16 consecutive columns: two 16-byte stores per block.
// RESULT (round 4): hipcc does not allocate this.  A wave has 512 registers but only 256 of them are arch VGPRs; the 256 accumulators must
// sit in AGPRs and everything VALU touches (fragments 64, packed results, addresses, the copy-out's temporaries) in the other 256.  As soon
// as VALU code reads the accumulators the allocator moves them between the two classes (1,200-1,660 v_accvgpr_read / write in the
// listing) and spills: 280 registers with the builtin MFMA, 272 with inline-asm MFMAs constrained to "a", 410 with explicit
// v_accvgpr_read, 386 with only half the tile kept packed (this version) — 74 TFLOP/s.  The K loop alone (gemm_w1.hip) allocates at 332
// registers and runs at 1,170-1,225 TFLOP/s.  The epilogue of this structure needs hand-allocated registers.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_w1e.hip -o /tmp/gemm_w1e && /tmp/gemm_w1e
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int BM = 256, BN = 256, BK = 32, STAGE = (BM + BN) * BK * 2, NST = 4;      // 32 KiB per stage

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MFMAs as inline asm with the accumulators constrained to AGPRs: left to hipcc, accumulators that VALU code reads (the copy-out) are
// allocated in arch VGPRs, of which a wave has only 256 however many registers it may use in all: 280 spilled
__device__ __forceinline__ void mfma_acc(f16v& c, h8 a, h8 b) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mfma_zero(f16v& c, h8 a, h8 b) {
    f16v z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
}
// an accumulator element for VALU code: read through an explicit v_accvgpr_read with an "a" operand — the only use of the accumulators
// outside the MFMAs, so that the allocator keeps them in AGPRs
__device__ __forceinline__ float acc_read(float x) { float y; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(x)); return y; }
__device__ __forceinline__ int nperm32(int q) { return (q & ~31) + 16 * ((q >> 2) & 1) + 4 * ((q & 31) >> 3) + (q & 3); }

__global__ __launch_bounds__(256, 1) void gemm_w1e(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const float* __restrict__ bias,
                                                   _Float16* __restrict__ out, int N, int K, int tiles_n, int tiles_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nh = K / BK;                                         // half-steps per tile
    const int my_tiles = (tiles_total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int H = my_tiles * nh;
    if (H <= 0) return;
    // DMA duty: wave w brings A rows w*64 .. +63 and W rows w*64 .. +63 of every half-step: 4 + 4 pieces of 16 rows x 64 B.  A piece's
    // source = (wave-uniform base of the tile's K slice, SGPRs) + (lane offset, the same four VGPRs for A and W: row-in-tile * K + slot)
    const int prow = lane >> 2, pslot = lane & 3;                  // row within a piece, physical 16-B slot
    uint32_t voff[4], voffw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 64 + j * 16 + prow;                   // LDS row of the tile
        voff[j] = (uint32_t)((r * K + (pslot ^ ((r >> 2) & 3)) * 8) * 2);
        voffw[j] = (uint32_t)((nperm32(r) * K + (pslot ^ ((r >> 2) & 3)) * 8) * 2);      // W: LDS row r holds W row nperm32(r)
    }
    float* sBias = (float*)(smem + NST * STAGE);
    for (int i = tid; i < N; i += 256) sBias[i] = bias[i];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;
    // the issue stream runs three half-steps ahead of the compute stream: its own tile / K counters (scalar)
    int i_kh = 0, i_t = blockIdx.x, i_h = 0;
    uint64_t i_ga = 0, i_gw = 0;
    auto i_tile = [&]() {
        const int tm = i_t / tiles_n, tn = i_t - tm * tiles_n;
        i_ga = (uint64_t)(A + (size_t)tm * BM * K);
        i_gw = (uint64_t)(W + (size_t)tn * BN * K);
    };
    i_tile();
    auto piece = [&](int j) {                                      // piece j of the current issue half-step: 0..3 = A, 4..7 = W
        const uint64_t gb = (j < 4 ? i_ga : i_gw) + (uint64_t)i_kh * (BK * 2);
        const uint32_t lds = smem_lds + (i_h & (NST - 1)) * STAGE + (j < 4 ? 0 : BM * 64) + (wave * 64 + (j & 3) * 16) * 64;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds), "v"(j < 4 ? voff[j & 3] : voffw[j & 3]), "s"(gb) : "memory");
    };
    auto i_next = [&]() {                                          // advance the issue stream by one half-step
        ++i_h;
        if (++i_kh == nh) { i_kh = 0; i_t += gridDim.x; i_tile(); }
    };
    f16v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    // fragment of the 16-wide K slice ks (0, 1) of block row/col b: row R = base + 32 b + frow, logical slot 2 ks + fh; the swizzle term
    // (R >> 2) & 3 = (frow >> 2) & 3 does not depend on b: two lane offsets per operand, everything else is an immediate
    const int sw = (frow >> 2) & 3;
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int o = frow * 64 + (((2 * ks + fh) ^ sw) << 4);
        offA[ks] = (wm * 128) * 64 + o;
        offB[ks] = BM * 64 + (wn * 128) * 64 + o;
    }
    h8 af[2][4], bf[2][4];
    // one fragment read: q = 0..3 -> af[buf][q], 4..7 -> bf[buf][q - 4]
    auto load1 = [&](int h, int ks, int buf, int q) {
        const char* st = smem + (h & (NST - 1)) * STAGE;
        if (q < 4) af[buf][q] = *(const h8*)(st + offA[ks] + q * 2048);
        else bf[buf][q - 4] = *(const h8*)(st + offB[ks] + (q - 4) * 2048);
    };
    // ---- epilogue state: the previous tile's results, packed (block n = (i, j) = (n >> 2, n & 3): 16 consecutive columns of one row) ----
    h8 pk[8][2];                                                  // blocks 8..15 of the previous tile (blocks 0..7 are stored as they are copied out)
    int c_kh = 0, c_t = blockIdx.x;                                // compute stream: half-step within the tile, tile
    int p_tm = 0, p_tn = 0;                                        // the tile `pk` belongs to
    bool have = false;                                             // pk holds a tile (wave-uniform)
    const uint32_t lane_out = (uint32_t)((frow * N + 16 * fh) * 2);
    auto copy_out = [&](int n, h8 (&dst)[2]) {                     // block n of the tile in the accumulators -> dst  (bias + convert)
        const int i = n >> 2, j = n & 3;
        const int tn = c_t % tiles_n;                              // (only called while c_t is still the finished tile... see the call sites)
        const float* bp = sBias + tn * BN + wn * 128 + j * 32 + 16 * fh;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f4 b0 = *(const f4*)(bp + 8 * q), b1 = *(const f4*)(bp + 8 * q + 4);
            h8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (_Float16)(acc_read(acc[i][j][8 * q + e]) + b0[e]);
                o[4 + e] = (_Float16)(acc_read(acc[i][j][8 * q + 4 + e]) + b1[e]);
            }
            dst[q] = o;
        }
    };
    auto store_blk = [&](int n, const h8 (&src)[2]) {             // the two 16-byte stores of block n of tile (p_tm, p_tn)
        const int i = n >> 2, j = n & 3;
        _Float16* base = out + ((size_t)(p_tm * BM + wm * 128 + i * 32) * N + p_tn * BN + wn * 128 + j * 32);      // wave-uniform
        char* op = (char*)base + lane_out;
        *(h8*)op = src[0];
        *(h8*)(op + 16) = src[1];
    };
    // prologue: half-steps 0, 1, 2 in flight; 0 and 1 landed and certified
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) piece(j);
        if (i_h + 1 < H) i_next(); else ++i_h;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) load1(0, 0, 0, q);
    int prev_t = c_t;
    for (int h = 0; h < H; ++h) {
        // set A: the 16 MFMAs of K slice 0 of half-step h; between them the 8 fragment reads of slice 1 and 4 DMA pieces of half-step h + 3.
        // The FIRST half-step of a tile restarts every block from C = 0 — right after the block's old contents (the previous tile) went to pk
        if (c_kh == 0) {
            const int keep_t = c_t;
            c_t = prev_t;                                          // copy_out reads the bias of the FINISHED tile
            p_tm = prev_t / tiles_n; p_tn = prev_t - p_tm * tiles_n;
            have = h > 0;
            _Float16* keep_out = out;
            if (!have) out = out + (size_t)0;                      // (first tile: the accumulators are zeros; the stores below write bias to tile prev_t = this tile, overwritten at its end)
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int i = n >> 2, j = n & 3;
                h8 tmp[2];
                if (n < 8) copy_out(n, tmp); else copy_out(n, pk[n - 8]);
                FENCE();
                mfma_zero(acc[i][j], bf[0][j], af[0][i]);
                FENCE();
                if (n < 8) { load1(h, 1, 1, n); store_blk(n, tmp); }      // blocks 0..7 leave at once, one block per MFMA gap
                else if (n < 12) piece(n - 8);
                FENCE();
            }
            out = keep_out;
            c_t = keep_t;
        } else {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int i = n >> 2, j = n & 3;
                mfma_acc(acc[i][j], bf[0][j], af[0][i]);
                FENCE();
                if (n < 8) load1(h, 1, 1, n);
                else if (n < 12) piece(n - 8);
                FENCE();
            }
        }
        // set B: slice 1; between them the reads of slice 0 of half-step h + 1 (certified at the end of half-step h - 1) and the other 4 pieces
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int i = n >> 2, j = n & 3;
            mfma_acc(acc[i][j], bf[1][j], af[1][i]);
            FENCE();
            if (n < 8) load1(h + 1, 0, 0, n);
            else if (n < 12) piece(n - 4);
            FENCE();
        }
        if (i_h + 1 < H) i_next(); else ++i_h;       // (past the end: the same addresses again, into the next stage)
        // two stores of the previous tile per half-step, LAST among this half-step's vector-memory operations
        if (have && c_kh >= 1 && c_kh <= 8) {
            switch (c_kh) {
#define ST(n) case n + 1: store_blk(8 + n, pk[n]); break;
                ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7)
#undef ST
            }
        }
        // the pieces of half-step h + 2 (issued during h - 1) must have landed: everything but what this half-step issued (8 pieces, <= 2 stores
        // behind them)
        if (c_kh == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // (first half-step: + 16 stores)
        __syncthreads();
        if (++c_kh == nh) { c_kh = 0; prev_t = c_t; c_t += gridDim.x; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the last tile: copy out and store everything
    {
        const int keep = c_t; c_t = prev_t;
        p_tm = prev_t / tiles_n; p_tn = prev_t - p_tm * tiles_n;
#pragma unroll
        for (int n = 0; n < 16; ++n) { h8 tmp[2]; copy_out(n, tmp); store_blk(n, tmp); }
        c_t = keep;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 277504, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    const int tiles_m = M / BM, tiles_n = N / BN;
    _Float16 *A, *W, *out; float* bias;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&W, (size_t)N * K * 2)); CHECK(hipMalloc(&out, (size_t)M * N * 2)); CHECK(hipMalloc(&bias, N * 4));
    std::vector<_Float16> hA((size_t)M * K), hW((size_t)N * K); std::vector<float> hb(N);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hA) v = (_Float16)rnd();
    for (auto& v : hW) v = (_Float16)(0.1f * rnd());
    for (auto& v : hb) v = rnd();
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    const int LDS = NST * STAGE + N * 4;
    CHECK(hipFuncSetAttribute((const void*)gemm_w1e, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)gemm_w1e));
    printf("gemm_w1e: %d registers per thread, %zu bytes of scratch per thread, %d bytes of LDS\n", fa.numRegs, (size_t)fa.localSizeBytes, LDS);
    const int tiles = tiles_m * tiles_n;
    CHECK(hipMemset(out, 0xff, (size_t)M * N * 2));
    hipLaunchKernelGGL(gemm_w1e, dim3(256), dim3(256), LDS, 0, A, W, bias, out, N, K, tiles_n, tiles);
    CHECK(hipDeviceSynchronize());
    {
        std::vector<_Float16> ho((size_t)M * N);
        CHECK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0; size_t nanc = 0;
        for (size_t i = 0; i < ho.size(); i += 9973) nanc += std::isnan((float)ho[i]);
        for (int t = 0; t < 20000; ++t) {
            s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M;
            s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
            double ref = hb[n];
            for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
            worst = std::max(worst, std::fabs(ref - (double)(float)ho[(size_t)m * N + n]));
        }
        printf("check: worst |C - ref| over 20000 samples of the whole output %.3e (fp16 rounding of values ~ 2: 1e-3), unwritten (NaN) samples %zu\n", worst, nanc);
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(gemm_w1e, dim3(256), dim3(256), LDS, 0, A, W, bias, out, N, K, tiles_n, tiles);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("M %d N %d K %d: %.1f us per launch, %.0f TFLOP/s (with the epilogue, one wave per SIMD)\n", M, N, K, ms / 10 * 1e3, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
