// Micro-benchmark (the "gate"): can ONE wave per SIMD keep the matrix pipe fed through a 256x256 tile's K loop?
// The production kernel (gemm16_h256.hip) runs two 256-register waves per SIMD so that one wave's LDS reads hide behind its sibling's
// MFMAs; its tile boundary then has no register room for a deferred epilogue.  The alternative — one 512-register wave per
// SIMD holding a 128x128 sub-tile (256 accumulators), fragments software-pipelined inside the wave — only makes sense if its MAIN LOOP
// alone reaches the rate the production main loop has without an epilogue (1,300-1,350 TFLOP/s).  This program measures exactly that:
// K loop only (accumulators are never stored except one checksum per lane at the end), persistent over `tiles` tiles per workgroup.
//
//   * 256 threads = 4 waves, wave w owns rows (w>>1)*128.., columns (w&1)*128.. of the tile: 4 x 4 blocks of v_mfma_f32_32x32x16_f16;
//   * operands HBM/L2 -> LDS by global_load_lds_dwordx4, FOUR stages of BK = 32 (4 x 32 KiB): the pieces of half-step h+3 are issued
//     during half-step h, waited for (vmcnt) at the end of h+1, certified by the barrier there — so the first fragments of h+2 can be
//     requested before half-step h+1 ends and the MFMA stream never waits for an LDS round trip behind a barrier;
//   * LDS rows are 64 bytes (32 halfs); 16-byte slot s of row r lives at slot s ^ ((r >> 2) & 3): a 16-lane group of a ds_read_b128
//     covers all 64 banks once;
//   * per half-step and wave: 32 MFMAs, 16 ds_read_b128 (8 per 16-wide K slice, requested one slice ahead), 8 DMA pieces.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_w1.hip -o /tmp/gemm_w1 && /tmp/gemm_w1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int BM = 256, BN = 256, BK = 32, STAGE = (BM + BN) * BK * 2, NST = 4;      // 32 KiB per stage

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// gemm_w1s (round 5): the same K loop + NS dummy 16-byte-per-lane stores per wave and half-step, issued LAST among the half-step's vector-memory
// operations into a real [M, N] fp16 output (the previous tile's positions, a 32 x 32 block per instruction as the production epilogue writes them):
// do a tile's 128 stores, trickled over the next tile (1.33 per wave and half-step at K = 768), hide behind the 768 LDS-DMA pieces that share the
// wave's in-order vector-memory queue?  NS = 0 / 1 / 2 = 0 / 96 / 192 stores per tile.
template <bool CHK, int NS, int SM = 0>
__global__ __launch_bounds__(256, 1) void gemm_w1(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ out,
                                                  int K, int tiles_n, int tiles_total, _Float16* __restrict__ C16, int ldc) {
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
    uint32_t voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 64 + j * 16 + prow;                   // row of the tile
        voff[j] = (uint32_t)((r * K + (pslot ^ ((r >> 2) & 3)) * 8) * 2);
    }
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
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds), "v"(voff[j & 3]), "s"(gb) : "memory");
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
    int c_t = blockIdx.x;                                           // compute stream's tile (for the store addresses)
    int c_kh = 0;
    const uint32_t st_lane = (uint32_t)((frow * ldc + fh * 8) * 2);
    auto trickle = [&](int h, int k) {                              // k-th dummy store of half-step h: block (2 * c_kh + k) & 15 of the wave's 4 x 4
        const int tm = c_t / tiles_n, tn = c_t - tm * tiles_n;
        const int blk = (c_kh * NS + k) & 15;
        if constexpr (SM == 1) {          // a 256-KiB window per workgroup, re-written every tile: the lines never leave the L2
            const char* base = (const char*)C16 + (size_t)blockIdx.x * 262144 + wave * 65536 + blk * 4096;
            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"((uint32_t)(lane * 16 + (c_kh & 3) * 1024)), "v"(bf[0][k & 3]), "s"(base) : "memory");
        } else if constexpr (SM == 2) {   // the big output, but 1 KiB contiguous per instruction
            const char* base = (const char*)C16 + ((size_t)(tm * BM + wm * 128 + (blk >> 2) * 32) * ldc + tn * BN + wn * 128 + (blk & 3) * 32) * 2;
            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"((uint32_t)(lane * 16)), "v"(bf[0][k & 3]), "s"(base) : "memory");
        } else {
            const char* base = (const char*)C16 + ((size_t)(tm * BM + wm * 128 + (blk >> 2) * 32) * ldc + tn * BN + wn * 128 + (blk & 3) * 32) * 2;
            if constexpr (SM == 3) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(st_lane), "v"(bf[0][k & 3]), "s"(base) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(st_lane), "v"(bf[0][k & 3]), "s"(base) : "memory");
        }
    };
    // one fragment read: q = 0..3 -> af[buf][q], 4..7 -> bf[buf][q - 4]
    auto load1 = [&](int h, int ks, int buf, int q) {
        const char* st = smem + (h & (NST - 1)) * STAGE;
        if (q < 4) af[buf][q] = *(const h8*)(st + offA[ks] + q * 2048);
        else bf[buf][q - 4] = *(const h8*)(st + offB[ks] + (q - 4) * 2048);
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
    for (int h = 0; h < H; ++h) {
        // Nothing conditional between the MFMAs (a branch there costs matrix-pipe time): past the end of the workgroup's half-steps the
        // issue stream re-loads its last half-step into stages nobody reads any more, and the fragment reads of "half-step H" read a
        // stage whose contents are never used.
        // set A: the 16 MFMAs of K slice 0 of half-step h; between them the 8 fragment reads of slice 1 and 4 DMA pieces of half-step h + 3
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int i = n >> 2, j = n & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0][j], af[0][i], acc[i][j], 0, 0, 0);
            FENCE();
            if (n < 8) load1(h, 1, 1, n);
            else if (n < 12) piece(n - 8);
            FENCE();
        }
        // set B: slice 1; between them the reads of slice 0 of half-step h + 1 (certified at the end of half-step h - 1) and the other 4 pieces
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int i = n >> 2, j = n & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[1][j], af[1][i], acc[i][j], 0, 0, 0);
            FENCE();
            if (n < 8) load1(h + 1, 0, 0, n);
            else if (n < 12) piece(n - 4);
            else if (n - 12 < NS) trickle(h, n - 12);
            FENCE();
        }
        if (i_h + 1 < H) i_next(); else ++i_h;       // (past the end: the same addresses again, into the next stage)
        if (++c_kh == nh) { c_kh = 0; c_t += gridDim.x; if (c_t >= tiles_total) c_t -= gridDim.x; }
        // the pieces of half-step h + 2 (issued during h - 1) must have landed: everything but the 8 pieces issued in this half-step, this
        // half-step's stores and the previous half-step's stores (younger than those pieces: in-order counter)
        if constexpr (NS == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // one checksum per lane (and, for the check run, the whole tile of the LAST tile this workgroup computed is NOT what this is — the
    // accumulators hold the SUM over all of the workgroup's tiles; the host check uses tiles_total <= gridDim.x so that it is one tile)
    if constexpr (CHK) {
        const int t = blockIdx.x;
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // D = B_op(W rows) x A_op(x rows)^T: lane holds x row frow, W rows (columns of C) 8 * (r >> 2) + 4 fh + (r & 3) of the 32-block
                    const int m = tm * BM + wm * 128 + i * 32 + frow;
                    const int n = tn * BN + wn * 128 + j * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                    out[(size_t)m * (tiles_n * BN) + n] = acc[i][j][r];
                }
    } else {
        // (a checksum over the 256 accumulators makes hipcc keep them in arch VGPRs for the whole kernel: 512 registers, 62 spilled; the
        //  tile's worth of plain stores leaves them in AGPRs: 336, none) — every workgroup writes its accumulators once, at the very end
        float* o = out + (size_t)blockIdx.x * 65536 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[((i * 4 + j) * 16 + r) * 256] = acc[i][j][r];
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 277504, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    const int tiles_m = M / BM, tiles_n = N / BN;
    _Float16 *A, *W; float* out;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&out, (size_t)96 << 20));       // the first 256 tiles' worth of C (the timed launches overwrite it with sums over tiles)
    std::vector<_Float16> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hA) v = (_Float16)rnd();
    for (auto& v : hW) v = (_Float16)(0.1f * rnd());
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    _Float16* C16; CHECK(hipMalloc(&C16, (size_t)M * N * 2));
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)gemm_w1<true, 2>));
    printf("gemm_w1: %d registers per thread, %zu bytes of scratch per thread, %d bytes of LDS\n", fa.numRegs, (size_t)fa.localSizeBytes, NST * STAGE);
    // 1. correctness: the first two row tiles x every column tile, one tile per workgroup
    {
        const int tt = 2 * tiles_n;
        hipLaunchKernelGGL((gemm_w1<true, 0>), dim3(tt), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tt, C16, N);
        CHECK(hipDeviceSynchronize());
        std::vector<float> ho((size_t)512 * N);
        CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
            s = s * 1664525u + 1013904223u; const int m = (s >> 8) % 512;
            s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
            worst = std::max(worst, std::fabs(ref - ho[(size_t)m * N + n]));
        }
        printf("check: worst |C - ref| over 4000 samples %.3e (values ~ %.2f)\n", worst, std::sqrt((double)K) * 0.058);
    }
    // 2. the K loop at production size
    const int tiles = tiles_m * tiles_n;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 9; ++rep) {
        const int ns = rep % 3;
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) {
            if (ns == 0) hipLaunchKernelGGL((gemm_w1<true, 0>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
            else if (ns == 1) hipLaunchKernelGGL((gemm_w1<true, 1>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
            else hipLaunchKernelGGL((gemm_w1<true, 2>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("M %d N %d K %d, %d trickled stores per wave and half-step (%d per tile): %.1f us per launch, %.0f TFLOP/s\n", M, N, K, ns, ns * 4 * (K / BK), ms / 10 * 1e3, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    }
    // what does a trickled store cost by DESTINATION / shape?  (NS = 1: 96 per tile)
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CHECK(hipFuncSetAttribute((const void*)gemm_w1<true, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    const char* names[4] = {"[M,N] output, 32 rows x 2 x 16 B per instruction", "256-KiB window per workgroup, re-written every tile (L2 resident)", "[M,N] output, 1 KiB contiguous per instruction", "[M,N] output, nt"};
    for (int rep = 0; rep < 8; ++rep) {
        const int sm = rep % 4;
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) {
            if (sm == 0) hipLaunchKernelGGL((gemm_w1<true, 1, 0>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
            else if (sm == 1) hipLaunchKernelGGL((gemm_w1<true, 1, 1>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
            else if (sm == 2) hipLaunchKernelGGL((gemm_w1<true, 1, 2>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
            else hipLaunchKernelGGL((gemm_w1<true, 1, 3>), dim3(256), dim3(256), NST * STAGE, 0, A, W, out, K, tiles_n, tiles, C16, N);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("1 store per wave and half-step -> %s: %.1f us per launch, %.0f TFLOP/s\n", names[sm], ms / 10 * 1e3, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
