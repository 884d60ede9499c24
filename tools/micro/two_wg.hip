// Micro-benchmark 5: would TWO independent 4-wave workgroups per CU (128x256x32 tiles, lock-step inside a workgroup,
// naturally desynchronised between workgroups) hide the epilogue better than one 8-wave staggered workgroup?
// Emulates: per K-step(32) 12 ds_read_b128 + 16 MFMA 32x32x16 + 6 global_load_lds per wave; every `per_tile` K-steps an
// epilogue of `epi_valu` dependent-free packed FMAs + 16 global stores per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define SB() do { FENCE(); asm volatile("s_barrier" ::: "memory"); FENCE(); } while (0)
// MODE 0: 2 WGs/CU x 4 waves (this file's question).  MODE 1: 1 WG/CU x 8 waves staggered groups (gemm16_s256's shape:
// per K-step(64) 24 reads + 32 MFMA + 8 glds), same epilogue per wave.
template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void k(const char* src, char* out, long long* cyc, float* sink, int ksteps, int per_tile,
                                                             int epi_valu, int win) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nthr = MODE == 0 ? 256 : 512;
    for (int i = tid; i < (MODE == 0 ? 64 : 128) * 1024 / 4; i += nthr) ((float*)smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f16v acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    constexpr int NR = MODE == 0 ? 12 : 24;
    h8 fr[NR];
    for (int j = 0; j < NR; ++j) fr[j] = *(const h8*)(smem + lane * 16 + j * 1024);
    const char* base = smem + (wave & 3) * 12 * 1024 + lane * 16;
    char* dst = smem + (MODE == 0 ? 48 : 96) * 1024 + (wave & 3) * 4096;
    const char* g = src + ((long)blockIdx.x * win) + wave * 8192 + lane * 16;
    char* op = out + ((long)blockIdx.x * nthr + tid) * 16;
    unsigned off = 0;
    auto dma = [&](int j) { glds16(g + ((off + j * 1024) & (win - 1)), dst + (j & 3) * 1024); };
    auto epilogue = [&]() {
        f2 x = {acc[0][0], acc[0][1]}, y = {1.0001f, 0.9999f};
        for (int i = 0; i < epi_valu; i += 4) {        // independent chains: VALU throughput, not latency
            f2 a0 = x, a1 = x + 1.f, a2 = x + 2.f, a3 = x + 3.f;
            a0 = __builtin_elementwise_fma(a0, y, y); a1 = __builtin_elementwise_fma(a1, y, y);
            a2 = __builtin_elementwise_fma(a2, y, y); a3 = __builtin_elementwise_fma(a3, y, y);
            x = a0 + a1 + a2 + a3;
            asm volatile("" : "+v"(x));
        }
        f4 v = {x[0], x[1], acc[1][0], acc[2][0]};
#pragma unroll
        for (int i = 0; i < 16; ++i) *(f4*)(op + (long)i * gridDim.x * nthr * 16) = v;
    };
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        for (int s = 0; s < ksteps; ++s) {
#pragma unroll
            for (int j = 0; j < 12; ++j) fr[j] = *(const h8*)(base + j * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FENCE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the batch issued one K-step ago
            SB();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[(i & 1) * 2 + (i >> 3)], fr[4 + ((i >> 1) & 3) * 2 + (i >> 3)], acc[i & 7], 0, 0, 0);
                if (i == 2 || i == 5 || i == 8 || i == 10 || i == 13 || i == 15) { FENCE(); dma(i & 7); FENCE(); }
            }
            off += 65536;
            if ((s % per_tile) == per_tile - 1) epilogue();
            SB();
        }
    } else {
        const int grp = wave >> 2;
        auto body = [&](int s) {
#pragma unroll
            for (int j = 0; j < 24; ++j) fr[j] = *(const h8*)(base + j * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FENCE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SB();
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[(i & 1) * 4 + (i >> 3)], fr[8 + ((i >> 1) & 3) * 4 + (i >> 3)], acc[i & 7], 0, 0, 0);
                if ((i & 3) == 3) { FENCE(); dma(i >> 2); FENCE(); }
            }
            off += 65536;
            if ((s % per_tile) == per_tile - 1) { SB(); epilogue(); }
            SB();
        };
        if (grp == 0) { for (int s = 0; s < ksteps; ++s) body(s); SB(); }
        else { SB(); for (int s = 0; s < ksteps; ++s) body(s); }
    }
    long long t1 = __builtin_readcyclecounter();
    f16v sv = acc[0];
    for (int a = 1; a < 8; ++a) sv += acc[a];
    float sm = sv[0] + sv[5];
    if (sm == 12345.f) for (int j = 0; j < NR; ++j) sm += (float)fr[j][0];
    sink[blockIdx.x * nthr + tid] = sm;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    char *src, *out; long long* cyc; float* sink;
    hipMalloc(&src, 1L << 30); hipMemset(src, 1, 1L << 30);
    hipMalloc(&out, 16L * 512 * 512 * 16); hipMalloc(&cyc, 512 * 8 * 8); hipMalloc(&sink, 512 * 512 * 4);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    std::vector<long long> h(4096);
    for (int win : {65536, 1 << 20})
    for (int epi : {0, 400, 1200}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int ksteps64 = 2400;                       // K-steps of 64 per workgroup-tile stream
            float ms = 0;
            for (int it = 0; it < 2; ++it) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) k<0><<<512, 256, 72 * 1024>>>(src, out, cyc, sink, ksteps64 * 2, 24, epi, win);
                else k<1><<<256, 512, 136 * 1024>>>(src, out, cyc, sink, ksteps64, 12, epi, win);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            // flops: mode 0: 512 WGs x 4 waves x 16 MFMA x 2*ksteps64 ; mode 1: 256 x 8 x 32 x ksteps64  (identical)
            const double flops = 256.0 * 8 * 32 * 32768.0 * ksteps64;
            printf("win=%7d epilogue=%4d pk-fma+16 stores per wave and tile | %s: %.2f ms, %.0f TFLOP/s\n", win, epi,
                   mode == 0 ? "2 WG/CU x 4 waves, 128x256x32" : "1 WG/CU x 8 waves staggered, 256x256x64", ms, flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
