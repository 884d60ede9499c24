// Micro-benchmark 3: how long do a "read slot" (24 ds_read_b128 per wave) and an "MFMA slot" (32 v_mfma 32x32x16 f16)
// take on one CU, alone and when the sibling wave of every SIMD runs the other kind of slot at the same time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
// mode bits: 1 = group 0 active (reads), 2 = group 1 active (mfma), 4 = both groups alternate read/mfma with barriers (staggered)
__global__ __launch_bounds__(512) void k(long long* cyc, float* sink, int iters, int mode, int nread, long long* rt = nullptr) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = wave >> 2;
    for (int i = tid; i < 128 * 1024 / 4; i += 512) ((float*)smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f16v acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    h8 fr[24];
    for (int j = 0; j < 24; ++j) fr[j] = *(const h8*)(smem + lane * 16 + j * 1024);
    const char* base = smem + (wave & 3) * 24 * 1024 + lane * 16;
    auto reads = [&]() {
#pragma unroll
        for (int j = 0; j < 24; ++j) if (j < nread) fr[j] = *(const h8*)(base + j * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[(a & 1) * 4 + ks], fr[8 + (a >> 1) * 4 + ks], acc[a], 0, 0, 0);
    };
#define SB() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    long long r0 = wall_clock64();
    long long t0 = __builtin_readcyclecounter();
    if (mode & 4) {
        if (grp == 0) {
            for (int it = 0; it < iters; ++it) { reads(); SB(); mfmas(); SB(); }
            SB();
        } else {
            SB();
            for (int it = 0; it < iters; ++it) { reads(); SB(); mfmas(); SB(); }
        }
    } else if (mode & 8) {   // lock-step: everyone reads, then everyone computes
        for (int it = 0; it < iters; ++it) { reads(); SB(); mfmas(); SB(); }
    } else {
        if (grp == 0 && (mode & 1)) for (int it = 0; it < iters; ++it) { __builtin_amdgcn_sched_barrier(0); reads(); __builtin_amdgcn_sched_barrier(0); }
        if (grp == 1 && (mode & 2)) for (int it = 0; it < iters; ++it) { __builtin_amdgcn_sched_barrier(0); mfmas(); __builtin_amdgcn_sched_barrier(0); }
    }
    long long t1 = __builtin_readcyclecounter();
    long long r1 = wall_clock64();
    if (rt && tid == 0) rt[blockIdx.x] = r1 - r0;
    f16v sv = acc[0];
    for (int a = 1; a < 8; ++a) sv += acc[a];
    float s = sv[0] + sv[5];
    if (s == 12345.f) for (int j = 0; j < 24; ++j) s += (float)fr[j][0];
    sink[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<long long> h(2048);
    const int iters = 2000;
    for (int grid : {1, 256})
    for (int nread : {24, 12, 0})
    for (int mode : {1, 2, 3, 4, 8}) {
        for (int it = 0; it < 2; ++it) { k<<<grid, 512, 128 * 1024>>>(cyc, sink, iters, mode, nread); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, grid * 64, hipMemcpyDeviceToHost);
        double g0 = 0, g1 = 0;
        for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? g0 : g1) += h[b * 8 + w];
        g0 /= grid * 4.0 * iters; g1 /= grid * 4.0 * iters;
        const char* names[] = {"", "reads only (grp0)", "mfma only (grp1)", "reads(grp0) || mfma(grp1) free-running", "staggered with barriers", "", "", "", "lock-step with barriers"};
        printf("grid=%3d nread=%2d %-42s: grp0 %.0f cyc/iter, grp1 %.0f cyc/iter\n", grid, nread, names[mode], g0, g1);
    }
    // clock check: long staggered run on the whole chip, wall time vs cycle counter vs 100 MHz real-time counter
    long long* rt; hipMalloc(&rt, 256 * 8);
    for (int grid : {1, 256}) for (int mode : {4, 2}) {
        const int it2 = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, 512, 128 * 1024>>>(cyc, sink, it2, mode, 24, rt); hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 512, 128 * 1024>>>(cyc, sink, it2, mode, 24, rt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
        long long r; hipMemcpy(&r, rt, 8, hipMemcpyDeviceToHost);
        const double flops = (double)grid * 8 * 32 * 32768.0 * it2 * (mode == 2 ? 0.5 : 1.0);
        printf("grid=%3d mode=%d: %.2f ms, cycle counter %lld (%.3f GHz vs wall), realtime ticks %lld (%.1f MHz), %.0f TFLOP/s\n", grid, mode, ms, h[4], h[4] / (ms * 1e6), r, r / (ms * 1e3), flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
