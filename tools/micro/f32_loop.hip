// Micro-benchmark 7: what between the MFMAs of an f32 product loop costs matrix-pipe time?
// The K-tile loop of the skinny f32 kernels (64 v_mfma_f32_16x16x4_f32 per wave and K-tile on 4 accumulators; DESIGN 6d: 55-72 TF
// against 140-155 TF for the bare MFMA stream, tools/micro/f32_peak.hip) rebuilt from switchable parts, every load L1/LDS-hot:
//   bit 0: 8 global_load_dwordx4 per thread and K-tile (operand rows + weight-tile staging), consumed one K-tile later
//   bit 1: 4 ds_write_b128 of the staged weight tile + the workgroup barrier behind them
//   bit 2: 16 ds_read_b128 fragment reads per K-tile feeding the MFMAs (else: fragments stay in registers)
//   bit 3: a bare workgroup barrier per K-tile (when bit 1 is off)
//   bit 5: 8 global_load_lds_dwordx4 per wave and K-tile (the same 8 KB, landing in LDS, no VGPRs) instead of bit 0's register loads
//   bit 4: the 8 global loads of bit 0 are issued one by one, after every 8th MFMA, instead of back to back ahead of the MFMAs
// 256 CUs x {1, 2, 3} workgroups of 256 threads, ITER K-tiles each.  Prints TFLOP/s per mode.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int mode>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ g, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float wl[2][64 * 68];
    __shared__ __attribute__((aligned(16))) float dma[(mode & 32) ? 4 : 1][(mode & 32) ? 8 * 256 : 4];   // bit 5: landing area of the LDS-DMA pieces, 8 KB per wave
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    for (int i = tid; i < 2 * 64 * 68; i += 256) (&wl[0][0])[i] = 0.001f * (i & 255);
    __syncthreads();
    f4 acc[4], w[16], xa[4], st[4], xn[4];
    for (int q = 0; q < 4; ++q) { acc[q] = (f4){0.f, 0.f, 0.f, 0.f}; xa[q] = (f4){0.01f * q, 0.02f, 0.03f, 0.04f}; st[q] = xa[q]; xn[q] = xa[q]; }
    const float* wb0 = &wl[0][0] + j * 68 + 4 * gq;
    for (int q = 0; q < 16; ++q) w[q] = *(const f4*)(wb0 + 16 * (q >> 2) * 68 + 16 * (q & 3));
    const float* gp = g + (size_t)(blockIdx.x & 255) * 4096 + tid * 4;
    for (int it = 0; it < iters; ++it) {
        float* buf = wl[it & 1];
        if (mode & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f4*)(buf + ((tid >> 4) + 16 * q) * 68 + (tid & 15) * 4) = st[q];
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        } else if (mode & 8) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr bool spread = (mode & 16) != 0;          // bit 4: the 8 loads go out one by one, after every 8th MFMA
        if (mode & 32) {                               // bit 5: the same 8 KB per wave as 8 global_load_lds_dwordx4 (no VGPRs)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 1024 * (q & 3) + 16 * (q >> 2)),
                                                 (__attribute__((address_space(3))) void*)(&dma[tid >> 6][q * 256]), 16, 0, 0);
        }
        if (mode & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { xa[q] = xn[q]; }
            if (!spread) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { st[q] = *(const f4*)(gp + 1024 * q); xn[q] = *(const f4*)(gp + 1024 * q + 16); }
            }
        }
        const float* wb = buf + j * 68 + 4 * gq;
        f4 xc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xc[q] = xa[q];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 ws[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) ws[f] = (mode & 4) ? *(const f4*)(wb + 16 * f * 68 + 16 * s) : w[4 * f + s];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[f][e], xc[s][e], acc[f], 0, 0, 0);
                if ((mode & 1) && spread && (e & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int q = 2 * s + (e >> 1);          // 0..7
                    if (q < 4) st[q] = *(const f4*)(gp + 1024 * q); else xn[q - 4] = *(const f4*)(gp + 1024 * (q - 4) + 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    f4 s4 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    sink[blockIdx.x * 256 + tid] = s4[0] + s4[1] + s4[2] + s4[3] + st[0][0] + xn[1][1];
}
int main() {
    float *g, *sink;
    hipMalloc(&g, 256 * 4096 * 4 + 65536); hipMemset(g, 0, 256 * 4096 * 4 + 65536); hipMalloc(&sink, 768 * 256 * 4);
    const int iters = 2000;
    const char* names[32] = {"MFMAs only", "+ global loads", "+ LDS staging + barrier", "+ loads + staging + barrier", "+ fragment reads", "+ loads + fragment reads",
                             "+ staging + barrier + fragment reads", "ALL (the kernel's loop)", "+ bare barrier", "+ loads + bare barrier", "", "", "+ fragment reads + bare barrier", "+ loads + fragment reads + bare barrier", "", "",
                             "", "+ loads SPREAD among the MFMAs", "", "", "", "+ spread loads + fragment reads", "", "ALL, loads spread among the MFMAs", "", "", "", "", "", "", "", ""};
    const char* dman[2] = {"+ 8 LDS-DMA pieces (global_load_lds) instead of register loads", "+ LDS-DMA pieces + staging + barrier + fragment reads"};
    for (int wgs : {1, 2, 3})
        for (int mode : {0, 1, 17, 32, 2, 4, 6, 7, 23, 38}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto go = [&]() {
                switch (mode) {
#define C(M) case M: k<M><<<256 * wgs, 256>>>(g, sink, iters); break
                    C(0); C(1); C(17); C(32); C(2); C(4); C(6); C(7); C(23); C(38);
#undef C
                }
            };
            go(); hipDeviceSynchronize();
            hipEventRecord(e0); go(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)256 * wgs * 4 * iters * 64 * 2048.0;
            printf("%d workgroup(s) per CU, mode %2d %-42s: %7.2f TFLOP/s  (%.0f cycles per K-tile and wave at 2.1 GHz)\n", wgs, mode, mode >= 32 ? dman[mode == 38] : names[mode], flop / ms / 1e9,
                   ms * 1e-3 * 2.1e9 / iters);
        }
    return 0;
}
