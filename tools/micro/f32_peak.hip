// Micro-benchmark 6: the practical ceiling of the f32-input matrix cores.  Every wave issues back-to-back
// v_mfma_f32_16x16x4_f32 on NA independent accumulators from registers (no memory at all); 1 / 2 / 4 waves per SIMD on 1 CU and on
// the whole chip.  Prints TFLOP/s from event time and cycles per MFMA and SIMD from the cycle counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NA>
__global__ __launch_bounds__(1024) void k(long long* cyc, float* sink, int iters) {
    const int tid = threadIdx.x;
    f4 acc[NA];
    for (int a = 0; a < NA; ++a) acc[a] = (f4){0.f, 0.f, 0.f, 0.f};
    float x = 0.001f * (tid & 63), y = 0.002f * (tid & 31);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    f4 s = acc[0];
    for (int a = 1; a < NA; ++a) s += acc[a];
    sink[blockIdx.x * 1024 + tid] = s[0] + s[3];
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}
template <int NA>
void run(long long* cyc, float* sink) {
    const int iters = 4000;
    std::vector<long long> h(256 * 16);
    for (int grid : {1, 256})
        for (int threads : {256, 512, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<NA><<<grid, threads>>>(cyc, sink, iters); hipDeviceSynchronize();
            hipEventRecord(e0); k<NA><<<grid, threads>>>(cyc, sink, iters); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, grid * 16 * 8, hipMemcpyDeviceToHost);
            double c = 0; const int waves = threads / 64;
            for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) c += h[b * 16 + w];
            c /= (double)grid * waves;
            const double nm = (double)iters * 16 * NA;                 // MFMAs per wave
            const double flop = nm * 2048.0 * waves * grid;
            printf("NA=%d grid=%3d waves/SIMD=%d : %7.2f TFLOP/s, %5.1f cycles per MFMA and SIMD (cycle counter), %.3f ms\n", NA, grid, waves / 4,
                   flop / ms / 1e9, c / nm / (waves / 4), ms);
        }
}
int main() {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    run<1>(cyc, sink); run<2>(cyc, sink); run<4>(cyc, sink);
    return 0;
}
