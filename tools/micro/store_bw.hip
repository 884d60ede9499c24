// Micro-benchmark: global-store throughput per CU as a function of how many CUs store at the same time.
// Each workgroup (256 or 512 threads) writes `bytes_per_wg` with 16-B stores, full 128-B lines, and reports cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void store_kernel(char* out, long bytes_per_wg, int reps, long long* cyc, int row_bytes, long ld) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    char* base = out + (long)blockIdx.x * bytes_per_wg;
    f4 v = {1.f * tid, 2.f, 3.f, 4.f};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        // row-structured: each row has row_bytes contiguous, rows `ld` apart (GEMM tile like)
        const int lanes_per_row = row_bytes / 16;
        const long rows = bytes_per_wg / row_bytes;
        for (long i = tid; i < rows * lanes_per_row; i += nthr) {
            const long row = i / lanes_per_row, c = i - row * lanes_per_row;
            *(f4*)(out + ((long)blockIdx.x * rows + row) * ld + c * 16) = v;
        }
        v[0] += 1.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const long bytes_per_wg = 128 * 1024;
    char* out; long long* cyc;
    const long total = 2048L * bytes_per_wg * 4;
    hipMalloc(&out, total); hipMalloc(&cyc, 4096 * 8);
    std::vector<long long> h(4096);
    for (int thr : {256, 512})
    for (int row_bytes : {128, 512}) 
    for (int grid : {1, 8, 32, 64, 128, 256, 512}) {
        const long ld = row_bytes;  // dense
        for (int it = 0; it < 3; ++it) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            store_kernel<<<grid, thr>>>(out, bytes_per_wg, 8, cyc, row_bytes, ld);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) {
                hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
                double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
                // s_memtime/readcyclecounter ticks at 100 MHz on gfx9 (constant clock) -> report both
                printf("thr=%d row=%dB grid=%4d: %.1f us kernel, %.0f ticks/WG avg, %.2f TB/s aggregate, %.1f B/ns/CU\n", thr, row_bytes, grid, ms * 1e3, avg,
                       grid * bytes_per_wg * 8.0 / (ms * 1e-3) / 1e12, bytes_per_wg * 8.0 / (ms * 1e6) );
            }
        }
    }
    return 0;
}
