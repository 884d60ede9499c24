// Can SCALAR loads warm the L2 for later VECTOR loads of an HBM-resident buffer?  (round 5: DESIGN 6i)
// One workgroup (64 threads = one wave) per CU-ish; each wave owns a 128 KiB chunk of a 1 GiB buffer (cold: beyond the Infinity Cache).
// mode 0: vector-load the chunk, timed.   mode 1: s_load_dword one dword per 64-byte piece (2,048 per chunk) with the results thrown away
// into a reserved SGPR, wait, THEN vector-load the chunk, timed separately.   mode 2: as 1 but one s_load per 128-byte line.
// Prints cycles (s_memtime) of the prefetch phase and of the vector phase, averaged over the waves.
//   hipcc -O3 --offload-arch=gfx950 sload_pf.hip -o sload_pf && ./sload_pf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 128 * 1024;

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void probe(const char* buf, size_t stride, int mode, long long* out, float* sink) {
    const char* base = buf + (size_t)blockIdx.x * stride;
    long long t0 = __builtin_readcyclecounter();
    if (mode >= 1) {
        const int step = mode == 1 ? 64 : 128;
        for (int off = 0; off < CHUNK; off += step * 8) {
            const char* a = base + off;
            if (mode == 1)
                asm volatile("s_load_dword s100, %0, 0x0\n\ts_load_dword s100, %0, 0x40\n\ts_load_dword s100, %0, 0x80\n\ts_load_dword s100, %0, 0xc0\n\t"
                             "s_load_dword s100, %0, 0x100\n\ts_load_dword s100, %0, 0x140\n\ts_load_dword s100, %0, 0x180\n\ts_load_dword s100, %0, 0x1c0\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "s"(a) : "memory", "s100");
            else
                asm volatile("s_load_dword s100, %0, 0x0\n\ts_load_dword s100, %0, 0x80\n\ts_load_dword s100, %0, 0x100\n\ts_load_dword s100, %0, 0x180\n\t"
                             "s_load_dword s100, %0, 0x200\n\ts_load_dword s100, %0, 0x280\n\ts_load_dword s100, %0, 0x300\n\ts_load_dword s100, %0, 0x380\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "s"(a) : "memory", "s100");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    f4 acc = {0, 0, 0, 0};
    // the epilogue's pattern: 8 x 16 B per lane in flight, then the next 8
    for (int off = threadIdx.x * 16; off < CHUNK; off += 64 * 16 * 8) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f4*)(base + off + j * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = t2 - t1; }
    sink[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int waves = 1024;                     // 4 per CU
    const size_t stride = 1 << 20;              // chunks 1 MiB apart: 1 GiB buffer
    char* buf; long long* out; float* sink;
    hipMalloc(&buf, stride * waves); hipMalloc(&out, waves * 16); hipMalloc(&sink, waves * 64 * 4);
    hipMemset(buf, 1, stride * waves);
    char* other; hipMalloc(&other, 1ull << 30);
    std::vector<long long> h(2 * waves);
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(other, rep, 1ull << 30);  // evict the buffer from the caches
            hipDeviceSynchronize();
            hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, buf, stride, mode, out, sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d FAILED: %s\n", mode, hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int i = 0; i < waves; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
            printf("rep %d mode %d: prefetch phase %9.0f cycles, vector phase %9.0f cycles per 128 KiB chunk (avg of %d waves)\n", rep, mode, a / waves, b / waves, waves);
        }
    return 0;
}
